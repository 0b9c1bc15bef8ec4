"""SparkLLM -- the speech-token generator behind ``AutoModelForCausalLM.generate`` as the
reference calls it (``cli/SparkTTS.py:197-204``), running on the HIP kernels of ``smi_llm.hip``.

``generate`` keeps the HF call shape (``input_ids`` (B, P), ``attention_mask``,
``max_new_tokens``, ``do_sample``, ``eos_token_id``, ``pad_token_id``) and returns (B, P + N)
ids, prompt included, exactly like the reference expects when it slices the prompt off at
``cli/SparkTTS.py:207-210``.  Greedy by default; ``do_sample=True`` runs the reference's temperature → top-k →
top-p → multinomial chain on the device (``k_sample``).
"""
from __future__ import annotations

import ctypes as C
import json
import warnings
from pathlib import Path
from typing import Iterable, List, Mapping, Optional, Sequence, Union

import numpy as np
import torch

from . import _lib
from .arena import llm_cfg_struct, pack_llm_arena
from .config import LLMConfig


EosLike = Union[None, int, Iterable[int]]


def eos_ids_from_generation_config(model_dir: Union[str, Path], cfg: Optional[LLMConfig] = None) -> List[int]:
    """The ids HF ``generate()`` stops on when the caller passes no ``eos_token_id`` -- which is how the reference
    calls it (``cli/SparkTTS.py:197-204``): every id of ``generation_config.json``'s ``eos_token_id`` (an int or a
    list), else ``config.json``'s."""
    ids: List[int] = []
    g = Path(model_dir) / "generation_config.json"
    if g.exists():
        e = json.loads(g.read_text()).get("eos_token_id")
        if e is not None:
            ids = [int(x) for x in (e if isinstance(e, (list, tuple)) else [e])]
    if not ids and cfg is not None and cfg.eos_token_id is not None:
        e = cfg.eos_token_id
        ids = [int(x) for x in (e if isinstance(e, (list, tuple)) else [e])]
    return ids


class SparkLLM:
    def __init__(self, cfg: LLMConfig, weights: Mapping[str, np.ndarray],
                 device: Union[str, torch.device] = "cuda:0", max_slots: int = 1,
                 max_positions: int = 4096, kv_dtype: str = "bf16", use_graph: bool = True,
                 arena: Optional[torch.Tensor] = None, eos_token_ids: EosLike = None,
                 kv_page_tokens: int = 0, kv_pages: int = 0, diag: bool = False, weights_exact: bool = False):
        """``eos_token_ids``: the model's default stop ids (``generation_config.json``; see
        ``eos_ids_from_generation_config``).  ``generate()`` falls back to them when the caller passes none, like HF.
        ``kv_page_tokens`` / ``kv_pages``: paged KV cache -- a pool of ``kv_pages`` pages of ``kv_page_tokens`` tokens
        shared by the ``max_slots`` sequences instead of ``max_positions`` reserved tokens per slot (sparkmi.h).
        ``weights_exact``: the verification mode of ``smi_llm_cfg.weights_exact`` -- the arena keeps the matrices in fp32 and every
        GEMM is an exact fp32 chain, so a checkpoint SAVED in fp32 (the published Spark-TTS-0.5B LLM is) gives the fp32 PyTorch
        CPU path's tokens instead of those of its bf16 rounding; opt-in, ~4x slower, 2x the weight bytes.
        ``diag``: put the handle on ``libsparkmi_diag.so`` (timing probes, scratch dumps, SPARKMI_* switches, the one-row engine:
        ``include/sparkmi_debug.h``) instead of the product library -- tools, bench probes and tests only."""
        cfg.validate()
        self.cfg = cfg
        if eos_token_ids is None and cfg.eos_token_id is not None:
            eos_token_ids = cfg.eos_token_id
        self.default_eos = self._eos_list(eos_token_ids)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.SparkMIError("SparkLLM runs on an MI355X only (device must be cuda:N); there is no CPU path")
        self._lib = _lib.pick(diag)
        torch.cuda.set_device(self.device)
        _lib.require_gfx950()
        self.max_slots, self.max_positions = max_slots, max_positions
        self._cs = llm_cfg_struct(cfg, max_slots, max_positions, kv_dtype, use_graph, kv_page_tokens, kv_pages, weights_exact)
        if arena is None:
            host = pack_llm_arena(cfg, weights, self._cs)
            arena = torch.from_numpy(host).to(self.device)
        else:
            # a packed arena says how it was packed (section LLM_TAG): its W_down tile order wins over what the environment
            # would choose now; any other disagreement with the config is refused by smi_llm_create
            off, nb = C.c_size_t(), C.c_size_t()
            self._lib.check(self._lib.smi_llm_arena_section(C.byref(self._cs), _lib.LLM_TAG, 0, C.byref(off), C.byref(nb)),
                            "smi_llm_arena_section")
            if arena.numel() >= off.value + nb.value:
                tag = _lib.LLMArenaTag.from_buffer_copy(arena[off.value: off.value + nb.value].cpu().numpy().tobytes())
                if tag.magic == b"SMIARENA":
                    self._cs.wd_plain = tag.wd_plain
        self.arena = arena  # uint8 device tensor; must outlive the handle
        self._h = C.c_void_p()
        self._lib.check(self._lib.smi_llm_create(C.byref(self._cs), C.c_void_p(arena.data_ptr()),
                                            arena.numel(), C.byref(self._h)), "smi_llm_create")

    # ------------------------------------------------------------------ plumbing
    def _stream(self) -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.smi_llm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @classmethod
    def from_pretrained(cls, llm_dir: Union[str, Path], device: Union[str, torch.device] = "cuda:0", **kw) -> "SparkLLM":
        """Stands where ``AutoModelForCausalLM.from_pretrained(f"{model_dir}/LLM")`` stands (``cli/SparkTTS.py:49``):
        config.json, safetensors and generation_config.json (stop ids) of a checkpoint directory."""
        from .weights import load_llm_state
        llm_dir = Path(llm_dir)
        cfg = LLMConfig.from_json(llm_dir / "config.json")
        return cls(cfg, load_llm_state(llm_dir), device, eos_token_ids=eos_ids_from_generation_config(llm_dir, cfg), **kw)

    @staticmethod
    def _eos_list(eos: EosLike) -> List[int]:
        if eos is None:
            return []
        ids = [int(eos)] if isinstance(eos, (int, np.integer)) else [int(e) for e in eos]
        if len(ids) > _lib.SMI_MAX_EOS:
            raise ValueError(f"{len(ids)} eos ids; the step kernel checks at most {_lib.SMI_MAX_EOS}")
        return ids

    def _eos_args(self, eos: EosLike):
        ids = self._eos_list(eos)
        arr = (C.c_int64 * max(len(ids), 1))(*ids)
        return arr, len(ids)

    # ------------------------------------------------------------------ generation
    def prefill(self, prompts: Sequence[Sequence[int]], eos_token_id: EosLike = None) -> None:
        """``eos_token_id``: an id, a list of ids (any of them stops the sequence) or None (never stop)."""
        B = len(prompts)
        lens = np.array([len(p) for p in prompts], dtype=np.int32)
        pmax = int(lens.max())
        ids = np.zeros((B, pmax), dtype=np.int64)
        for b, p in enumerate(prompts):
            ids[b, : len(p)] = np.asarray(p, dtype=np.int64)
        eos_arr, n_eos = self._eos_args(eos_token_id)
        self._lib.check(self._lib.smi_llm_prefill(
            self._h, ids.ctypes.data_as(C.POINTER(C.c_int64)), lens.ctypes.data_as(C.POINTER(C.c_int32)),
            B, pmax, eos_arr, n_eos, self._stream()), "smi_llm_prefill")
        self._B, self._lens = B, lens

    def decode(self, n_steps: int) -> None:
        self._lib.check(self._lib.smi_llm_decode(self._h, int(n_steps), self._stream()), "smi_llm_decode")

    def all_done(self) -> bool:
        d = C.c_int(0)
        self._lib.check(self._lib.smi_llm_all_done(self._h, C.byref(d), self._stream()), "smi_llm_all_done")
        return bool(d.value)

    def tokens(self, cap: int) -> List[List[int]]:
        out = np.zeros((self._B, cap), dtype=np.int64)
        lens = np.zeros(self._B, dtype=np.int32)
        self._lib.check(self._lib.smi_llm_get_tokens(
            self._h, out.ctypes.data_as(C.POINTER(C.c_int64)), lens.ctypes.data_as(C.POINTER(C.c_int32)),
            cap, self._stream()), "smi_llm_get_tokens")
        return [out[b, : lens[b]].tolist() for b in range(self._B)]

    def set_sampling(self, do_sample: bool, temperature: float = 0.8, top_k: int = 50, top_p: float = 0.95,
                     seed: Optional[int] = None) -> None:
        if seed is None:
            seed = int(np.random.SeedSequence().generate_state(1, dtype=np.uint64)[0]) if do_sample else 0
        self._lib.check(self._lib.smi_llm_set_sampling(self._h, int(bool(do_sample)), float(temperature), int(top_k),
                                                  float(top_p), int(seed) & (2 ** 64 - 1)), "smi_llm_set_sampling")

    def generate_ids(self, prompts: Sequence[Sequence[int]], max_new_tokens: int,
                     eos_token_id: EosLike = None, check_every: int = 32, do_sample: bool = False,
                     temperature: float = 0.8, top_k: int = 50, top_p: float = 0.95,
                     seed: Optional[int] = None) -> List[List[int]]:
        """Generation for B ragged prompts; returns only the new ids per sequence (eos included
        when emitted).  Greedy by default; ``do_sample`` selects the reference's temperature /
        top-k / top-p sampler.  With no eos the whole run is enqueued without a host sync."""
        self.set_sampling(do_sample, temperature, top_k, top_p, seed)
        if max_new_tokens < 1:
            raise ValueError("max_new_tokens must be >= 1")
        longest = max(len(p) for p in prompts)
        if longest + max_new_tokens > self.max_positions:
            raise ValueError(f"prompt ({longest}) + max_new_tokens ({max_new_tokens}) exceeds "
                             f"max_positions ({self.max_positions})")
        self.prefill(prompts, eos_token_id)
        remaining = max_new_tokens - 1
        if not self._eos_list(eos_token_id):
            self.decode(remaining)
        else:
            while remaining > 0 and not self.all_done():
                n = min(check_every, remaining)
                self.decode(n)
                remaining -= n
        return self.tokens(max_new_tokens)

    @torch.no_grad()
    def generate(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                 max_new_tokens: int = 3000, do_sample: bool = False, eos_token_id: EosLike = None,
                 pad_token_id: Optional[int] = None, temperature: float = 1.0, top_k: int = 50,
                 top_p: float = 1.0, seed: Optional[int] = None, **unused) -> torch.Tensor:
        """HF-shaped entry.  ``attention_mask`` marks real tokens of right- or left-padded rows.  Like HF's
        ``generate``: with no ``eos_token_id`` the model's own stop ids apply (``generation_config.json``, given at
        construction) -- the reference's call at ``cli/SparkTTS.py:197-204`` relies on exactly that -- and generation
        ends at the context limit when ``max_new_tokens`` would pass it (HF's limit is the model's 32k positions; here it
        is ``max_positions``, so a long clone prompt shortens the budget instead of failing)."""
        ids = input_ids.detach().cpu().numpy().astype(np.int64)
        if ids.ndim == 1:
            ids = ids[None]
        if attention_mask is not None:
            msk = attention_mask.detach().cpu().numpy().astype(bool)
            prompts = [ids[b][msk[b]].tolist() for b in range(ids.shape[0])]
        else:
            prompts = [ids[b].tolist() for b in range(ids.shape[0])]
        eos = self._eos_list(eos_token_id) or self.default_eos
        room = self.max_positions - max(len(p) for p in prompts)
        if room < 1:
            raise ValueError(f"prompt of {max(len(p) for p in prompts)} tokens leaves no room in max_positions={self.max_positions}")
        new = self.generate_ids(prompts, min(int(max_new_tokens), room), eos, do_sample=do_sample, temperature=temperature,
                                top_k=top_k, top_p=top_p, seed=seed)
        pad = pad_token_id if pad_token_id is not None else (eos[0] if eos else 0)
        n = max(len(t) for t in new)
        out = np.full((ids.shape[0], ids.shape[1] + n), pad, dtype=np.int64)
        out[:, : ids.shape[1]] = ids
        for b, t in enumerate(new):
            out[b, ids.shape[1]: ids.shape[1] + len(t)] = t
        return torch.from_numpy(out).to(input_ids.device)

    # ------------------------------------------------------------------ continuous batching
    def session_begin(self, eos_token_id: EosLike = None) -> None:
        """Empty in-flight-batching session: sequences are admitted and retired between decode steps."""
        eos_arr, n_eos = self._eos_args(eos_token_id)
        self._lib.check(self._lib.smi_llm_session_begin(self._h, eos_arr, n_eos, self._stream()), "smi_llm_session_begin")

    def admit(self, prompts: Sequence[Sequence[int]]) -> List[int]:
        """Prefill new prompts into free KV slots (first token emitted); returns their slot ids."""
        n = len(prompts)
        lens = np.array([len(p) for p in prompts], dtype=np.int32)
        pmax = int(lens.max())
        ids = np.zeros((n, pmax), dtype=np.int64)
        for b, p in enumerate(prompts):
            ids[b, : len(p)] = np.asarray(p, dtype=np.int64)
        slots = np.zeros(n, dtype=np.int32)
        self._lib.check(self._lib.smi_llm_admit(self._h, ids.ctypes.data_as(C.POINTER(C.c_int64)), lens.ctypes.data_as(C.POINTER(C.c_int32)),
                                           n, pmax, slots.ctypes.data_as(C.POINTER(C.c_int32)), self._stream()), "smi_llm_admit")
        return slots.tolist()

    def retire(self, slot: int) -> None:
        self._lib.check(self._lib.smi_llm_retire(self._h, int(slot), self._stream()), "smi_llm_retire")

    def retire_many(self, slots: Sequence[int]) -> None:
        """Several sequences leave at once; no host round trip (the device row list is compacted in place)."""
        arr = np.asarray(list(slots), dtype=np.int32)
        self._lib.check(self._lib.smi_llm_retire_many(self._h, arr.ctypes.data_as(C.POINTER(C.c_int32)), len(arr), self._stream()),
                   "smi_llm_retire_many")

    def slots_tokens(self, slots: Sequence[int], cap: int):
        """[(tokens, finished)] of several slots (live or retired and not yet reused) in one device round trip."""
        arr = np.asarray(list(slots), dtype=np.int32)
        out = np.zeros((len(arr), max(cap, 1)), dtype=np.int64)
        n = np.zeros(len(arr), dtype=np.int32)
        fin = np.zeros(len(arr), dtype=np.int32)
        self._lib.check(self._lib.smi_llm_slots_tokens(self._h, arr.ctypes.data_as(C.POINTER(C.c_int32)), len(arr),
                                                  out.ctypes.data_as(C.POINTER(C.c_int64)), max(cap, 1), n.ctypes.data_as(C.POINTER(C.c_int32)),
                                                  fin.ctypes.data_as(C.POINTER(C.c_int32)), self._stream()), "smi_llm_slots_tokens")
        return [(out[i, : n[i]].tolist(), bool(fin[i])) for i in range(len(arr))]

    def slot_tokens(self, slot: int, cap: int):
        """(tokens emitted so far by the sequence in ``slot``, finished flag)."""
        out = np.zeros(max(cap, 1), dtype=np.int64)
        n, fin = C.c_int32(0), C.c_int32(0)
        self._lib.check(self._lib.smi_llm_slot_tokens(self._h, int(slot), out.ctypes.data_as(C.POINTER(C.c_int64)), cap, C.byref(n),
                                                 C.byref(fin), self._stream()), "smi_llm_slot_tokens")
        return out[: n.value].tolist(), bool(fin.value)

    def kv_pages(self):
        """(pages in the pool, pages free) of a paged KV cache; (0, 0) when the cache is not paged."""
        tot, free = C.c_int32(0), C.c_int32(0)
        self._lib.check(self._lib.smi_llm_kv_pages(self._h, C.byref(tot), C.byref(free)), "smi_llm_kv_pages")
        return tot.value, free.value

    def status(self):
        """(tokens emitted, finished flag) per KV slot, as two int32 arrays of SMI_MAX_ROWS -- one device round trip."""
        cnt = np.zeros(_lib.SMI_MAX_ROWS, dtype=np.int32)
        fin = np.zeros(_lib.SMI_MAX_ROWS, dtype=np.int32)
        self._lib.check(self._lib.smi_llm_status(self._h, cnt.ctypes.data_as(C.POINTER(C.c_int32)), fin.ctypes.data_as(C.POINTER(C.c_int32)),
                                            self._stream()), "smi_llm_status")
        return cnt, fin

    def serve(self, requests, max_live: Optional[int] = None, decode_stride: int = 8):
        """In-flight batching driver: ``requests`` yields (key, prompt ids, max_new_tokens, eos id or None -- one eos for
        the session: the first request's); yields (key, new ids) as each sequence finishes.  New requests are admitted
        whenever a slot is free, so short utterances never wait for long ones."""
        it = iter(requests)
        max_live = min(max_live or self.max_slots, self.max_slots)
        live = {}                      # slot -> (key, max_new)
        pending = next(it, None)
        started = False
        while pending is not None or live:
            batch = []
            while pending is not None and len(live) + len(batch) < max_live:
                batch.append(pending)
                pending = next(it, None)
            if batch:                      # all free slots are filled by ONE admission (one prefill launch sequence)
                if not started:
                    self.session_begin(batch[0][3])
                    started = True
                slots = self.admit([list(b[1]) for b in batch])
                for slot, b in zip(slots, batch):
                    live[slot] = (b[0], int(b[2]))
            self.decode(decode_stride)
            cnt, fin = self.status()
            leave = [slot for slot in live if fin[slot] or cnt[slot] >= live[slot][1]]
            if leave:                      # their tokens in one round trip, their rows dropped on the device
                got = self.slots_tokens(leave, max(live[slot][1] for slot in leave))
                self.retire_many(leave)
                for slot, (toks, _) in zip(leave, got):
                    key, max_new = live.pop(slot)
                    yield key, toks[:max_new]

    def generate_ragged(self, prompts: Sequence[Sequence[int]], max_new_tokens: Sequence[int], eos_token_id: EosLike = None,
                        check_every: int = 16, on_prefilled=None) -> List[List[int]]:
        """One batch of prompts with PER-ROW token budgets, rows retired as they finish (their budget, or eos): the decode
        step then runs on the rows still alive instead of padding finished ones to the longest (HF ``generate`` pads; the
        reference's TensorRT-LLM deployment batches in flight, run.sh:50-65).  Rows are independent in every kernel, so
        row i's tokens are exactly those of ``generate_ids`` truncated to its budget.  The captured step of every row
        count is cached in the library, so retiring costs a row-table upload, not a graph capture.  Greedy or the
        sampler set by ``set_sampling``; ``on_prefilled()`` is called after the prompts' prefill was enqueued."""
        n = len(prompts)
        want = [int(w) for w in max_new_tokens]
        if n != len(want) or n > self.max_slots or min(want) < 1:
            raise ValueError("generate_ragged: one budget >= 1 per prompt, at most max_slots prompts")
        if max(len(p) + w for p, w in zip(prompts, want)) > self.max_positions:
            raise ValueError("generate_ragged: prompt + budget exceeds max_positions")
        eos = self._eos_list(eos_token_id)
        self.session_begin(eos or None)
        slots = self.admit([list(p) for p in prompts])
        if on_prefilled is not None:
            on_prefilled()
        live = {slot: i for i, slot in enumerate(slots)}
        done = 1                                   # tokens every live row has emitted (the prefill emits the first)
        while live:
            fin = None
            if eos:
                _, fin = self.status()             # one device round trip
            leave = [slot for slot, i in live.items() if done >= want[i] or (fin is not None and fin[slot])]
            if leave:
                self.retire_many(leave)            # enqueued behind the steps so far: no host round trip
                for slot in leave:
                    del live[slot]
            if not live:
                break
            steps = min(want[i] for i in live.values()) - done
            if eos:
                steps = min(steps, check_every)
            self.decode(steps)
            done += steps
        # histories are per KV slot and stay until a slot is reused: all rows in one round trip
        got = self.slots_tokens(slots, max(want))
        return [t[: want[i]] for i, (t, _) in enumerate(got)]

    # ------------------------------------------------------------------ test / bench entries
    def forward_logits(self, ids: Sequence[int]) -> torch.Tensor:
        """Teacher-forced logits (S, V) for one sequence fed at positions 0..S-1."""
        a = np.asarray(ids, dtype=np.int64)
        out = torch.empty((a.shape[0], self.cfg.vocab_size), dtype=torch.float32, device=self.device)
        self._lib.check(self._lib.smi_llm_forward_logits(
            self._h, a.ctypes.data_as(C.POINTER(C.c_int64)), a.shape[0], C.c_void_p(out.data_ptr()),
            self._stream()), "smi_llm_forward_logits")
        return out

    # ------------------------------------------------------------------ diagnostics (include/sparkmi_debug.h; diag=True handles)
    def _need_diag(self, what: str) -> None:
        if not self._lib.is_diag:
            raise _lib.SparkMIError(f"SparkLLM.{what} is a diagnostics entry (include/sparkmi_debug.h): construct the engine with "
                                    "diag=True (libsparkmi_diag.so); the product library does not export it")

    def engine_info(self) -> dict:
        """Whether one-row decode steps run as one persistent launch, and why / why not."""
        self._need_diag("engine_info")
        on = C.c_int32(0)
        info = (C.c_int32 * 4)()
        why = C.create_string_buffer(200)
        self._lib.check(self._lib.smi_llm_engine(self._h, C.byref(on), info, why, 200), "smi_llm_engine")
        return {"enabled": bool(on.value), "built": bool(info[3]), "cus": int(info[0]), "images_per_wave": int(info[1]),
                "lds_bytes": int(info[2]), "why": why.value.decode(errors="replace")}

    def set_engine(self, on: bool) -> None:
        """Runtime switch between the engine and the four-launches-per-layer path (same bits; A/B runs and tests)."""
        self._need_diag("set_engine")
        self._lib.check(self._lib.smi_llm_set_engine(self._h, 1 if on else 0), "smi_llm_set_engine")

    def engine_stamps(self) -> np.ndarray:
        """(3, layers, 16) microseconds of the last engine launch (needs SPARKMI_ENGINE_STAMPS=1 in the environment)."""
        self._need_diag("engine_stamps")
        n = 3 * self.cfg.num_hidden_layers * 16
        out = (C.c_double * n)()
        self._lib.check(self._lib.smi_llm_engine_stamps(self._h, out, n), "smi_llm_engine_stamps")
        return np.array(out, dtype=np.float64).reshape(3, self.cfg.num_hidden_layers, 16)

    def debug_hidden(self) -> np.ndarray:
        """The residual row of row 0 as the last step left it (tests)."""
        self._need_diag("debug_hidden")
        out = np.zeros(self.cfg.hidden_size, dtype=np.float32)
        self._lib.check(self._lib.smi_llm_debug_hidden(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), out.size), "smi_llm_debug_hidden")
        return out

    def debug_read(self, what: int) -> np.ndarray:
        """Raw bytes of one scratch buffer (include/sparkmi_debug.h: smi_llm_debug_read)."""
        self._need_diag("debug_read")
        buf = np.zeros(1 << 22, dtype=np.uint8)
        got = C.c_size_t(0)
        self._lib.check(self._lib.smi_llm_debug_read(self._h, int(what), buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(got)),
                   "smi_llm_debug_read")
        return buf[: got.value].copy()

    # ---- op-level entries (include/sparkmi_debug.h: smi_llm_debug_layer / _set_kv / _get_kv)
    def debug_set_kv(self, layer: int, slot: int, k: np.ndarray, v: np.ndarray, pos0: int = 0) -> None:
        """k, v: (n, num_kv_heads, 64) fp32 in transformers' dim order (keys already rotated) -> cache positions pos0.."""
        self._need_diag("debug_set_kv")
        k = np.ascontiguousarray(k, dtype=np.float32)
        v = np.ascontiguousarray(v, dtype=np.float32)
        assert k.shape == v.shape == (k.shape[0], self.cfg.num_key_value_heads, 64)
        fp = C.POINTER(C.c_float)
        self._lib.check(self._lib.smi_llm_debug_set_kv(self._h, layer, slot, pos0, k.shape[0], k.ctypes.data_as(fp), v.ctypes.data_as(fp)),
                        "smi_llm_debug_set_kv")

    def debug_get_kv(self, layer: int, slot: int, pos0: int, n: int):
        self._need_diag("debug_get_kv")
        k = np.zeros((n, self.cfg.num_key_value_heads, 64), dtype=np.float32)
        v = np.zeros_like(k)
        fp = C.POINTER(C.c_float)
        self._lib.check(self._lib.smi_llm_debug_get_kv(self._h, layer, slot, pos0, n, k.ctypes.data_as(fp), v.ctypes.data_as(fp)),
                        "smi_llm_debug_get_kv")
        return k, v

    @staticmethod
    def _from_triples(raw: np.ndarray, K: int, M: int) -> np.ndarray:
        """[K / 32][3][4][M][8] bf16 (hi, mid, lo planes of an exact split) -> (M, K) fp32, k = 32 * tile + 8 * k8 + e."""
        from .weights import bf16_bits_to_f32
        a = bf16_bits_to_f32(raw.view(np.uint16)).reshape(K // 32, 3, 4, M, 8)
        x = (a[:, 0] + a[:, 1]) + a[:, 2]                       # exact: the three terms do not overlap
        return np.ascontiguousarray(x.transpose(2, 0, 1, 3)).reshape(M, K)

    def debug_layer(self, layer: int, rows, hidden: np.ndarray, stage: int) -> dict:
        """One layer's kernels up to ``stage`` on caller rows (``rows``: (slot, pos) pairs; ``hidden``: (M, hidden) fp32) through
        the step's own launch builders; returns that stage's outputs in transformers' layouts:
        0 {"q" (M, heads, 64), "k" / "v" (M, kv heads, 64)}, 1 {"attn" (M, heads * 64)}, 2 {"h" (M, hidden)},
        3 {"act" (M, intermediate)}, 4 {"h" (M, hidden)}."""
        self._need_diag("debug_layer")
        c = self.cfg
        rows = np.ascontiguousarray(np.asarray(rows, dtype=np.int32).reshape(-1, 2))
        M = rows.shape[0]
        hidden = np.ascontiguousarray(hidden, dtype=np.float32)
        assert hidden.shape == (M, c.hidden_size)
        self._lib.check(self._lib.smi_llm_debug_layer(self._h, layer, M, rows.ctypes.data_as(C.POINTER(C.c_int32)),
                                                      hidden.ctypes.data_as(C.POINTER(C.c_float)), stage), "smi_llm_debug_layer")
        nh, Q = c.num_attention_heads, c.num_attention_heads * 64
        unpair = (np.arange(64) >> 1) + 32 * (np.arange(64) & 1)          # kernel row i of a head holds transformers' dim unpair[i]
        if stage == 0:
            qk = self.debug_read(0).view(np.float32).reshape(M, nh, 64)
            q = np.empty_like(qk)
            q[:, :, unpair] = qk
            kv = [self.debug_get_kv(layer, int(s), int(p), 1) for s, p in rows]
            return {"q": q, "k": np.concatenate([k for k, _ in kv]), "v": np.concatenate([v for _, v in kv])}
        if stage == 1:
            t = self._from_triples(self.debug_read(1), Q, M).reshape(M, 2, nh, 32)     # k tile = half * heads + head
            return {"attn": np.ascontiguousarray(t.transpose(0, 2, 1, 3)).reshape(M, Q)}
        if stage == 3:
            return {"act": self._from_triples(self.debug_read(2), c.intermediate_size, M)}
        fused_one = stage == 2 and M == 1 and self.debug_fused_o()
        return {"h": self.debug_read(8 if fused_one else 4).view(np.float32).reshape(M, c.hidden_size)[:M].copy()}

    def debug_fused_o(self) -> bool:
        """One live row takes the fused attention + o_proj kernel (smi_llm.hip: fuse_o_now) unless SPARKMI_NO_FUSE_O=1 was set
        when the engine was built or the head count has no fused instantiation."""
        import os
        return os.environ.get("SPARKMI_NO_FUSE_O") is None and self.cfg.num_attention_heads in (4, 14)

    def debug_sample(self, logits_row: Optional[np.ndarray], n_rows: int, seed: int, use_bound: bool = True) -> np.ndarray:
        """The device sampler alone on a caller's logits row (``smi_llm_debug_sample``): ``n_rows`` independent draws with
        the parameters of the last ``set_sampling``; ``logits_row`` None keeps the previous call's row."""
        self._need_diag("debug_sample")
        out = np.zeros(n_rows, dtype=np.int32)
        ptr = None
        if logits_row is not None:
            row = np.ascontiguousarray(logits_row, dtype=np.float32)
            assert row.shape == (self.cfg.vocab_size,)
            ptr = row.ctypes.data_as(C.POINTER(C.c_float))
        self._lib.check(self._lib.smi_llm_debug_sample(self._h, ptr, int(n_rows), C.c_uint64(int(seed)), 1 if use_bound else 0,
                                                       out.ctypes.data_as(C.POINTER(C.c_int32))), "smi_llm_debug_sample")
        return out

    KERNELS = ("qkv", "attn", "o_proj", "gate_up", "down", "lm_head", "finalize", "step", "layers")

    def time_kernel(self, name: str, iters: int = 48, layer: int = 0, in_sequence: bool = False) -> float:
        """Average milliseconds per launch of one decode-step kernel (HIP events on this stream).
        ``in_sequence`` times a layer kernel where it runs -- after its producers, which prefetch
        part of its weights into L2 -- as (iters layers) - (the same layers without it)."""
        self._need_diag("time_kernel")
        ms = C.c_float(0)
        self._lib.check(self._lib.smi_llm_time_kernel(self._h, self.KERNELS.index(name) + (16 if in_sequence else 0), layer, iters,
                                                 C.byref(ms), self._stream()), "smi_llm_time_kernel")
        return float(ms.value)

    # algorithmic bytes -------------------------------------------------------------------
    def weight_bytes(self) -> dict:
        c = self.cfg
        h, q, kv, i = c.hidden_size, c.q_dim, c.kv_dim, c.intermediate_size
        return {"qkv": 2 * (q + 2 * kv) * h, "o_proj": 2 * h * q, "gate_up": 2 * 2 * i * h, "down": 2 * h * i,
                "lm_head": 2 * c.vocab_size * h}

    def step_weight_bytes(self) -> int:
        w = self.weight_bytes()
        per_layer = w["qkv"] + w["o_proj"] + w["gate_up"] + w["down"]
        return self.cfg.num_hidden_layers * per_layer + w["lm_head"]

    def kv_bytes_per_token(self) -> int:
        esz = 4 if self._cs.kv_dtype else 2
        return self.cfg.num_hidden_layers * 2 * self.cfg.kv_dim * esz
