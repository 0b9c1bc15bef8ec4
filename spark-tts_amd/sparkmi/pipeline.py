"""SparkTTS -- drop-in for the reference pipeline class (``cli/SparkTTS.py``) on one MI355X.

Same constructor and ``inference()`` signature, same attributes (``device``, ``model_dir``,
``configs``, ``sample_rate``, ``tokenizer``, ``model``, ``audio_tokenizer``), same prompt
strings, same token parsing, same float32 numpy waveform.  The two heavy calls run on the HIP
kernels: ``self.model.generate`` (``SparkLLM``) and ``self.audio_tokenizer.detokenize``
(``BiCodecTokenizer``).  Keyword-only additions: ``do_sample``, ``max_new_tokens``,
``prompt_tokens`` (pre-computed prompt audio tokens), ``inference_batch`` and
``inference_stream`` (chunked vocoding while the LLM generates, the reference's decoupled Triton mode).
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Iterator, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib
from .bicodec import BiCodecTokenizer
from .config import LLMConfig, TopConfig
from .llm import SparkLLM, eos_ids_from_generation_config
from .pipeline_text import (GENDER_MAP, LEVELS_MAP, TASK_TOKEN_MAP, build_clone_prompt,
                            build_control_prompt, parse_global, parse_semantic)
from .streaming import ChunkScheduler
from .weights import load_llm_state


class _TokenMap:
    """id -> bicodec index tables read from the tokenizer's own vocabulary, used to skip the
    decode + regex round trip of ``cli/SparkTTS.py:213-228`` when it provably gives the same
    answer (every generated id is either a bicodec token or a special token)."""

    def __init__(self, tokenizer):
        import re
        self.sem: Dict[int, int] = {}
        self.glob: Dict[int, int] = {}
        rs, rg = re.compile(r"^<\|bicodec_semantic_(\d+)\|>$"), re.compile(r"^<\|bicodec_global_(\d+)\|>$")
        for tok, idx in tokenizer.get_vocab().items():
            m = rs.match(tok)
            if m:
                self.sem[idx] = int(m.group(1))
                continue
            m = rg.match(tok)
            if m:
                self.glob[idx] = int(m.group(1))
        self.special = set(int(i) for i in getattr(tokenizer, "all_special_ids", []) or [])
        # The reference decodes with skip_special_tokens=True (cli/SparkTTS.py:213): a bicodec token that a tokenizer lists
        # as SPECIAL would be dropped there before the regex sees it.  The id tables cannot reproduce that, so such a
        # tokenizer always takes the reference's decode + regex route.
        self.usable = not (self.special & (set(self.sem) | set(self.glob)))

    def fast_parse(self, ids: Sequence[int]) -> Optional[Tuple[List[int], List[int]]]:
        if not self.usable:
            return None
        sem, glob = [], []
        for i in ids:
            if i in self.sem:
                sem.append(self.sem[i])
            elif i in self.glob:
                glob.append(self.glob[i])
            elif i not in self.special:
                return None   # ordinary text token: fall back to the reference's decode + regex
        return sem, glob


class SparkTTS:
    """Spark-TTS for text-to-speech generation (MI355X-native hot path)."""

    def __init__(self, model_dir: Path, device: torch.device = torch.device("cuda:0"), *,
                 max_batch: int = 1, max_positions: int = 4096, kv_dtype: str = "bf16",
                 max_frames: int = 3000):
        self.device = torch.device(device)
        self.model_dir = model_dir
        top = TopConfig.from_yaml(Path(model_dir) / "config.yaml")
        self.configs = {"sample_rate": top.sample_rate, "ref_segment_duration": top.ref_segment_duration,
                        "latent_hop_length": top.latent_hop_length, "volume_normalize": top.volume_normalize}
        self.sample_rate = self.configs["sample_rate"]
        self._max_batch, self._max_positions, self._kv_dtype, self._max_frames = max_batch, max_positions, kv_dtype, max_frames
        self._initialize_inference()

    def _initialize_inference(self):
        """Tokenizer (HF, host side), LLM and audio tokenizer (both on the HIP kernels)."""
        from transformers import AutoTokenizer
        llm_dir = Path(self.model_dir) / "LLM"
        self.tokenizer = AutoTokenizer.from_pretrained(str(llm_dir))
        cfg = LLMConfig.from_json(llm_dir / "config.json")
        # HF generate() as the reference calls it (no eos argument, cli/SparkTTS.py:197-204) stops on EVERY id of
        # generation_config.json's eos_token_id; all of them go down to the step kernel
        self._eos = eos_ids_from_generation_config(llm_dir, cfg)
        if not self._eos and self.tokenizer.eos_token_id is not None:
            self._eos = [int(self.tokenizer.eos_token_id)]
        self.model = SparkLLM(cfg, load_llm_state(llm_dir), self.device, max_slots=self._max_batch,
                              max_positions=self._max_positions, kv_dtype=self._kv_dtype, eos_token_ids=self._eos)
        self.audio_tokenizer = BiCodecTokenizer(self.model_dir, device=self.device, max_batch=self._max_batch,
                                                max_frames=self._max_frames)
        self._map = _TokenMap(self.tokenizer)

    # ------------------------------------------------------------------ prompts
    def process_prompt(self, text: str, prompt_speech_path: Path, prompt_text: str = None,
                       prompt_tokens: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> Tuple[str, torch.Tensor]:
        """Voice-cloning prompt.  Returns (prompt string, global token ids (1, 1, Ntok))."""
        if prompt_tokens is not None:
            global_token_ids, semantic_token_ids = prompt_tokens
        else:
            global_token_ids, semantic_token_ids = self.audio_tokenizer.tokenize(prompt_speech_path)
        global_token_ids = torch.as_tensor(global_token_ids)
        semantic_token_ids = torch.as_tensor(semantic_token_ids)
        inputs = build_clone_prompt(text, global_token_ids.reshape(-1).tolist(),
                                    semantic_token_ids.reshape(-1).tolist(), prompt_text)
        return inputs, global_token_ids

    def process_prompt_control(self, gender: str, pitch: str, speed: str, text: str):
        """Voice-creation prompt (gender: female | male; pitch/speed: very_low .. very_high)."""
        return build_control_prompt(gender, pitch, speed, text)

    # ------------------------------------------------------------------ inference
    def _parse(self, new_ids: Sequence[int]) -> Tuple[List[int], List[int]]:
        fast = self._map.fast_parse(new_ids)
        if fast is not None:
            return fast
        predicts = self.tokenizer.batch_decode([list(new_ids)], skip_special_tokens=True)[0]
        return parse_semantic(predicts), parse_global(predicts)

    @torch.no_grad()
    def inference(self, text: str, prompt_speech_path: Path = None, prompt_text: str = None,
                  gender: str = None, pitch: str = None, speed: str = None,
                  temperature: float = 0.8, top_k: float = 50, top_p: float = 0.95, *,
                  do_sample: bool = True, max_new_tokens: int = 3000, seed: Optional[int] = None,
                  prompt_tokens: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> np.ndarray:
        """Text (+ optional prompt audio / style labels) -> float32 waveform at ``sample_rate``."""
        return self.inference_batch([dict(text=text, prompt_speech_path=prompt_speech_path, prompt_text=prompt_text,
                                          gender=gender, pitch=pitch, speed=speed, prompt_tokens=prompt_tokens)],
                                    temperature=temperature, top_k=top_k, top_p=top_p, do_sample=do_sample,
                                    max_new_tokens=max_new_tokens, seed=seed)[0]

    @torch.no_grad()
    def inference_batch(self, requests: Sequence[dict], temperature: float = 0.8, top_k: float = 50,
                        top_p: float = 0.95, *, do_sample: bool = True, max_new_tokens: int = 3000,
                        seed: Optional[int] = None) -> List[np.ndarray]:
        """Several independent utterances in one ragged batch (<= max_batch).  Greedy: each result equals the
        single-utterance call for that request -- exactly with an f32 KV cache; with the default bf16 cache up to near-tie
        arg-max flips between the prefill kernels the two call shapes select (include/sparkmi.h, smi_llm_session_begin)."""
        if len(requests) > self._max_batch:
            raise ValueError(f"{len(requests)} requests > max_batch={self._max_batch}")
        prompts, globals_ = [], []
        # voice-clone requests that come with prompt FILES: all their prompt encodes run side by side (parallel HIP streams)
        need = [i for i, r in enumerate(requests)
                if r.get("gender") is None and r.get("prompt_tokens") is None and r.get("prompt_speech_path") is not None]
        if len(need) > 1:
            toks = self.audio_tokenizer.tokenize_many([requests[i]["prompt_speech_path"] for i in need])
            requests = [dict(r) for r in requests]
            for i, t in zip(need, toks):
                requests[i]["prompt_tokens"] = t
        for r in requests:
            if r.get("gender") is not None:
                prompts.append(self.process_prompt_control(r["gender"], r.get("pitch"), r.get("speed"), r["text"]))
                globals_.append(None)
            else:
                p, g = self.process_prompt(r["text"], r.get("prompt_speech_path"), r.get("prompt_text"),
                                           r.get("prompt_tokens"))
                prompts.append(p)
                globals_.append(g)
        ids = [self.tokenizer([p], return_tensors="pt").input_ids[0].tolist() for p in prompts]
        # the reference's budget (3000) against a 32k-position model never binds; here the KV arena holds max_positions
        # tokens per sequence, so the budget shrinks with the prompt (as inference_stream and serve do) -- only a
        # prompt that itself does not fit is an error
        room = self._max_positions - max(len(i) for i in ids)
        if room < 1:
            raise ValueError(f"a prompt of {max(len(i) for i in ids)} tokens does not fit max_positions={self._max_positions}")
        max_new_tokens = min(int(max_new_tokens), room)
        if len(ids) > 1 and self._eos and len(ids) <= self.model.max_slots:
            # a batch: rows are retired at their own eos (SparkLLM.generate_ragged), so the step runs on the rows still
            # speaking instead of padding the finished ones to the longest utterance; same tokens per row
            self.model.set_sampling(bool(do_sample), temperature, int(top_k), float(top_p), seed)
            new = self.model.generate_ragged(ids, [max_new_tokens] * len(ids), self._eos)
        elif do_sample:
            new = self.model.generate_ids(ids, max_new_tokens, self._eos, do_sample=True, temperature=temperature,
                                          top_k=int(top_k), top_p=float(top_p), seed=seed)
        else:
            new = self.model.generate_ids(ids, max_new_tokens, self._eos)
        sems, globs, lens = [], [], []
        ntok = self.audio_tokenizer.model.cfg.spk_token_num
        for b, toks in enumerate(new):
            sem, glob = self._parse(toks)
            if globals_[b] is None:
                g = torch.tensor(glob, dtype=torch.long)
            else:
                g = torch.as_tensor(globals_[b]).reshape(-1).long()
            if g.numel() != ntok:
                raise ValueError(f"request {b}: {g.numel()} global tokens generated, the speaker encoder needs {ntok}")
            if not sem:
                raise ValueError(f"request {b}: the model generated no semantic tokens")
            sems.append(sem)
            globs.append(g)
            lens.append(len(sem))
        T = max(lens)
        sem_t = torch.zeros((len(sems), T), dtype=torch.long)
        for b, s in enumerate(sems):
            sem_t[b, : len(s)] = torch.tensor(s)
        wav = self.audio_tokenizer.model.detokenize(sem_t, torch.stack(globs).unsqueeze(1), lengths=lens)
        wav = wav.squeeze(1).cpu().numpy()
        hop = self.audio_tokenizer.model.hop
        return [wav[b, : lens[b] * hop].copy() for b in range(len(sems))]

    @torch.no_grad()
    def inference_stream(self, text: str, prompt_speech_path: Path = None, prompt_text: str = None,
                         gender: str = None, pitch: str = None, speed: str = None,
                         temperature: float = 0.8, top_k: float = 50, top_p: float = 0.95, *,
                         do_sample: bool = True, max_new_tokens: int = 3000, seed: Optional[int] = None,
                         prompt_tokens: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                         audio_chunk_duration: float = 1.0, max_audio_chunk_duration: float = 30.0,
                         audio_chunk_size_scale_factor: float = 8.0, audio_chunk_overlap_duration: float = 0.1,
                         decode_stride: int = 10) -> Iterator[np.ndarray]:
        """Yields float32 waveform chunks while the LLM is still generating, cut and overlapped as
        the reference's decoupled Triton model does (model.py:347-385; run.sh:53-56 defaults); join
        them with ``sparkmi.streaming.crossfade(chunks, int(overlap_duration * sample_rate))``
        (client_grpc.py:390-415).  Every chunk is the vocoder's output for that chunk's tokens alone.
        ``decode_stride`` = decode steps enqueued between host checks for new tokens."""
        if gender is not None:
            prompt, glob = self.process_prompt_control(gender, pitch, speed, text), None
        else:
            prompt, g = self.process_prompt(text, prompt_speech_path, prompt_text, prompt_tokens)
            glob = torch.as_tensor(g).reshape(-1).long()
        ids = self.tokenizer([prompt], return_tensors="pt").input_ids[0].tolist()
        if len(ids) + max_new_tokens > self._max_positions:
            max_new_tokens = self._max_positions - len(ids)
        voc = self.audio_tokenizer.model
        ntok, hop = voc.cfg.spk_token_num, voc.hop
        frame_rate = self.sample_rate // hop
        sched = ChunkScheduler(audio_chunk_duration, max_audio_chunk_duration, audio_chunk_size_scale_factor,
                               audio_chunk_overlap_duration, frame_rate)
        self.model.set_sampling(do_sample, temperature, int(top_k), float(top_p), seed)
        self.model.prefill([ids], self._eos)
        produced, n_sem, pending = 1, 0, []

        def vocode(chunk: List[int]) -> np.ndarray:
            wav = voc.detokenize(torch.tensor([chunk], dtype=torch.long), glob.reshape(1, 1, -1), lengths=[len(chunk)])
            return wav.reshape(-1)[: len(chunk) * hop].cpu().numpy().copy()

        while True:
            toks = self.model.tokens(max_new_tokens)[0]
            sem, gl = self._parse(toks)
            if glob is None and len(gl) >= ntok:        # voice creation: the speaker tokens are generated first
                glob = torch.tensor(gl[:ntok], dtype=torch.long)
            pending += sched.push(sem[n_sem:])
            n_sem = len(sem)
            done = produced >= max_new_tokens or self.model.all_done()
            if done:
                pending += sched.flush()
            if glob is not None:
                for chunk in pending:
                    yield vocode(chunk)
                pending = []
            if done:
                break
            n = min(decode_stride, max_new_tokens - produced)
            self.model.decode(n)
            produced += n
        if glob is None:
            raise ValueError(f"{ntok} global tokens were not generated; the speaker encoder needs them")

    @torch.no_grad()
    def serve(self, requests, temperature: float = 0.8, top_k: float = 50, top_p: float = 0.95, *, do_sample: bool = True,
              max_new_tokens: int = 3000, seed: Optional[int] = None, decode_stride: int = 8):
        """In-flight batching front end (the functional analogue of the reference's Triton deployment,
        runtime/triton_trtllm/run.sh:50-65): ``requests`` is an iterable of the dicts ``inference_batch`` takes;
        yields ``(index, waveform)`` as each utterance finishes.  Up to ``max_batch`` utterances are live; a new
        request is admitted into the LLM's free KV slot as soon as one retires, so short utterances do not wait for
        long ones.  Greedy results equal ``inference()`` of the same request."""
        voc = self.audio_tokenizer.model
        ntok, hop = voc.cfg.spk_token_num, voc.hop
        globals_: Dict[int, Optional[torch.Tensor]] = {}
        self.model.set_sampling(do_sample, temperature, int(top_k), float(top_p), seed)

        def llm_requests():
            for i, r in enumerate(requests):
                if r.get("gender") is not None:
                    prompt, g = self.process_prompt_control(r["gender"], r.get("pitch"), r.get("speed"), r["text"]), None
                else:
                    prompt, g = self.process_prompt(r["text"], r.get("prompt_speech_path"), r.get("prompt_text"), r.get("prompt_tokens"))
                globals_[i] = g
                ids = self.tokenizer([prompt], return_tensors="pt").input_ids[0].tolist()
                yield i, ids, min(max_new_tokens, self._max_positions - len(ids) - decode_stride), self._eos

        for i, toks in self.model.serve(llm_requests(), max_live=self._max_batch, decode_stride=decode_stride):
            stops = [toks.index(e) for e in self._eos if e in toks]
            if stops:
                toks = toks[: min(stops) + 1]
            sem, glob = self._parse(toks)
            g = torch.tensor(glob, dtype=torch.long) if globals_[i] is None else torch.as_tensor(globals_[i]).reshape(-1).long()
            if g.numel() != ntok:
                raise ValueError(f"request {i}: {g.numel()} global tokens, the speaker encoder needs {ntok}")
            if not sem:
                raise ValueError(f"request {i}: the model generated no semantic tokens")
            wav = voc.detokenize(torch.tensor([sem], dtype=torch.long), g.reshape(1, 1, -1), lengths=[len(sem)])
            yield i, wav.reshape(-1)[: len(sem) * hop].cpu().numpy().copy()
