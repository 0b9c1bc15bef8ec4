"""ORACLE (test infrastructure, never the product path): CPU restatement of the Qwen2
causal-LM forward and HF greedy ``generate`` that ``cli/SparkTTS.py:197-204`` calls.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The arithmetic lives in a third-party dependency of the reference
(``transformers==4.46.2`` pinned at ``requirements.txt:12``; 5.15.0 is installed in this image),
not in ``/root/reference``.  Each function cites the ``modeling_qwen2.py`` lines (abbrev. ``MQ``,
transformers 5.15.0) it restates.  Pinning: ``tests/test_oracle_llm.py`` checks this file against
``transformers.Qwen2ForCausalLM`` itself (logits, prefill+decode cache path, greedy tokens) and
against the committed fixtures in ``tests/golden/``.

Plain fp32 torch-CPU ops, B=1 per call (the reference never batches the LM:
``cli/SparkTTS.py:194`` tokenises a single prompt).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F


def _bf16_round(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


class Qwen2Ref:
    """Weights are a ``name -> fp32 ndarray/tensor`` mapping under HF parameter names."""

    def __init__(self, cfg, weights, kv_dtype: str = "f32"):
        self.cfg = cfg
        self.kv_dtype = kv_dtype  # "f32" (reference behaviour) or "bf16" (emulates the GPU KV cache)
        g = lambda n: torch.as_tensor(np.asarray(weights[n]), dtype=torch.float32)  # noqa: E731
        self.embed = g("model.embed_tokens.weight")
        self.lm_head = self.embed if cfg.tie_word_embeddings else g("lm_head.weight")
        self.final_norm = g("model.norm.weight")
        self.layers = []
        for i in range(cfg.num_hidden_layers):
            p = f"model.layers.{i}."
            self.layers.append(dict(
                ln1=g(p + "input_layernorm.weight"), ln2=g(p + "post_attention_layernorm.weight"),
                wq=g(p + "self_attn.q_proj.weight"), bq=g(p + "self_attn.q_proj.bias"),
                wk=g(p + "self_attn.k_proj.weight"), bk=g(p + "self_attn.k_proj.bias"),
                wv=g(p + "self_attn.v_proj.weight"), bv=g(p + "self_attn.v_proj.bias"),
                wo=g(p + "self_attn.o_proj.weight"),
                wg=g(p + "mlp.gate_proj.weight"), wu=g(p + "mlp.up_proj.weight"),
                wd=g(p + "mlp.down_proj.weight"),
            ))
        d = cfg.head_dim
        # MQ:91-93  inv_freq = 1 / theta^(2i/d)
        self.inv_freq = 1.0 / (cfg.rope_theta ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
        self.reset()

    # ------------------------------------------------------------------ pieces
    def reset(self) -> None:
        self.k_cache: List[Optional[torch.Tensor]] = [None] * self.cfg.num_hidden_layers
        self.v_cache: List[Optional[torch.Tensor]] = [None] * self.cfg.num_hidden_layers
        self.pos = 0

    def rmsnorm(self, x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        """MQ:247-252: x * rsqrt(mean(x^2) + eps) in fp32, then * weight."""
        var = x.pow(2).mean(-1, keepdim=True)
        return w * (x * torch.rsqrt(var + self.cfg.rms_norm_eps))

    def rope(self, x: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
        """MQ:95-102 + MQ:105-133: x*cos + rotate_half(x)*sin on the full head_dim.
        x: (S, H, D); positions: (S,)."""
        freqs = positions.to(torch.float32)[:, None] * self.inv_freq[None, :]
        emb = torch.cat((freqs, freqs), dim=-1)
        cos, sin = emb.cos()[:, None, :], emb.sin()[:, None, :]
        h = x.shape[-1] // 2
        rot = torch.cat((-x[..., h:], x[..., :h]), dim=-1)
        return x * cos + rot * sin

    def attention(self, q, k, v, q_pos: torch.Tensor) -> torch.Tensor:
        """MQ:150-173 eager path: repeat_kv, softmax(q k^T * d^-0.5 + causal mask) in fp32, @ v.
        q: (S, Hq, D); k, v: (T, Hkv, D) full cache; q_pos: absolute positions (S,)."""
        cfg = self.cfg
        rep = cfg.num_attention_heads // cfg.num_key_value_heads
        kk = k.repeat_interleave(rep, dim=1)  # (T, Hq, D)  == repeat_kv (MQ:138-147)
        vv = v.repeat_interleave(rep, dim=1)
        scores = torch.einsum("shd,thd->hst", q, kk) * (cfg.head_dim ** -0.5)
        t_idx = torch.arange(k.shape[0])
        mask = t_idx[None, :] > q_pos[:, None]  # key position beyond the query's -> masked
        scores = scores.masked_fill(mask[None], float("-inf"))
        p = torch.softmax(scores, dim=-1, dtype=torch.float32)
        return torch.einsum("hst,thd->shd", p, vv).reshape(q.shape[0], -1)

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, ids: Sequence[int], last_only: bool = False,
                return_hidden: bool = False):
        """Append ``ids`` at positions pos..pos+S-1 (KV cache kept) and return logits
        (S, V) (or (1, V) when ``last_only``).  MQ:342-470."""
        cfg = self.cfg
        ids_t = torch.as_tensor(np.asarray(ids), dtype=torch.long)
        S = ids_t.shape[0]
        positions = torch.arange(self.pos, self.pos + S)
        h = self.embed[ids_t]
        hiddens = []
        for li, L in enumerate(self.layers):
            x = self.rmsnorm(h, L["ln1"])
            q = F.linear(x, L["wq"], L["bq"]).view(S, cfg.num_attention_heads, cfg.head_dim)
            k = F.linear(x, L["wk"], L["bk"]).view(S, cfg.num_key_value_heads, cfg.head_dim)
            v = F.linear(x, L["wv"], L["bv"]).view(S, cfg.num_key_value_heads, cfg.head_dim)
            q, k = self.rope(q, positions), self.rope(k, positions)
            if self.kv_dtype == "bf16":
                k, v = _bf16_round(k), _bf16_round(v)
            if self.k_cache[li] is not None:
                k = torch.cat((self.k_cache[li], k), dim=0)
                v = torch.cat((self.v_cache[li], v), dim=0)
            self.k_cache[li], self.v_cache[li] = k, v
            a = self.attention(q, k, v, positions)
            h = h + F.linear(a, L["wo"])
            x = self.rmsnorm(h, L["ln2"])
            # MQ:46-48  down( silu(gate x) * up x )
            h = h + F.linear(F.silu(F.linear(x, L["wg"])) * F.linear(x, L["wu"]), L["wd"])
            if return_hidden:
                hiddens.append(h.clone())
        self.pos += S
        hn = self.rmsnorm(h[-1:] if last_only else h, self.final_norm)
        logits = F.linear(hn, self.lm_head)
        return (logits, hiddens) if return_hidden else logits

    @torch.no_grad()
    def generate_greedy(self, prompt_ids: Sequence[int], max_new_tokens: int,
                        eos_ids: Sequence[int] = ()) -> List[int]:
        """HF ``generate(do_sample=False)`` for one sequence: prefill emits token 1, each
        decode forward emits the next; stops after emitting an EOS id or at
        ``max_new_tokens``.  Returns only the new tokens (the slice the reference takes at
        ``cli/SparkTTS.py:207-210``).  ``torch.argmax`` breaks ties toward the lowest index."""
        self.reset()
        eos = set(int(e) for e in eos_ids)
        logits = self.forward(prompt_ids, last_only=True)
        out: List[int] = []
        for _ in range(max_new_tokens):
            tok = int(torch.argmax(logits[-1]).item())
            out.append(tok)
            if tok in eos or len(out) == max_new_tokens:
                break
            logits = self.forward([tok], last_only=True)
        return out
