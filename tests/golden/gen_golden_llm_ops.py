#!/usr/bin/env python3
"""Generates ``tests/golden/llm_ops.npz``: one Qwen2 decoder layer taken apart, from transformers' OWN classes.

The LLM arithmetic of the reference is third-party (``AutoModelForCausalLM`` at ``cli/SparkTTS.py:49,197``;
``transformers/models/qwen2/modeling_qwen2.py``, abbreviated MQ).  ``ops_layers.npz`` already holds RMSNorm / RoPE /
attention / MLP vectors at head_dim 32, which the oracle is checked against; the HIP kernels are built for head_dim 64
(``smi_llm_cfg.head_dim``), so this file repeats them at the tiny test shape (hidden 256, 4 query / 2 kv heads x 64,
intermediate 608) with bf16-representable weights, and ``tests/test_llm_ops_gpu.py`` runs the step's own launch builders
on them one stage at a time (``smi_llm_debug_layer``).  Imported classes, not restatements:

  stage 0  MQ.Qwen2RMSNorm -> Qwen2Attention.q_proj / k_proj / v_proj (nn.Linear with bias) -> MQ.apply_rotary_pos_emb
  stage 1  MQ.eager_attention_forward (GQA: repeat_kv, softmax in fp32) of every row against ITS OWN cached keys
  stage 2  Qwen2Attention.o_proj + residual
  stage 3  MQ.Qwen2RMSNorm -> act_fn(gate_proj(x)) * up_proj(x)      (the inside of MQ.Qwen2MLP.forward)
  stage 4  MQ.Qwen2MLP (whole) + residual

    python tests/golden/gen_golden_llm_ops.py
"""
from __future__ import annotations

import os

import numpy as np
import torch
from transformers import Qwen2Config
from transformers.models.qwen2 import modeling_qwen2 as MQ

HERE = os.path.dirname(os.path.abspath(__file__))


def bf16r(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


def main() -> None:
    g = torch.Generator().manual_seed(4242)
    H, NH, NKV, D, I = 256, 4, 2, 64, 608
    hc = Qwen2Config(vocab_size=64, hidden_size=H, intermediate_size=I, num_hidden_layers=1, num_attention_heads=NH,
                     num_key_value_heads=NKV, rms_norm_eps=1e-6, rope_theta=1000000.0, max_position_embeddings=512,
                     attn_implementation="eager")
    attn = MQ.Qwen2Attention(hc, layer_idx=0).eval()
    mlp = MQ.Qwen2MLP(hc).eval()
    ln1, ln2 = MQ.Qwen2RMSNorm(H, eps=1e-6), MQ.Qwen2RMSNorm(H, eps=1e-6)
    rot = MQ.Qwen2RotaryEmbedding(config=hc)
    with torch.no_grad():
        for mod in (attn, mlp):
            for n, p in mod.named_parameters():
                r = torch.randn(p.shape, generator=g)
                p.copy_(0.3 * r if n.endswith("bias") else bf16r(r * (1.6 / np.sqrt(p.shape[-1]))))   # matrices bf16-exact (the arena's dtype)
        for ln in (ln1, ln2):
            ln.weight.copy_(1.0 + 0.2 * torch.randn(H, generator=g))
    # rows: (KV slot, position).  Slots 0..2 hold cached contexts of 9 / 130 / 300 tokens (the last crosses the attention
    # kernel's 256-token chunk); five rows: three single decode rows and two consecutive rows of slot 1 (a prefill chunk)
    ctx = {0: 9, 1: 130, 2: 300}
    rows = [(0, 9), (1, 130), (1, 131), (2, 300), (0, 10)]
    M = len(rows)
    x = torch.randn(M, H, generator=g)
    kc = {s: torch.randn(NKV, n, D, generator=g) for s, n in ctx.items()}     # cached keys, ALREADY rotated (as a cache holds them)
    vc = {s: torch.randn(NKV, n, D, generator=g) for s, n in ctx.items()}
    out = {"rows": np.array(rows, dtype=np.int32), "x": x.numpy()}
    for s in ctx:
        out[f"kcache{s}"], out[f"vcache{s}"] = kc[s].numpy(), vc[s].numpy()
    with torch.no_grad():
        pos = torch.tensor([[p for _, p in rows]])
        xn = ln1(x[None])                                                       # (1, M, H)
        q = attn.q_proj(xn).view(1, M, NH, D).transpose(1, 2)                  # (1, NH, M, D)
        k = attn.k_proj(xn).view(1, M, NKV, D).transpose(1, 2)
        v = attn.v_proj(xn).view(1, M, NKV, D).transpose(1, 2)
        cos, sin = rot(q, pos)
        qr, kr = MQ.apply_rotary_pos_emb(q, k, cos, sin)
        out["q_rot"], out["k_rot"], out["v"] = qr[0].transpose(0, 1).numpy(), kr[0].transpose(0, 1).numpy(), v[0].transpose(0, 1).numpy()   # (M, heads, D)
        # stage 1: every row attends to its slot's cached keys plus the rows of that slot up to itself (causal), one call per row
        ao = torch.zeros(M, NH * D)
        for i, (s, p) in enumerate(rows):
            ks, vs = [kc[s]], [vc[s]]
            for j, (s2, p2) in enumerate(rows):
                if s2 == s and p2 <= p:
                    ks.append(kr[0, :, j: j + 1]); vs.append(v[0, :, j: j + 1])
            K, V = torch.cat(ks, dim=1)[None], torch.cat(vs, dim=1)[None]
            assert K.shape[2] == p + 1
            o, _ = MQ.eager_attention_forward(attn, qr[:, :, i: i + 1], K, V, attention_mask=None, scaling=D ** -0.5, dropout=0.0)
            ao[i] = o.reshape(-1)                                                # (1, 1, NH, D) -> NH * D
        out["attn_out"] = ao.numpy()
        h_mid = x + attn.o_proj(ao)
        out["h_mid"] = h_mid.numpy()
        xn2 = ln2(h_mid)
        out["act"] = (mlp.act_fn(mlp.gate_proj(xn2)) * mlp.up_proj(xn2)).numpy()
        out["h_out"] = (h_mid + mlp(xn2)).numpy()
    for prefix, mod in (("attn", attn), ("mlp", mlp)):
        for k2, v2 in mod.state_dict().items():
            out[f"{prefix}/{k2}"] = v2.numpy().copy()
    out["ln1"], out["ln2"] = ln1.weight.detach().numpy().copy(), ln2.weight.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, "llm_ops.npz"), **out)
    print(f"[llm_ops] {len(out)} arrays; rows {rows}; |q| {float(qr.abs().mean()):.3f} |attn| {float(ao.abs().mean()):.3f} "
          f"|act| {float(torch.from_numpy(out['act']).abs().mean()):.3f}")


if __name__ == "__main__":
    main()
