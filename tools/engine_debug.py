#!/usr/bin/env python3
"""Stage-by-stage comparison of the one-row engine with the launch path on a ONE-layer tiny model (GPU): q|k|v, attention
output, h + o_proj, act, final h -- finds the first stage that differs."""
import os, sys
os.environ["SPARKMI_NO_FUSE_O"] = "1"     # the launch path then leaves every stage in a scratch buffer (same bits)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "spark-tts_amd"))
import numpy as np
from sparkmi import config as C, weights as W
from sparkmi.llm import SparkLLM

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = C.tiny_llm(layers=layers)
syn = W.SyntheticLLM(cfg)
llm = SparkLLM(cfg, syn, "cuda:0", max_positions=320, use_graph=False, diag=True)
llm.set_engine(True)
print(llm.engine_info())
H, Q, KV, I = cfg.hidden_size, cfg.q_dim, cfg.kv_dim, cfg.intermediate_size
prompt = np.random.Generator(np.random.PCG64(11)).integers(0, cfg.vocab_size, size=37).tolist()


def triples(raw, K):
    """[K/32][3][4][16 B] bf16 pieces -> f32 values (hi + mid + lo, exact)."""
    a = raw.view(np.uint16).reshape(K // 32, 3, 4, 8).astype(np.uint32) << 16
    f = a.view(np.float32)
    return (f[:, 0] + f[:, 1] + f[:, 2]).reshape(K)


def run(engine):
    llm.set_engine(engine)
    llm.prefill([prompt])           # its last step (first token) runs the one-row path under test
    out = {"h": llm.debug_read(4).view(np.float32).copy(), "tok": llm.tokens(1)[0]}
    if engine:
        g = llm.debug_read(5).view(np.uint64).reshape(2, -1)
        b = g[(layers - 1) & 1]
        val = (b & np.uint64(0xffffffff)).astype(np.uint32).view(np.float32)
        tag = (b >> np.uint64(32)).astype(np.uint32)
        o = 0
        for name, n in (("A", H), ("qkv", Q + 2 * KV), ("attn", Q), ("hmid", H), ("act", I)):
            out[name] = val[o:o + n].copy(); out[name + "_tag"] = tag[o:o + n].copy(); o += n
    out["k0"] = llm.debug_read(7).view(np.uint16)[: 64 * (len(prompt) + 1)].copy()
    if not engine:
        out["q"] = llm.debug_read(0).view(np.float32).copy()
        out["attn"] = triples(llm.debug_read(1), Q)
        out["act"] = triples(llm.debug_read(2), I)
    return out


e = run(True)
r = run(False)
print("tokens", e["tok"], r["tok"])
for name in ("qkv", "attn", "hmid", "act"):
    print(name, "tags", np.unique(e[name + "_tag"]))


def cmp(name, a, b):
    bad = np.flatnonzero(a.view(np.uint32) != b.view(np.uint32))
    print(f"{name:8s} n={a.size:5d} differing {bad.size:5d}  max|d| {np.abs(a - b).max():.3e}  first bad {bad[:12].tolist()}")
    if bad.size:
        i = bad[0]
        print("          engine", a[i:i + 4], "launch", b[i:i + 4])


cmp("q", e["qkv"][:Q], r["q"])
# attention output: the launch path's operand is stored head-interleaved; element (head, d) -> k tile (d >> 5) * heads + head
nh = cfg.num_attention_heads
perm = np.array([((d >> 5) * nh + h) * 32 + (d & 31) for h in range(nh) for d in range(64)])
cmp("attn", e["attn"], r["attn"][perm])
cmp("act", e["act"], r["act"])
kd = np.flatnonzero(e["k0"] != r["k0"])
print("layer-0 K rows (slot 0, head 0): differing elements", kd.size, "first at token", (kd[0] // 64 if kd.size else None))
