"""GPU parity of the HIP LLM path (through the C ABI) against the CPU oracle and the committed
golden vectors.  Tolerances: the kernels multiply exactly (bf16 weights x exactly-split fp32
activations) and accumulate in fp32, so logits agree with the fp32 oracle to summation-order
noise; 2e-3 absolute on logits of std ~1.6 is the bound written here (observed ~1e-5)."""
import os

import numpy as np
import pytest
import torch

from oracle.llm_ref import Qwen2Ref
from sparkmi import config as C, weights as W

pytestmark = pytest.mark.gpu

LOGIT_ATOL = 2e-3


def _wants_diag():
    """A test that sets a SPARKMI_* switch is exercising an A/B path: those exist in the diagnostics build only (the product
    library reads no environment variable), so its engines go on libsparkmi_diag.so -- same sources, same kernels."""
    return any(k.startswith("SPARKMI_") for k in os.environ)


def _llm(cfg, syn, **kw):
    from sparkmi.llm import SparkLLM
    kw.setdefault("diag", _wants_diag())
    return SparkLLM(cfg, syn, device="cuda:0", **kw)


@pytest.fixture(scope="module")
def tiny():
    cfg = C.tiny_llm()
    return cfg, W.SyntheticLLM(cfg)


@pytest.mark.parametrize("kv", ["f32", "bf16"])
def test_teacher_forced_logits_match_oracle(tiny, kv):
    cfg, syn = tiny
    ids = np.random.Generator(np.random.PCG64(7)).integers(0, cfg.vocab_size, size=45)  # 2 prefill chunks
    llm = _llm(cfg, syn, max_positions=128, kv_dtype=kv)
    got = llm.forward_logits(ids).cpu().numpy()
    ref = Qwen2Ref(cfg, syn, kv_dtype=kv).forward(ids).numpy()
    assert got.shape == ref.shape
    err = np.abs(got - ref).max()
    assert err < LOGIT_ATOL, f"max |logit diff| {err}"
    assert (got.argmax(-1) == ref.argmax(-1)).all()


def test_logits_match_golden_vectors(tiny, golden_dir):
    cfg, syn = tiny
    g = np.load(os.path.join(golden_dir, "llm_tiny.npz"))
    llm = _llm(cfg, syn, max_positions=128, kv_dtype="f32")
    got = llm.forward_logits(g["prompt"]).cpu().numpy()
    assert np.abs(got - g["logits"]).max() < LOGIT_ATOL


@pytest.mark.parametrize("kv,graph", [("f32", True), ("bf16", True), ("bf16", False)])
def test_greedy_tokens_match_oracle_and_golden(tiny, golden_dir, kv, graph):
    cfg, syn = tiny
    g = np.load(os.path.join(golden_dir, "llm_tiny.npz"))
    n = len(g["greedy"])
    llm = _llm(cfg, syn, max_positions=128, kv_dtype=kv, use_graph=graph)
    got = llm.generate_ids([g["prompt"].tolist()], n)[0]
    ref = Qwen2Ref(cfg, syn, kv_dtype=kv).generate_greedy(g["prompt"], n)
    assert got == ref
    if kv == "f32":
        assert got == g["greedy"].tolist()   # what transformers' generate() produced


def test_hf_shaped_generate_and_eos(tiny):
    cfg, syn = tiny
    prompt = [11, 22, 33, 44, 55]
    llm = _llm(cfg, syn, max_positions=128, kv_dtype="f32")
    free = llm.generate_ids([prompt], 20)[0]
    eos = free[6]
    ids = torch.tensor([prompt])
    out = llm.generate(ids, attention_mask=torch.ones_like(ids), max_new_tokens=20, do_sample=False, eos_token_id=eos)
    new = out[0, len(prompt):].tolist()
    stop = free.index(eos) + 1
    assert new[:stop] == free[:stop] and len(new) == stop
    assert Qwen2Ref(cfg, syn).generate_greedy(prompt, 20, eos_ids=[eos]) == new


def _tv_draws(support: int) -> int:
    """Draws that put the EXPECTED total-variation distance of an exact sampler (~ sqrt(S / (2 pi n)) for S comparable
    probabilities) near 0.012, so that the 0.03 bar below is a test of the sampler and not of the sample size; >= 20 480."""
    return max(20480, 1200 * support)


def _check_draws(counts, want, what):
    n = counts.sum()
    assert (counts[want == 0] == 0).all(), f"{what}: sampled a token outside the top-k / nucleus set"
    tv = 0.5 * np.abs(counts / n - want).sum()
    assert tv < 0.03, f"{what}: total variation distance {tv:.4f} over {int(n)} draws"
    return tv


def test_sampling_distribution_matches_the_warper_chain(tiny):
    """End to end (prefill -> lm_head -> sampler) at the reference's default parameters (cli/SparkTTS.py:166-168): the empirical
    distribution of >= 40 000 first tokens against oracle/sampling_ref.py, which tests/test_oracle_sampling.py pins to
    transformers' own warper classes."""
    from oracle.sampling_ref import sampling_probs
    cfg, syn = tiny
    prompt = [5, 17, 200, 33, 9, 410, 77]
    T, K, P = 0.8, 50, 0.95
    logits = Qwen2Ref(cfg, syn, kv_dtype="bf16").forward(prompt, last_only=True)[0]
    want = sampling_probs(logits, T, K, P).numpy()
    llm = _llm(cfg, syn, max_slots=64, max_positions=64)
    counts = np.zeros(cfg.vocab_size)
    for seed in range(40960 // 64):
        for t in llm.generate_ids([prompt] * 64, 1, do_sample=True, temperature=T, top_k=K, top_p=P, seed=seed):
            counts[t[0]] += 1
    _check_draws(counts, want, "tiny model, T 0.8 / k 50 / p 0.95")
    assert (counts > 0).sum() > 5


def _fixture_row(g, name):
    if name != "big":
        return g[f"{name}.logits"]
    seed, v = (int(x) for x in g["big.seed"])
    return (np.random.Generator(np.random.PCG64(seed)).standard_normal(v) * float(g["big.std"])).astype(np.float32)


@pytest.mark.parametrize("name", ["tiny", "tie", "edge"])
def test_sampler_on_the_transformers_fixture_rows(golden_dir, name):
    """The device sampler alone (smi_llm_debug_sample: k_sample_scan + k_sample as a step launches them) on the rows of
    tests/golden/sampling.npz, against what transformers' OWN warpers keep there (the fixture itself, not a restatement):
    a tie at the k-th value (more than k survivors), nucleus cuts next to a cumulative probability, top_k = 1."""
    g = np.load(os.path.join(golden_dir, "sampling.npz"))
    row = _fixture_row(g, name)
    cfg = C.tiny_llm()
    if cfg.vocab_size != len(row):
        import dataclasses
        cfg = dataclasses.replace(cfg, vocab_size=len(row))
    llm = _llm(cfg, W.SyntheticLLM(cfg), max_slots=64, max_positions=32, diag=True)
    for i, (t, k, p) in enumerate(g[f"{name}.params"]):
        want = np.zeros(len(row))
        want[g[f"{name}.{i}.ids"]] = g[f"{name}.{i}.probs"]
        llm.set_sampling(True, float(t), int(k), float(p), 0)
        counts = np.zeros(len(row))
        draws = _tv_draws(int((want > 0).sum()))
        for seed in range((draws + 63) // 64):
            np.add.at(counts, llm.debug_sample(row if seed == 0 else None, 64, 1000 + seed), 1)
        _check_draws(counts, want, f"{name}[{i}] T {t} k {k} p {p:.4f}")
        if name == "tie" and p == 1.0:
            assert int((want > 0).sum()) == 51 and (counts[want > 0] > 0).all(), "every token that ties with the k-th value must be drawable"


def test_sampling_is_seeded_and_top_k_1_is_greedy(tiny):
    cfg, syn = tiny
    prompt = [3, 1, 4, 1, 5, 9, 2, 6]
    llm = _llm(cfg, syn, max_positions=96)
    greedy = llm.generate_ids([prompt], 24)[0]
    assert llm.generate_ids([prompt], 24, do_sample=True, top_k=1, seed=9)[0] == greedy
    a = llm.generate_ids([prompt], 24, do_sample=True, seed=1234)[0]
    b = llm.generate_ids([prompt], 24, do_sample=True, seed=1234)[0]
    c = llm.generate_ids([prompt], 24, do_sample=True, seed=4321)[0]
    assert a == b and a != c and a != greedy
    assert llm.generate_ids([prompt], 24)[0] == greedy     # switching back to greedy works
    from sparkmi._lib import SparkMIError
    with pytest.raises(SparkMIError):
        llm.generate_ids([prompt], 4, do_sample=True, top_k=1000)


def test_ragged_batch_equals_single_sequence_runs(tiny):
    """B ragged sequences in one batch produce bit-identical tokens to B separate B=1 runs."""
    cfg, syn = tiny
    rng = np.random.Generator(np.random.PCG64(21))
    B = 19   # > 16 exercises the two-m-tile MFMA path
    prompts = [rng.integers(0, cfg.vocab_size, size=int(rng.integers(1, 40))).tolist() for _ in range(B)]
    batched = _llm(cfg, syn, max_slots=B, max_positions=96, kv_dtype="bf16").generate_ids(prompts, 24)
    single = _llm(cfg, syn, max_slots=1, max_positions=96, kv_dtype="bf16")
    for b in range(B):
        assert single.generate_ids([prompts[b]], 24)[0] == batched[b], f"sequence {b}"
    # and the oracle agrees on a few of them
    for b in (0, 7, 18):
        assert Qwen2Ref(cfg, syn, kv_dtype="bf16").generate_greedy(prompts[b], 24) == batched[b]


def test_run_twice_is_bitwise_deterministic(tiny):
    cfg, syn = tiny
    ids = np.arange(1, 40)
    llm = _llm(cfg, syn, max_positions=128)
    a = llm.forward_logits(ids).clone()
    b = llm.forward_logits(ids)
    assert torch.equal(a, b)


def test_bad_arguments_are_reported(tiny):
    from sparkmi._lib import SparkMIError
    cfg, syn = tiny
    llm = _llm(cfg, syn, max_positions=64)
    with pytest.raises(SparkMIError):
        llm.prefill([[cfg.vocab_size + 5]])
    with pytest.raises(ValueError):
        llm.generate_ids([[1, 2, 3]], 100)
    with pytest.raises(SparkMIError):
        _llm(cfg, syn, max_positions=64).decode(1)   # decode before prefill


def test_full_size_0p5b_against_transformers_golden(golden_dir, full_llm, full_llm_oracle):
    """Spark-TTS-0.5B shape (24 layers, vocab 166000), synthetic weights: last-position logits and
    the 150 greedy tokens transformers produced in the build container."""
    from conftest import FULL_MAX_POS
    cfg, syn, arena = full_llm
    g = np.load(os.path.join(golden_dir, "llm_full.npz"))
    llm = _llm(cfg, None, max_positions=FULL_MAX_POS, kv_dtype="f32", arena=arena)
    logits = llm.forward_logits(g["prompt"])[-1].cpu().numpy()
    np.testing.assert_allclose(logits[g["last_top_ids"]], g["last_top_vals"], rtol=0, atol=LOGIT_ATOL)
    assert abs(float(np.abs(logits.astype(np.float64)).sum()) - float(g["last_logits_abs"])) < 1e-4 * float(g["last_logits_abs"])
    got = llm.generate_ids([g["prompt"].tolist()], 150)[0]
    assert got == g["greedy"].tolist()
    # production setting (bf16 KV): teacher-forced on the golden sequence.  The bar for the KERNELS is the CPU oracle run with the
    # same bf16 KV cache (every position's arg-max; >= 0.97 asked, near-ties aside they are equal).  Against transformers' fp32-KV
    # tokens the same pass agrees at 0.96 at this shape with synthetic weights -- that gap is the cache precision north_star fixes
    # (bf16 KV), not summation order: the oracle's own bf16-KV pass shows the same positions.
    llm16 = _llm(cfg, None, max_positions=FULL_MAX_POS, kv_dtype="bf16", arena=arena)
    seq = np.concatenate([g["prompt"], g["greedy"][:-1]])
    lg = llm16.forward_logits(seq)[127:].argmax(-1).cpu().numpy()
    full_llm_oracle.kv_dtype = "bf16"
    full_llm_oracle.reset()
    want16 = full_llm_oracle.forward(seq)[127:].argmax(-1).numpy()
    full_llm_oracle.reset()
    agree16 = float((lg == want16).mean())
    assert agree16 >= 0.97, f"bf16-KV teacher-forced arg-max agreement with the bf16-KV oracle {agree16}"
    agree = float((lg == g["greedy"]).mean())
    assert agree >= 0.95 and abs(agree - float((want16 == g["greedy"]).mean())) <= 0.02, f"vs the fp32-KV golden tokens: {agree}"


def test_full_size_batch_is_bit_identical_to_single_runs(full_llm):
    """0.5B shape, 20 ragged sequences (two m-tiles, K-chunked LDS staging) vs B=1 runs: identical
    tokens AND identical logits bits -- the summation order of a row never depends on the batch."""
    from conftest import FULL_MAX_POS
    cfg, syn, arena = full_llm
    rng = np.random.Generator(np.random.PCG64(99))
    B = 20
    prompts = [rng.integers(0, cfg.vocab_size, size=int(rng.integers(3, 70))).tolist() for _ in range(B)]
    from sparkmi.llm import SparkLLM
    big = SparkLLM(cfg, None, "cuda:0", max_slots=B, max_positions=FULL_MAX_POS, arena=arena)
    one = SparkLLM(cfg, None, "cuda:0", max_slots=1, max_positions=FULL_MAX_POS, arena=arena)
    batched = big.generate_ids(prompts, 12)
    for b in (0, 5, 13, 19):
        assert one.generate_ids([prompts[b]], 12)[0] == batched[b], f"sequence {b}"
    # logits bits: a 40-token teacher-forced pass (M=32 and M=8 chunks) vs the same tokens fed one at a time
    ids = rng.integers(0, cfg.vocab_size, size=40)
    chunked = one.forward_logits(ids)[-1].clone()
    one.prefill([ids.tolist()])          # last row runs as an M=1 step with sampling off; compare via argmax + top logits
    step_tok = one.tokens(1)[0][0]
    assert int(chunked.argmax()) == step_tok


@pytest.mark.parametrize("size", ["tiny", "full"])
def test_few_row_down_proj_kernels_keep_the_bits_of_the_general_kernel(tiny, full_llm, monkeypatch, size):
    """k_down1 / k_downS (1 .. 8 rows: four chains per wave, full load instructions) against k_gemm's row-part form
    (SPARKMI_TUNE2 bits 1048576 | 2097152 keep it): teacher-forced logits of 1 .. 8 rows are equal BIT FOR BIT, and so are
    the greedy tokens of ragged batches of 2 .. 8 sequences."""
    from conftest import FULL_MAX_POS
    from sparkmi.llm import SparkLLM
    if size == "tiny":
        cfg, syn = tiny
        mk = lambda slots: _llm(cfg, syn, max_slots=slots, max_positions=96, kv_dtype="bf16")
    else:
        cfg, syn, arena = full_llm
        mk = lambda slots: SparkLLM(cfg, None, "cuda:0", max_slots=slots, max_positions=FULL_MAX_POS, arena=arena, diag=_wants_diag())
    rng = np.random.Generator(np.random.PCG64(77))
    seqs = {S: rng.integers(0, cfg.vocab_size, size=S) for S in range(1, 9)}
    prompts = [rng.integers(0, cfg.vocab_size, size=int(rng.integers(3, 30))).tolist() for _ in range(8)]
    out = {}
    for mode in ("new", "old"):
        if mode == "old":
            monkeypatch.setenv("SPARKMI_TUNE2", str(1048576 | 2097152))
        else:
            monkeypatch.delenv("SPARKMI_TUNE2", raising=False)
        one = mk(1)
        logits = {S: one.forward_logits(ids).clone() for S, ids in seqs.items()}
        toks = {B: mk(B).generate_ids(prompts[:B], 10) for B in (2, 3, 4, 5, 8)}
        toks[1] = [one.generate_ids([q], 10)[0] for q in prompts[:3]]
        out[mode] = (logits, toks)
    if size == "tiny":   # the plain W_down tile order (smi_llm_cfg.wd_plain: a packer-side A/B) through the same kernels: same bits again
        monkeypatch.delenv("SPARKMI_TUNE2", raising=False)
        monkeypatch.setenv("SPARKMI_WD_PLAIN", "1")
        plain = mk(1)
        assert plain._cs.wd_plain == 1
        for S, ids in seqs.items():
            assert torch.equal(plain.forward_logits(ids), out["new"][0][S]), f"{S} rows, plain W_down layout"
        monkeypatch.delenv("SPARKMI_WD_PLAIN")
    for S in seqs:
        assert torch.equal(out["new"][0][S], out["old"][0][S]), f"{S} rows: logits differ from the general kernel's"
    assert out["new"][1] == out["old"][1]
    for B in (2, 3, 4, 5, 8):      # and a row does not depend on its batch
        k = min(B, 3)
        assert out["new"][1][B][:k] == out["new"][1][1][:k], f"batch of {B}"


def test_prefetch_switches_change_speed_only(tiny, monkeypatch):
    """The helper-block / inline L2 prefetches touch caches, never values: every setting gives the same
    tokens and the same logits bits."""
    cfg, syn = tiny
    rng = np.random.Generator(np.random.PCG64(5150))
    prompt = rng.integers(0, cfg.vocab_size, size=37).tolist()
    ids = rng.integers(0, cfg.vocab_size, size=9)
    ref = None
    for env in ({}, {"SPARKMI_NO_PREFETCH": "1", "SPARKMI_PF_INLINE": "0"}, {"SPARKMI_PREFETCH": "7", "SPARKMI_PF_QKV": "8"},
                {"SPARKMI_PREFETCH": "2"}):
        for k in ("SPARKMI_NO_PREFETCH", "SPARKMI_PF_INLINE", "SPARKMI_PREFETCH", "SPARKMI_PF_QKV"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        llm = _llm(cfg, syn, max_positions=128)
        got = (llm.generate_ids([prompt], 40)[0], llm.forward_logits(ids).clone())
        if ref is None:
            ref = got
        assert got[0] == ref[0] and torch.equal(got[1], ref[1]), env


def test_prefill_paths_agree(tiny, monkeypatch):
    """More than 32 prompt rows run as row-grouped decode GEMMs (one launch per layer kernel) or through the prefill GEMM
    (k_pgemm), chosen per kernel by row count (SPARKMI_PGEMM_MIN_ROWS / _QKV / _O / _GU / _D); every mix -- the RESID kernels
    leave their RMSNorm partials in different layouts for the next kernel -- must give the tokens of the 32-row-chunk path and
    of the oracle."""
    cfg, syn = tiny
    rng = np.random.Generator(np.random.PCG64(314))
    B = 12
    prompts = [rng.integers(0, cfg.vocab_size, size=int(rng.integers(30, 60))).tolist() for _ in range(B)]
    assert sum(len(p) - 1 for p in prompts) >= 384

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        out = _llm(cfg, syn, max_slots=B, max_positions=96).generate_ids(prompts, 16)
        for k in env:
            monkeypatch.delenv(k)
        return out

    chunked = run({"SPARKMI_PREFILL_CHUNKS": "1"})
    assert run({"SPARKMI_PGEMM_MIN_ROWS": "100000"}) == chunked          # row-grouped decode GEMMs only
    assert run({"SPARKMI_PGEMM_MIN_ROWS": "0"}) == chunked               # the prefill GEMM only
    assert run({"SPARKMI_PGEMM_MIN_ROWS": "0", "SPARKMI_TUNE2": "16384"}) == chunked   # ... in its register-staged form
    assert run({}) == chunked                                            # the default mix at this row count
    assert run({"SPARKMI_ATTN_PF2": "0"}) == chunked                     # prefill attention per (row, head) instead of 16-row MFMA tiles
    for one in ("QKV", "O", "GU", "D"):
        assert run({"SPARKMI_PGEMM_MIN_ROWS": "100000", "SPARKMI_PGEMM_MIN_" + one: "0"}) == chunked, one
        assert run({"SPARKMI_PGEMM_MIN_ROWS": "0", "SPARKMI_PGEMM_MIN_" + one: "100000"}) == chunked, one
    for b in (0, 5, 11):
        assert Qwen2Ref(cfg, syn, kv_dtype="bf16").generate_greedy(prompts[b], 16) == chunked[b]


def test_generate_ragged_retires_rows_and_equals_the_static_batch(tiny):
    """Per-row budgets with rows retired as they finish (the live row count shrinks, each count's captured step comes from the
    library's cache): row i's tokens are those of the padded static batch truncated to its budget -- greedy, with an eos, on a
    second call (cache hits), and after a sampling-mode change emptied the cache."""
    cfg, syn = tiny
    rng = np.random.Generator(np.random.PCG64(515))
    B = 11
    prompts = [rng.integers(0, cfg.vocab_size, size=int(rng.integers(3, 40))).tolist() for _ in range(B)]
    budgets = [int(v) for v in rng.integers(1, 45, size=B)]
    budgets[3] = budgets[7] = 19          # two rows leave at the same step
    budgets[0] = 1                         # leaves right after the prefill
    llm = _llm(cfg, syn, max_slots=16, max_positions=128)
    static = llm.generate_ids(prompts, max(budgets))
    want = [t[:n] for t, n in zip(static, budgets)]
    assert llm.generate_ragged(prompts, budgets) == want
    assert llm.generate_ragged(prompts, budgets) == want                  # every row count's step now comes from the cache
    eos = static[5][10]
    def cut(t, n):
        t = t[:n]
        return t[: t.index(eos) + 1] if eos in t else t
    assert llm.generate_ragged(prompts, budgets, eos, check_every=3) == [cut(t, n) for t, n in zip(static, budgets)]
    llm.set_sampling(True, 0.8, 50, 0.95, 1234)                           # sampler parameters are kernel arguments: cache emptied
    sampled = llm.generate_ragged(prompts, budgets)
    assert [len(t) for t in sampled] == budgets
    llm.set_sampling(False)
    assert llm.generate_ragged(prompts, budgets) == want
    single = _llm(cfg, syn, max_slots=1, max_positions=128)
    for i in (2, 6, 10):
        assert single.generate_ids([prompts[i]], budgets[i])[0] == want[i]


def test_continuous_batching_equals_standalone_runs(tiny):
    """Sequences admitted and retired between decode steps (in-flight batching) produce exactly the tokens of
    their own B = 1 runs, whatever else is live and whichever KV slot they land in."""
    cfg, syn = tiny
    rng = np.random.Generator(np.random.PCG64(77))
    reqs = []
    for i in range(9):
        prompt = rng.integers(0, cfg.vocab_size, size=int(rng.integers(2, 50))).tolist()
        reqs.append((i, prompt, int(rng.integers(3, 40))))
    single = _llm(cfg, syn, max_slots=1, max_positions=128)
    want = {i: single.generate_ids([p], n)[0] for i, p, n in reqs}
    oracle = Qwen2Ref(cfg, syn, kv_dtype="bf16")          # the CPU oracle, not another HIP engine: three requests
    for i in (0, 4, 8):
        assert oracle.generate_greedy(np.asarray(reqs[i][1]), reqs[i][2]) == want[i], f"request {i} vs oracle"
    longest = max(want, key=lambda i: len(want[i]))
    eos = want[longest][len(want[longest]) // 2]   # a token one of the sequences emits: it must stop there
    want = {i: single.generate_ids([p], n, eos_token_id=eos)[0] for i, p, n in reqs}
    llm = _llm(cfg, syn, max_slots=4, max_positions=128)
    for i in (1, 5, longest):                              # with the stop id: the oracle's run cut at its first eos
        ref = oracle.generate_greedy(np.asarray(reqs[i][1]), reqs[i][2])
        cut = ref[: ref.index(eos) + 1] if eos in ref else ref
        assert cut == want[i], f"request {i} vs oracle (eos)"
    for stride in (1, 5):
        got = dict(llm.serve(((i, p, n, eos) for i, p, n in reqs), max_live=3, decode_stride=stride))
        assert set(got) == set(want)
        for i, p, n in reqs:
            w = want[i]
            g = got[i][: len(w)]           # the driver checks for eos every `stride` steps: extra tokens after it are ignored
            assert g == w, f"request {i} (stride {stride})"
            assert len(got[i]) <= n
    # the low-level calls: slots are reused, a retired slot's history does not leak into the next sequence
    llm.session_begin(None)
    a, = llm.admit([reqs[0][1]])
    b, = llm.admit([reqs[1][1]])
    llm.decode(6)
    ta, _ = llm.slot_tokens(a, 64)
    assert ta == single.generate_ids([reqs[0][1]], 7)[0]
    llm.retire(a)
    c, = llm.admit([reqs[2][1]])
    assert c == a                          # the freed slot
    llm.decode(4)
    assert llm.slot_tokens(c, 64)[0] == single.generate_ids([reqs[2][1]], 5)[0]
    assert llm.slot_tokens(b, 64)[0] == single.generate_ids([reqs[1][1]], 11)[0]
    from sparkmi._lib import SparkMIError
    with pytest.raises(SparkMIError):
        llm.retire(3)                      # not live
    with pytest.raises(SparkMIError):
        llm.admit([[1, 2]] * 4)            # 2 live + 4 new > 4 slots


def test_full_size_sampling_fast_path_matches_the_warper_chain(full_llm, golden_dir):
    """0.5B vocabulary (166 000 logits, 512 lm_head blocks).  (1) End to end: the one-pass candidate collection (bound = the
    top_k-th largest lm_head block maximum) behind the real lm_head, >= 20 480 first tokens against the oracle chain on the
    kernel's own logits.  (2) The sampler alone on the fixture's 166 000-entry row against what transformers' warpers keep
    there, through the bound path AND the exact radix selection, three parameter sets (up to 256 survivors)."""
    from conftest import FULL_MAX_POS
    from oracle.sampling_ref import sampling_probs
    cfg, syn, arena = full_llm
    prompt = np.random.Generator(np.random.PCG64(5)).integers(0, cfg.vocab_size, size=24).tolist()
    llm = _llm(cfg, None, max_slots=32, max_positions=FULL_MAX_POS, arena=arena, diag=True)
    logits = llm.forward_logits(prompt)[-1].cpu()
    T_, K, P = 0.8, 50, 0.95
    want = sampling_probs(logits, T_, K, P).numpy()
    counts = np.zeros(cfg.vocab_size)
    for seed in range(20480 // 32):
        for t in llm.generate_ids([prompt] * 32, 1, do_sample=True, temperature=T_, top_k=K, top_p=P, seed=seed):
            counts[t[0]] += 1
    _check_draws(counts, want, "0.5B end to end")
    greedy = llm.generate_ids([prompt], 12)[0]
    assert llm.generate_ids([prompt], 12, do_sample=True, top_k=1, seed=3)[0] == greedy
    g = np.load(os.path.join(golden_dir, "sampling.npz"))
    row = _fixture_row(g, "big")
    assert len(row) == cfg.vocab_size
    for i, (t, k, p) in enumerate(g["big.params"]):
        want = np.zeros(len(row))
        want[g[f"big.{i}.ids"]] = g[f"big.{i}.probs"]
        llm.set_sampling(True, float(t), int(k), float(p), 0)
        for bound in (True, False):
            counts = np.zeros(len(row))
            draws = _tv_draws(int((want > 0).sum())) if bound else 20480
            for seed in range((draws + 31) // 32):
                np.add.at(counts, llm.debug_sample(row if seed == 0 else None, 32, 7000 + seed, use_bound=bound), 1)
            if bound or (want > 0).sum() <= 64:
                _check_draws(counts, want, f"big[{i}] bound={bound}")
            else:   # radix path at up to 256 survivors: fewer draws, so only the support and a looser distance
                assert (counts[want == 0] == 0).all()
                assert 0.5 * np.abs(counts / counts.sum() - want).sum() < 0.08
    llm.set_sampling(False)


def test_cache_boundaries_and_long_contexts(tiny):
    """Generation right up to max_positions, and contexts that cross the attention kernel's 256-token chunks
    (1, 2 and 3 chunks; the last chunk partially filled), against the oracle."""
    cfg, syn = tiny
    rng = np.random.Generator(np.random.PCG64(123))
    ref = Qwen2Ref(cfg, syn, kv_dtype="bf16")
    p10 = rng.integers(0, cfg.vocab_size, size=10).tolist()
    llm = _llm(cfg, syn, max_positions=64)
    assert llm.generate_ids([p10], 54)[0] == ref.generate_greedy(p10, 54)      # 10 + 54 == max_positions
    with pytest.raises(ValueError):
        llm.generate_ids([p10], 55)
    long_prompt = rng.integers(0, cfg.vocab_size, size=250).tolist()
    llm = _llm(cfg, syn, max_slots=2, max_positions=640)
    got = llm.generate_ids([long_prompt, long_prompt[:77]], 300)               # ctx 250 -> 550 and 77 -> 377
    assert got[0] == ref.generate_greedy(long_prompt, 300)
    assert got[1] == ref.generate_greedy(long_prompt[:77], 300)
    # beyond 1024 keys the context is cut into segments over several blocks (k_attn_merge): a sequence that crosses the
    # boundary while a short one runs beside it, and the same long sequence alone
    vlong = rng.integers(0, cfg.vocab_size, size=1005).tolist()
    want = ref.generate_greedy(vlong, 40)
    llm = _llm(cfg, syn, max_slots=2, max_positions=1100)
    got = llm.generate_ids([vlong, long_prompt[:30]], 40)
    assert got[0] == want and got[1] == ref.generate_greedy(long_prompt[:30], 40)
    assert llm.generate_ids([vlong], 40)[0] == want
    # a prompt that is itself longer than a segment: the prefill attention (one wave per row and head, k_attn_pf)
    # walks the whole context in one loop, and decode then starts on a two-segment context
    xlong = rng.integers(0, cfg.vocab_size, size=1300).tolist()
    llm = _llm(cfg, syn, max_slots=1, max_positions=1320)
    assert llm.generate_ids([xlong], 12)[0] == ref.generate_greedy(xlong, 12)


def test_sixty_four_concurrent_sequences_equal_single_runs(tiny):
    """SMI_MAX_ROWS = 64 live sequences per step (four 16-row block rows in the few-tile GEMVs, two 32-row ones in
    gate_up, four lm_head passes): every sequence still equals its own B = 1 run."""
    cfg, syn = tiny
    rng = np.random.Generator(np.random.PCG64(64))
    B = 64
    prompts = [rng.integers(0, cfg.vocab_size, size=int(rng.integers(1, 30))).tolist() for _ in range(B)]
    batched = _llm(cfg, syn, max_slots=B, max_positions=64).generate_ids(prompts, 20)
    single = _llm(cfg, syn, max_slots=1, max_positions=64)
    for b in (0, 15, 16, 31, 32, 47, 48, 63):
        assert single.generate_ids([prompts[b]], 20)[0] == batched[b], f"sequence {b}"
    from sparkmi._lib import SparkMIError
    with pytest.raises(SparkMIError):
        _llm(cfg, syn, max_slots=65, max_positions=64)


def test_generate_called_exactly_as_the_reference_calls_it(tiny):
    """cli/SparkTTS.py:197-204: ``model.generate(**model_inputs, max_new_tokens=3000, do_sample=True, top_k=..,
    top_p=.., temperature=..)`` -- no eos argument, a budget far beyond this engine's cache.  HF stops on ANY id of
    generation_config.json's eos_token_id (a list), so the drop-in must too, and must not raise over the budget."""
    cfg, syn = tiny
    prompt = [11, 22, 33, 44, 55]
    free = _llm(cfg, syn, max_positions=128, kv_dtype="f32").generate_ids([prompt], 40)[0]
    # two stop ids; the SECOND of the list is emitted first (position 9), the first one later (position 20)
    second, first = free[9], free[20]
    assert second not in free[:9] and first not in free[:9]
    llm = _llm(cfg, syn, max_positions=128, kv_dtype="f32", eos_token_ids=[first, second])
    ids = torch.tensor([prompt])
    out = llm.generate(input_ids=ids, attention_mask=torch.ones_like(ids), max_new_tokens=3000, do_sample=True,
                       top_k=1, top_p=0.95, temperature=0.8)      # top_k = 1: the sampled path, deterministic
    new = out[0, len(prompt):].tolist()
    assert new == free[:10]
    assert Qwen2Ref(cfg, syn).generate_greedy(prompt, 3000, eos_ids=[first, second]) == new
    # no stop id emitted within the cache: generation ends at the context limit instead of raising
    llm2 = _llm(cfg, syn, max_positions=64, kv_dtype="f32", eos_token_ids=[cfg.vocab_size - 1])
    out2 = llm2.generate(input_ids=ids, max_new_tokens=3000)
    assert out2.shape[1] == 64 or (cfg.vocab_size - 1) in out2[0].tolist()
    with pytest.raises(ValueError):
        llm.generate_ids([prompt], 10, eos_token_id=[1, 2, 3, 4, 5])     # more stop ids than the step kernel checks


def test_fp32_checkpoint_is_rounded_and_reported(tiny):
    """An fp32-saved checkpoint is NOT what the bf16 weight arena holds: packing rounds it (north_star fixes bf16
    weights) and says so.  The GPU then computes exactly the bf16-rounded model; its distance from the fp32 model is
    what a user of an fp32 checkpoint sees against the reference's CPU run."""
    import warnings
    from sparkmi import arena as A
    cfg, _ = tiny
    raw = W.SyntheticLLM(cfg, bf16_exact=False)
    ids = np.random.Generator(np.random.PCG64(17)).integers(0, cfg.vocab_size, size=40)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        llm = _llm(cfg, raw, max_positions=96, kv_dtype="f32")
    assert any("not bf16-representable" in str(w.message) for w in rec)
    with pytest.raises(ValueError, match="bf16-representable"):
        A.pack_llm_arena(cfg, raw, A.llm_cfg_struct(cfg, 1, 96, "f32", True), strict_bf16=True)
    got = llm.forward_logits(ids).cpu().numpy()
    rounded = {n: (W.round_bf16(raw[n]) if raw[n].ndim == 2 else raw[n]) for n in raw.names()}
    same = Qwen2Ref(cfg, rounded).forward(ids).numpy()
    assert np.abs(got - same).max() < LOGIT_ATOL                   # the GPU runs the rounded model exactly
    full = Qwen2Ref(cfg, raw).forward(ids).numpy()
    dev = float(np.abs(got - full).max())
    agree = float((got.argmax(-1) == full.argmax(-1)).mean())
    print(f"fp32 checkpoint through the bf16 arena: max |logit diff| {dev:.3e}, greedy agreement {agree:.3f}")
    assert LOGIT_ATOL < dev < 0.5 and agree > 0.8                  # visible, bounded: bf16 weight rounding (2^-9 relative)


@pytest.mark.parametrize("size", ["tiny", "full"])
def test_exact_weights_mode_gives_the_fp32_models_tokens(tiny, size):
    """smi_llm_cfg.weights_exact (verification mode): on a checkpoint SAVED in fp32 -- not bf16-representable, like the published
    Spark-TTS-0.5B LLM/model.safetensors that the reference loads as saved (cli/SparkTTS.py:48-51) -- the GPU's teacher-forced
    logits equal the UNROUNDED fp32 oracle's to summation-order noise and the free-running greedy tokens are the oracle's (where
    the default bf16 arena stands ~1e-2 away and a tenth of the decisions change: the test above).  north_star's acceptance
    sentence on such a checkpoint; the mode's one-row step time is printed."""
    import time
    if size == "tiny":
        cfg, _ = tiny
        n_new, max_pos = 48, 160
    else:
        cfg = C.spark_0p5b_llm()
        n_new, max_pos = 24, 96
    raw = W.SyntheticLLM(cfg, bf16_exact=False)
    rng = np.random.Generator(np.random.PCG64(17))
    ids = rng.integers(0, cfg.vocab_size, size=40)
    prompt = rng.integers(0, cfg.vocab_size, size=33).tolist()
    ref = Qwen2Ref(cfg, raw)                                        # the fp32 model as saved, fp32 KV
    llm = _llm(cfg, raw, max_slots=4, max_positions=max_pos, kv_dtype="f32", weights_exact=True)
    got = llm.forward_logits(ids).cpu().numpy()
    want = ref.forward(ids).numpy()
    err = float(np.abs(got - want).max())
    assert err < LOGIT_ATOL, f"exact-weights logits: max |diff| {err} from the fp32 oracle"
    assert (got.argmax(-1) == want.argmax(-1)).all()
    ref.reset()
    toks = llm.generate_ids([prompt], n_new)[0]
    assert toks == ref.generate_greedy(prompt, n_new)
    # a ragged batch through the same mode: rows are independent here too
    prompts = [rng.integers(0, cfg.vocab_size, size=int(k)).tolist() for k in (5, 19, 33)]
    batch = llm.generate_ids(prompts + [prompt], 8)
    assert batch[3] == toks[:8]
    for b in range(3):
        ref.reset()
        assert batch[b] == ref.generate_greedy(prompts[b], 8), f"row {b}"
    if size == "full":
        llm.prefill([prompt]); llm.decode(4); torch.cuda.synchronize()
        t0 = time.perf_counter(); llm.decode(32); torch.cuda.synchronize()
        print(f"exact-weights mode, 0.5B, one row: {(time.perf_counter() - t0) / 32 * 1e6:.0f} us per decode step")


def test_arena_and_config_must_agree_on_how_the_weights_are_stored(tiny):
    """The arena carries a layout tag (include/sparkmi.h: smi_llm_arena_tag): an arena packed for the exact-weights mode, another
    W_down tile order or other dimensions is refused by smi_llm_create instead of being streamed as something it is not."""
    from sparkmi import arena as A
    from sparkmi._lib import SparkMIError
    from sparkmi.llm import SparkLLM
    cfg, syn = tiny
    packed = torch.from_numpy(A.pack_llm_arena(cfg, syn, A.llm_cfg_struct(cfg, 1, 96, "bf16", True))).to("cuda:0")
    SparkLLM(cfg, None, "cuda:0", max_positions=96, arena=packed)                                  # fine
    with pytest.raises(SparkMIError, match="dimensions|too small"):
        SparkLLM(cfg, None, "cuda:0", max_positions=128, arena=packed)                             # another RoPE table size
    with pytest.raises(SparkMIError, match="weights_exact|too small"):
        SparkLLM(cfg, None, "cuda:0", max_positions=96, arena=packed, weights_exact=True)


def test_sampled_tokens_do_not_depend_on_batch_composition(tiny):
    """The sampler's stream is keyed by (seed; the sequence's admission number, its own token index): a request
    draws the same tokens whether it runs alone, beside others, or lands in another row after a neighbour retires."""
    cfg, syn = tiny
    rng = np.random.Generator(np.random.PCG64(31))
    reqs = [(i, rng.integers(0, cfg.vocab_size, size=int(rng.integers(3, 30))).tolist(), int(rng.integers(6, 30)), None)
            for i in range(6)]
    llm = _llm(cfg, syn, max_slots=4, max_positions=96)
    llm.set_sampling(True, 0.9, 40, 0.95, seed=77)
    alone = dict(llm.serve(iter(reqs), max_live=1, decode_stride=3))
    llm.set_sampling(True, 0.9, 40, 0.95, seed=77)
    crowd = dict(llm.serve(iter(reqs), max_live=4, decode_stride=3))
    for i, _, n, _ in reqs:
        assert crowd[i][:n] == alone[i][:n], f"request {i}"
    llm.set_sampling(True, 0.9, 40, 0.95, seed=78)
    other = dict(llm.serve(iter(reqs), max_live=4, decode_stride=3))
    assert any(other[i] != crowd[i] for i, *_ in reqs)
    # the same serving path with the sampler off against the CPU oracle (the sampled runs above can only be compared with
    # each other: another generator's draws are not reproducible bit for bit)
    llm.set_sampling(False)
    greedy = dict(llm.serve(iter(reqs), max_live=4, decode_stride=3))
    oracle = Qwen2Ref(cfg, syn, kv_dtype="bf16")
    for i in (0, 2, 5):
        assert greedy[i][: reqs[i][2]] == oracle.generate_greedy(np.asarray(reqs[i][1]), reqs[i][2]), f"request {i} vs oracle"


def test_paged_kv_cache_serves_more_sequences_than_it_could_reserve(tiny):
    """Paged KV (the functional analogue of TensorRT-LLM's paged KV under in-flight batching, run.sh:50-65): a pool of
    pages shared by the slots instead of max_positions reserved per slot.  Tokens equal the contiguous B = 1 runs; the
    pool is far smaller than slots x max_positions; pages return at retire; a short pool refuses cleanly."""
    from sparkmi._lib import SparkMIError
    cfg, syn = tiny
    rng = np.random.Generator(np.random.PCG64(404))
    reqs = []
    for i in range(10):
        prompt = rng.integers(0, cfg.vocab_size, size=int(rng.integers(2, 90))).tolist()
        reqs.append((i, prompt, int(rng.integers(5, 60)), None))
    single = _llm(cfg, syn, max_slots=1, max_positions=256)
    want = {i: single.generate_ids([p], n)[0] for i, p, n, _ in reqs}
    # 8 slots x 256 positions would reserve 2048 tokens; the pool holds 40 pages x 16 = 640
    for kv in ("bf16", "f32"):
        llm = _llm(cfg, syn, max_slots=8, max_positions=256, kv_dtype=kv, kv_page_tokens=16, kv_pages=40)
        if kv == "f32":
            single32 = _llm(cfg, syn, max_slots=1, max_positions=256, kv_dtype="f32")
            want = {i: single32.generate_ids([p], n)[0] for i, p, n, _ in reqs}
        assert llm.kv_pages() == (40, 40)
        got = dict(llm.serve(iter(reqs), max_live=5, decode_stride=4))
        for i, p, n, _ in reqs:
            assert got[i][:n] == want[i], f"request {i} ({kv})"
        oracle = Qwen2Ref(cfg, syn, kv_dtype=kv)               # the paged engine against the CPU oracle, three requests
        for i in (1, 6, 9):
            assert got[i][: reqs[i][2]] == oracle.generate_greedy(np.asarray(reqs[i][1]), reqs[i][2]), f"request {i} ({kv}) vs oracle"
        assert llm.kv_pages() == (40, 40)                     # everything retired: every page is back
    # plain (static-batch) generation on a paged engine, ragged prompts, two m-tiles worth of rows
    llm = _llm(cfg, syn, max_slots=8, max_positions=256, kv_page_tokens=32, kv_pages=24)
    prompts = [r[1] for r in reqs[:6]]
    single = _llm(cfg, syn, max_slots=1, max_positions=256)
    assert llm.generate_ids(prompts, 30) == [single.generate_ids([p], 30)[0] for p in prompts]
    tot, free = llm.kv_pages()
    assert tot == 24 and 0 < free < 24
    ids = np.asarray(prompts[3] + [1, 2, 3])
    assert torch.equal(llm.forward_logits(ids), _llm(cfg, syn, max_positions=256).forward_logits(ids))
    # ONE sequence on a paged engine takes the one-row kernels (one-row attention through the page table, fused o_proj):
    # tokens equal the contiguous engine's and the oracle's, across several page boundaries, graph and eager
    for graph in (True, False):
        one = _llm(cfg, syn, max_slots=2, max_positions=256, kv_page_tokens=16, kv_pages=20, use_graph=graph)
        for p in (prompts[0], prompts[4]):
            got1 = one.generate_ids([p], 70)[0]
            assert got1 == single.generate_ids([p], 70)[0]
        assert got1 == Qwen2Ref(cfg, syn, kv_dtype="bf16").generate_greedy(np.asarray(prompts[4]), 70)
    # a pool that cannot hold the request refuses it and changes nothing
    small = _llm(cfg, syn, max_slots=4, max_positions=256, kv_page_tokens=16, kv_pages=6)
    small.session_begin(None)
    a, = small.admit([list(range(1, 60))])                    # 59 tokens -> 4 pages
    with pytest.raises(SparkMIError, match="pool exhausted"):
        small.admit([list(range(1, 70))])                     # 5 more pages: only 2 free
    assert small.kv_pages() == (6, 2)
    small.decode(5)                                           # 64 positions: still 4 pages
    with pytest.raises(SparkMIError, match="pool exhausted"):
        small.decode(40)                                      # would need 7 pages
    assert small.slot_tokens(a, 64)[0] == _llm(cfg, syn, max_positions=256).generate_ids([list(range(1, 60))], 6)[0]
    small.retire(a)
    assert small.kv_pages() == (6, 6)
    with pytest.raises(SparkMIError):
        _llm(cfg, syn, max_positions=250, kv_page_tokens=16, kv_pages=8)      # page size must divide max_positions
