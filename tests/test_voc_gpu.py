"""GPU parity of the HIP vocoder (through the C ABI) against the CPU oracle and the vectors the
reference's own modules produced.  Default mode: the dense stack on the bf16 matrix pipe with both operands split into two
bf16 planes (three products, fp32 accumulate) -- 5e-5 from the fp32 result at the 0.5B shape; exact_fp32=True: every
contraction on the exact-fp32 pipe (summation-order noise only).  The bound written here, 1e-4 on a waveform in [-1, 1]
(3e-4 at full size; north_star allows 1e-3), holds for both; the mode-specific tests are at the end."""
import os

import numpy as np
import pytest
import torch

from oracle.bicodec_ref import BiCodecDetokRef
from sparkmi import config as C, weights as W

pytestmark = pytest.mark.gpu

WAV_ATOL = 1e-4


def _voc(cfg, sd, **kw):
    from sparkmi.bicodec import BiCodecVocoder
    kw.setdefault("diag", any(k.startswith("SPARKMI_") for k in os.environ))   # SPARKMI_* switches exist in the diagnostics build only
    return BiCodecVocoder(cfg, sd, device="cuda:0", **kw)


@pytest.fixture(scope="module")
def tiny():
    cfg = C.tiny_bicodec()
    sd = W.bicodec_detok_state(cfg)
    return cfg, sd, BiCodecDetokRef(cfg, W.fold_weight_norm(sd))


@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_waveform_matches_reference_vectors(tiny, golden_dir, case):
    cfg, sd, _ = tiny
    g = np.load(os.path.join(golden_dir, "vocoder_tiny.npz"))
    voc = _voc(cfg, sd, max_frames=160)
    wav = voc.detokenize(torch.from_numpy(g[f"c{case}_semantic"]), torch.from_numpy(g[f"c{case}_global"])).cpu().numpy()
    assert wav.shape == g[f"c{case}_wav"].shape
    err = np.abs(wav - g[f"c{case}_wav"]).max()
    assert err < WAV_ATOL, f"max |wav diff| {err}"
    d = voc.debug_stage(-1, 0, 0, 1).cpu().numpy()
    assert np.abs(d - g[f"c{case}_d_vector"]).max() < 1e-5


def test_intermediate_stages_match_reference_vectors(tiny, golden_dir, monkeypatch):
    monkeypatch.setenv("SPARKMI_VOC_DEBUG", "1")
    cfg, sd, _ = tiny
    g = np.load(os.path.join(golden_dir, "vocoder_tiny.npz"))
    voc = _voc(cfg, sd, max_frames=64)
    T = g["c0_semantic"].shape[1]
    voc.detokenize(torch.from_numpy(g["c0_semantic"]), torch.from_numpy(g["c0_global"]))
    zq = voc.debug_stage(0, cfg.vq_input_dim, T, 1).cpu().numpy()
    assert np.abs(zq - g["c0_z_q"]).max() < 1e-5
    x0 = voc.debug_stage(1, cfg.dec_input_channel, T, 1).cpu().numpy()
    assert np.abs(x0 - g["c0_prenet_plus_d"]).max() < 1e-4
    L, ch = T, cfg.dec_channels
    s0 = voc.debug_stage(2, ch, L, 1).cpu().numpy()
    assert np.abs(s0 - g["c0_wavegen0"]).max() < 2e-4
    for i, r in enumerate(cfg.dec_rates):
        L *= r
        si = voc.debug_stage(3 + i, ch >> (i + 1), L, 1).cpu().numpy()
        assert np.abs(si - g[f"c0_wavegen{i + 1}"]).max() < 5e-4, f"block {i}"


def test_ragged_batch_equals_unpadded_rows(tiny):
    cfg, sd, ref = tiny
    rng = np.random.Generator(np.random.PCG64(77))
    lens = [33, 1, 70, 8, 64]
    B, T = len(lens), max(lens)
    sem = rng.integers(0, cfg.codebook_size, size=(B, T))
    glob = rng.integers(0, 4096, size=(B, 1, cfg.spk_token_num))
    voc = _voc(cfg, sd, max_batch=B, max_frames=80)
    wav = voc.detokenize(torch.from_numpy(sem), torch.from_numpy(glob), lengths=lens).cpu().numpy()
    # the same rows in other batch positions, next to other neighbours, behind different padding: bit-identical
    # (every load is masked by the row's own length; nothing is reduced across rows)
    perm = [3, 0, 4, 1, 2]
    sem2 = sem[perm].copy()
    for i, b in enumerate(perm):
        sem2[i, lens[b]:] = rng.integers(0, cfg.codebook_size, size=T - lens[b])      # different padding garbage
    wav2 = voc.detokenize(torch.from_numpy(sem2), torch.from_numpy(glob[perm]), lengths=[lens[b] for b in perm]).cpu().numpy()
    for i, b in enumerate(perm):
        assert np.array_equal(wav2[i], wav[b]), f"row {b} changed when it moved to batch position {i}"
    one = _voc(cfg, sd, max_batch=1, max_frames=80)
    for b, n in enumerate(lens):
        assert not wav[b, 0, n * cfg.hop:].any()
        # a call of another shape (B, longest row) may tile a layer differently and split its channel sum over the
        # four waves or not: the row on its own equals its batched self to fp32 re-association
        solo = one.detokenize(torch.from_numpy(sem[b:b + 1, :n]), torch.from_numpy(glob[b:b + 1])).cpu().numpy()
        assert np.abs(wav[b, 0, : n * cfg.hop] - solo[0, 0]).max() < 2e-5, f"row {b} differs from its un-padded run"
        oracle = ref.detokenize(torch.from_numpy(sem[b:b + 1, :n]), torch.from_numpy(glob[b:b + 1])).numpy()
        assert np.abs(solo - oracle).max() < WAV_ATOL


def test_facade_and_determinism(tiny):
    from sparkmi.bicodec import BiCodecTokenizer
    cfg, sd, ref = tiny
    tok = BiCodecTokenizer(device="cuda:0", cfg=cfg, state=sd, max_frames=64)
    rng = np.random.Generator(np.random.PCG64(5))
    sem = torch.from_numpy(rng.integers(0, cfg.codebook_size, size=(1, 21)))
    glob = torch.from_numpy(rng.integers(0, 4096, size=(1, cfg.spk_token_num)))
    a = tok.detokenize(glob, sem)
    b = tok.detokenize(glob, sem)
    assert a.shape == (21 * cfg.hop,) and a.dtype == np.float32
    assert np.array_equal(a, b)
    assert np.abs(a - ref.detokenize_numpy(glob, sem)).max() < WAV_ATOL
    with pytest.raises(FileNotFoundError):
        tok.tokenize("x.wav")      # no such prompt file (tokenize itself: tests/test_enc_gpu.py, test_pipeline_gpu.py)


def test_bad_arguments_are_reported(tiny):
    from sparkmi._lib import SparkMIError
    cfg, sd, _ = tiny
    voc = _voc(cfg, sd, max_batch=1, max_frames=16)
    with pytest.raises(SparkMIError):
        voc.detokenize(torch.zeros((1, 17), dtype=torch.long), torch.zeros((1, 1, cfg.spk_token_num), dtype=torch.long))
    with pytest.raises(SparkMIError):
        voc.detokenize(torch.zeros((2, 4), dtype=torch.long), torch.zeros((2, 1, cfg.spk_token_num), dtype=torch.long))


def test_full_size_0p5b_against_reference_vectors(golden_dir):
    """Spark-TTS-0.5B BiCodec shape, synthetic weights: 150 frames -> 48 000 samples."""
    cfg = C.spark_0p5b_bicodec()
    sd = W.bicodec_detok_state(cfg)
    g = np.load(os.path.join(golden_dir, "vocoder_full.npz"))
    voc = _voc(cfg, sd, max_batch=2, max_frames=160)
    for case in (0, 1):
        wav = voc.detokenize(torch.from_numpy(g[f"c{case}_semantic"]), torch.from_numpy(g[f"c{case}_global"])).cpu().numpy()
        err = np.abs(wav - g[f"c{case}_wav"]).max()
        assert err < 1e-3, f"case {case}: max |wav diff| {err}"   # north_star bound
        assert err < WAV_ATOL * 3, f"case {case}: max |wav diff| {err}"
    # the two as one ragged batch
    T = 150
    sem = np.zeros((2, T), np.int64)
    sem[0], sem[1, :23] = g["c0_semantic"][0], g["c1_semantic"][0]
    glob = np.concatenate([g["c0_global"], g["c1_global"]])
    wav = voc.detokenize(torch.from_numpy(sem), torch.from_numpy(glob), lengths=[150, 23]).cpu().numpy()
    assert np.abs(wav[0] - g["c0_wav"][0]).max() < 3e-4
    assert np.abs(wav[1, :, : 23 * 320] - g["c1_wav"][0]).max() < 3e-4


def test_exact_fp32_mode_and_the_bf16_split_pipe_agree(golden_dir, full_voc):
    """Default: the dense stack runs on the bf16 matrix pipe with both operands split into two bf16 planes (three products,
    fp32 accumulate).  exact_fp32=True keeps every contraction on the exact-fp32 MFMA (verification mode).  Both against the
    reference's own vector at the 0.5B shape, and against each other: the split costs < 1e-4 on a waveform in [-1, 1]
    (north_star allows 1e-3); the exact mode is summation-order noise only."""
    cfg, sd, _ = full_voc
    g = np.load(os.path.join(golden_dir, "vocoder_full.npz"))
    sem, glob = torch.from_numpy(g["c0_semantic"]), torch.from_numpy(g["c0_global"])
    exact = _voc(cfg, sd, max_batch=1, max_frames=160, exact_fp32=True)
    fast = _voc(cfg, sd, max_batch=1, max_frames=160, exact_fp32=False)
    assert exact.exact_fp32 and not fast.exact_fp32
    we = exact.detokenize(sem, glob).cpu().numpy()
    wf = fast.detokenize(sem, glob).cpu().numpy()
    assert np.abs(we - g["c0_wav"]).max() < 3e-5
    assert np.abs(wf - g["c0_wav"]).max() < 2e-4
    assert np.abs(wf - we).max() < 2e-4
    assert np.array_equal(wf, fast.detokenize(sem, glob).cpu().numpy())     # deterministic


def test_tiny_exact_fp32_mode_matches_reference_vectors(tiny, golden_dir):
    cfg, sd, _ = tiny
    g = np.load(os.path.join(golden_dir, "vocoder_tiny.npz"))
    voc = _voc(cfg, sd, max_frames=160, exact_fp32=True)
    for case in range(4):
        wav = voc.detokenize(torch.from_numpy(g[f"c{case}_semantic"]), torch.from_numpy(g[f"c{case}_global"])).cpu().numpy()
        assert np.abs(wav - g[f"c{case}_wav"]).max() < WAV_ATOL



def test_multi_phase_transposed_convs_at_batch(full_voc, monkeypatch):
    """k_convbT keeps PH output phases of a ConvTranspose1d in one block (the chunk staged once, a lane stores the PH consecutive
    outputs of each of its rows) wherever the grid still fills the chip: at the 0.5B shape every transposed conv of a 6-row batch
    (strides 8, 5, 4, 2 -> 4, 5, 4 and 2 phases per block).  Per phase the taps and 16-channel steps run in k_convb's order with
    64-channel chunks, so the ragged batch equals the same call on k_convb alone (SPARKMI_CBT=0, diagnostics build) bit for bit,
    and its rows agree with their solo runs (another launch plan: fp32 re-association only)."""
    cfg, sd, _ = full_voc
    rng = np.random.Generator(np.random.PCG64(78))
    B, T = 6, 150
    lens = [150, 97, 150, 33, 128, 1]
    sem = torch.from_numpy(rng.integers(0, cfg.codebook_size, size=(B, T)))
    glob = torch.from_numpy(rng.integers(0, 4096, size=(B, 1, cfg.spk_token_num)))
    new = _voc(cfg, sd, max_batch=B, max_frames=T + 10, diag=False)
    w_new = new.detokenize(sem, glob, lengths=lens).cpu().numpy()
    monkeypatch.setenv("SPARKMI_CBT", "0")
    old = _voc(cfg, sd, max_batch=B, max_frames=T + 10, diag=True)
    w_old = old.detokenize(sem, glob, lengths=lens).cpu().numpy()
    assert np.array_equal(w_new, w_old)
    for b in (1, 3, 5):
        solo = new.detokenize(sem[b:b + 1, :lens[b]], glob[b:b + 1]).cpu().numpy()
        assert np.abs(solo[0, :, : lens[b] * cfg.hop] - w_new[b, :, : lens[b] * cfg.hop]).max() < 1e-4
        assert not w_new[b, :, lens[b] * cfg.hop:].any()


def test_fused_residual_units_agree_with_the_two_launch_form(full_voc, golden_dir, monkeypatch):
    """k_resunit runs a whole ResidualUnit -- 7-tap conv, Snake, 1x1 conv, residual -- in one launch at C = 96 and 192 (a block
    holds every channel of its time tile; the 7-tap conv's output never leaves the CU).  Same products as the two launches; its
    7-tap sum walks 48- instead of 32-channel chunks: the waveform agrees with the two-launch form (SPARKMI_RESFUSE=0,
    diagnostics build) to fp32 re-association and with the reference's own vector within the usual bound; six launches fewer."""
    cfg, sd, _ = full_voc
    g = np.load(os.path.join(golden_dir, "vocoder_full.npz"))
    T = 150
    sem = np.zeros((2, T), np.int64)
    sem[0], sem[1, :23] = g["c0_semantic"][0], g["c1_semantic"][0]
    glob = np.concatenate([g["c0_global"], g["c1_global"]])
    fused = _voc(cfg, sd, max_batch=2, max_frames=160, diag=False)
    wf = fused.detokenize(torch.from_numpy(sem), torch.from_numpy(glob), lengths=[150, 23]).cpu().numpy()
    names = [fused.time_launch(i, iters=1)[0] for i in range(fused.launches())]
    assert sum(n.endswith(".conv7+conv1+res") for n in names) == 6, names
    monkeypatch.setenv("SPARKMI_RESFUSE", "0")
    two = _voc(cfg, sd, max_batch=2, max_frames=160, diag=True)
    wt = two.detokenize(torch.from_numpy(sem), torch.from_numpy(glob), lengths=[150, 23]).cpu().numpy()
    assert two.launches() == fused.launches() + 6
    assert np.abs(wf - wt).max() < 2e-5
    assert np.abs(wf[0] - g["c0_wav"][0]).max() < 3e-4
    assert np.abs(wf[1, :, : 23 * 320] - g["c1_wav"][0]).max() < 3e-4
    assert not wf[1, :, 23 * 320:].any()


def test_round4_conv_kernels_at_odd_shapes(full_voc, monkeypatch):
    """The round-4 kernels (multi-phase transposed convs, fused residual units, the small-grid prefetch forms) at a frame count that is
    no multiple of any tile (97 frames -> 31 040 samples: partial last tiles at every resolution) in a ragged 9-row batch, against the
    same call with all of them switched off (diagnostics build): fp32 re-association only (the fused unit's 48-channel chunks), rows
    zero behind their own length."""
    cfg, sd, _ = full_voc
    rng = np.random.Generator(np.random.PCG64(79))
    B, T = 9, 97
    lens = [97, 96, 65, 64, 63, 33, 32, 31, 2]
    sem = torch.from_numpy(rng.integers(0, cfg.codebook_size, size=(B, T)))
    glob = torch.from_numpy(rng.integers(0, 4096, size=(B, 1, cfg.spk_token_num)))
    new = _voc(cfg, sd, max_batch=B, max_frames=T + 3, diag=False)
    w_new = new.detokenize(sem, glob, lengths=lens).cpu().numpy()
    names = [new.time_launch(i, iters=1)[0] for i in range(new.launches())]
    assert sum(n.endswith(".conv7+conv1+res") for n in names) == 6
    for k in ("SPARKMI_CBT", "SPARKMI_RESFUSE", "SPARKMI_CB_CHG"):
        monkeypatch.setenv(k, "0")
    monkeypatch.setenv("SPARKMI_CB_NOWPF", "1")
    monkeypatch.setenv("SPARKMI_CB_NOWALL", "1")
    old = _voc(cfg, sd, max_batch=B, max_frames=T + 3, diag=True)
    w_old = old.detokenize(sem, glob, lengths=lens).cpu().numpy()
    assert old.launches() == new.launches() + 6
    assert np.abs(w_new - w_old).max() < 2e-5
    for b in range(B):
        assert not w_new[b, :, lens[b] * cfg.hop:].any()
        assert np.abs(w_new[b, :, : lens[b] * cfg.hop]).max() > 1e-3
