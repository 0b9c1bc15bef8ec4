"""Op-level HIP tests on the reference-class fixtures (tests/golden/ops_layers.npz, written by gen_golden.py from the
reference's OWN layer classes): k_conv / k_convb / k_dwln run ONE block at a time through smi_voc_block_run -- built by the
functions smi_voc_forward builds its launches with -- and are compared with the outputs the reference classes produced.
tests/test_oracle_ops.py pins the oracle on the same vectors; together they localise a mismatch to one block."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# exact-fp32 matrix pipe: fp32 re-association only.  Default pipe: the dense layers with >= 32 channels on both sides run as
# two-plane bf16 splits (3 of the 4 partial products kept): relative 2^-16 per product, far inside north_star's 1e-3.
TOL_EXACT, TOL_SPLIT = 1e-5, 2e-4


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "ops_layers.npz"))


def _sub(g, prefix):
    return {"L." + k[len(prefix) + 1:]: g[k] for k in g.files if k.startswith(prefix + "/")}


def _snake(x, alpha):
    from oracle.bicodec_ref import snake
    return snake(x, torch.from_numpy(np.asarray(alpha)).reshape(1, -1, 1))


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("dil", [1, 3, 9])
def test_residual_unit_on_hip(g, dil, exact):
    """sparktts/modules/blocks/layers.py:51-67: conv7 (dilated) with the second Snake in its epilogue, 1x1 + residual"""
    from sparkmi.bicodec import BLOCK_RESUNIT, run_block
    p = _sub(g, f"resunit{dil}")
    x = torch.from_numpy(g[f"resunit{dil}.x"])
    xs = _snake(x, p["L.block.0.alpha"])            # the vocoder leaves a block's leading Snake to the producer's epilogue
    y = run_block(BLOCK_RESUNIT, p, x.cuda(), xs.cuda(), dil=dil, exact_fp32=exact).cpu().numpy()
    np.testing.assert_allclose(y, g[f"resunit{dil}.y"], rtol=0, atol=TOL_EXACT if exact else TOL_SPLIT)


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("k,s", [(16, 8), (11, 5), (8, 4), (4, 2)])
def test_decoder_block_on_hip(g, k, s, exact):
    """sparktts/modules/encoder_decoder/wave_generator.py:29-53: polyphase ConvTranspose1d (every (kernel, stride) of the 0.5B
    decoder, k = 11 / s = 5 with two- and three-tap phases) + three ResidualUnits, the Snakes riding in the epilogues"""
    from sparkmi.bicodec import BLOCK_DECBLOCK, run_block
    p = _sub(g, f"decblock{k}_{s}")
    x = torch.from_numpy(g[f"decblock{k}_{s}.x"])
    xs = _snake(x, p["L.block.0.alpha"])
    y = run_block(BLOCK_DECBLOCK, p, None, xs.cuda(), K=k, S=s, exact_fp32=exact).cpu().numpy()
    want = g[f"decblock{k}_{s}.y"]
    assert y.shape == want.shape
    np.testing.assert_allclose(y, want, rtol=0, atol=2 * TOL_EXACT if exact else TOL_SPLIT)


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("tag", ["ln", "adaln"])
def test_convnext_block_on_hip(g, tag, exact):
    """sparktts/modules/blocks/vocos.py:26-110: depthwise conv7 + LayerNorm / AdaLayerNorm (k_dwln), pwconv1 + GELU, pwconv2 +
    layer scale + residual (k_convb on the default pipe: 48 and 112 channels)"""
    from sparkmi.bicodec import BLOCK_CONVNEXT, run_block
    p = _sub(g, f"convnext_{tag}")
    x = torch.from_numpy(g[f"convnext_{tag}.x"])
    cond = torch.from_numpy(g[f"convnext_{tag}.cond"]).cuda() if tag == "adaln" else None
    y = run_block(BLOCK_CONVNEXT, p, x.cuda(), None, cond, exact_fp32=exact).cpu().numpy()
    np.testing.assert_allclose(y, g[f"convnext_{tag}.y"], rtol=0, atol=TOL_EXACT if exact else TOL_SPLIT)


def test_ragged_rows_of_a_block_equal_solo_runs(g):
    """every kernel masks loads beyond a row's own length: row b of a ragged batch == that row run alone, bit for bit"""
    from sparkmi.bicodec import BLOCK_DECBLOCK, run_block
    p = _sub(g, "decblock8_4")
    x = torch.from_numpy(g["decblock8_4.x"])
    xs = _snake(x, p["L.block.0.alpha"]).cuda()
    L = xs.shape[-1]
    lens = [L, L - 5]
    y = run_block(BLOCK_DECBLOCK, p, None, xs, lens=lens, K=8, S=4).cpu().numpy()
    solo = run_block(BLOCK_DECBLOCK, p, None, xs[1:, :, : L - 5].contiguous(), K=8, S=4).cpu().numpy()
    np.testing.assert_array_equal(y[1, :, : (L - 5) * 4], solo[0])
    full = run_block(BLOCK_DECBLOCK, p, None, xs[:1].contiguous(), K=8, S=4).cpu().numpy()
    np.testing.assert_array_equal(y[0], full[0])


def test_block_entry_rejects_what_the_kernels_cannot_run(g):
    from sparkmi import _lib
    from sparkmi.bicodec import BLOCK_DECBLOCK, run_block
    p = _sub(g, "decblock8_4")
    xs = torch.zeros(1, 32, 13, device="cuda")
    with pytest.raises(_lib.SparkMIError):
        run_block(BLOCK_DECBLOCK, p, None, xs, K=9, S=4)      # (K - S) odd: no symmetric padding
    with pytest.raises(_lib.SparkMIError):
        run_block(BLOCK_DECBLOCK, p, None, xs, lens=[99], K=8, S=4)
