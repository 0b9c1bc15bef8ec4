"""Host-side checks of the one-row decode engine (csrc/smi_eng.h, smi_eng_host.h) that need no GPU: the static work plan --
which CU owns which 4-row part of each layer matrix, which wave runs which (part, chain set) job, and the order of the 1-KiB
weight images in every CU's stream.  smi_llm_engine_plan rebuilds the plan and checks that every image is placed exactly once
and that the stream order is the job order; here the counts are checked against the model's arithmetic."""
import ctypes as C

import pytest

from sparkmi import _lib, config as Cf
from sparkmi.arena import llm_cfg_struct


def _plan(cfg, ncu):
    cs = llm_cfg_struct(cfg, 1, 512, "bf16", True)
    st = (C.c_int32 * 8)()
    rc = _lib.diag().smi_llm_engine_plan(C.byref(cs), ncu, st)      # include/sparkmi_debug.h: the diagnostics build
    return rc, list(st), _lib.diag().smi_last_error().decode()


def _images_per_layer(cfg):
    """4-row parts x MFMA images per part, from the chain structure of the launch path's kernels (k tile kt -> chain kt mod NW):
    a chain set (4 chains) of a matrix with KT k tiles takes ceil((KT - 4a) / NW) images."""
    H, Q, KV, I = cfg.hidden_size, cfg.q_dim, cfg.kv_dim, cfg.intermediate_size
    nwo = cfg.num_attention_heads if cfg.num_attention_heads in (4, 14) else 8
    total = 0
    for rows, K, NW in (((Q + 2 * KV), H, 16), (H, Q, nwo), (2 * I, H, 8), (H, I, 16)):
        KT = K // 32
        per_part = sum(max(0, (KT - 1 - 4 * a) // NW + 1) if KT - 1 - 4 * a >= 0 else 0 for a in range((NW + 3) // 4))
        total += rows // 4 * per_part
    return total


@pytest.mark.parametrize("name,ncu", [("spark_0p5b_llm", 256), ("spark_0p5b_llm", 304), ("tiny_llm", 256), ("tiny_llm", 64)])
def test_plan_places_every_weight_image_once(name, ncu):
    cfg = getattr(Cf, name)()
    rc, st, err = _plan(cfg, ncu)
    assert rc == 0, err
    maxlen, wave_phase, slots, jobs, lo, hi, lds, total = st
    assert total == _images_per_layer(cfg)
    assert lds <= 160 * 1024 - 512            # one workgroup per CU
    assert wave_phase <= 14 and slots <= 20 and jobs <= 16
    assert lo <= hi == maxlen
    if name == "spark_0p5b_llm" and ncu == 256:
        # 29 344 images = 30.0 MB per layer (29.9 MB of weights + the zero padding of o_proj's last chain set);
        # no CU streams more than 123 KiB per layer, i.e. 5 us at the 25 GB/s one loader wave moves
        assert total == 29344 and hi <= 128


def test_plan_refuses_devices_it_does_not_fit():
    cfg = Cf.spark_0p5b_llm()
    for ncu in (16, 64, 128):                 # too few CUs: a phase's images of a CU would not fit its 112-KiB ring
        rc, _, err = _plan(cfg, ncu)
        assert rc != 0 and "engine plan" in err
