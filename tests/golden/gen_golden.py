#!/usr/bin/env python3
"""Generates the committed golden vectors in this directory.

Runs ONLY in the build container, where the upstream reference is mounted read-only at
/root/reference: it imports the reference's own modules (the vocoder half) and the
third-party ``transformers`` Qwen2 (the LLM half, which is what ``cli/SparkTTS.py:49,197``
runs), feeds them the build's deterministic synthetic weights, and stores inputs + outputs as
small ``.npz`` files.  Nothing of the reference travels: fixtures are data only.

    python tests/golden/gen_golden.py [--full]    # --full also regenerates the 0.5B-size vectors

The vocoder path is assembled exactly as ``BiCodec.detokenize`` does
(``sparktts/models/bicodec.py:183-187``; ``bicodec.py`` itself cannot be imported here because
omegaconf/torchaudio are absent -- an ordinary ModuleNotFoundError):
quantizer.detokenize -> speaker_encoder.detokenize(onnx_export_mode=True, the einx-free but
numerically identical branch) -> prenet(z_q, d) -> + d.unsqueeze(-1) -> decoder.
"""
from __future__ import annotations

import argparse
import os
import sys
import time
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "spark-tts_amd"))
sys.path.insert(0, "/root/reference")
warnings.filterwarnings("ignore")

from sparkmi import config as C  # noqa: E402
from sparkmi import weights as W  # noqa: E402
from sparkmi.pipeline_text import build_control_prompt, build_clone_prompt  # noqa: E402


def build_reference_vocoder(cfg: C.BiCodecConfig, sd_unfolded):
    from sparktts.modules.encoder_decoder.wave_generator import WaveGenerator
    from sparktts.modules.encoder_decoder.feat_decoder import Decoder
    from sparktts.modules.vq.factorized_vector_quantize import FactorizedVectorQuantize
    from sparktts.modules.speaker.speaker_encoder import SpeakerEncoder

    y = cfg.to_yaml_dict()["audio_tokenizer"]
    mods = torch.nn.ModuleDict(dict(
        quantizer=FactorizedVectorQuantize(**y["quantizer"]),
        prenet=Decoder(**y["prenet"]),
        decoder=WaveGenerator(**y["decoder"]),
        speaker_encoder=SpeakerEncoder(**y["speaker_encoder"]),
    ))
    state = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_unfolded.items()}
    res = mods.load_state_dict(state, strict=False)
    assert not res.unexpected_keys, f"synthetic keys unknown to the reference: {res.unexpected_keys}"
    # every reference parameter that detokenize touches must have been supplied
    touched = ("quantizer.codebook", "quantizer.out_project", "prenet.", "decoder.",
               "speaker_encoder.quantizer.project_out", "speaker_encoder.project.")
    miss = [k for k in res.missing_keys if k.startswith(touched)]
    assert not miss, f"detokenize parameters missing from the synthetic state: {miss}"
    mods.eval()

    def _rm(m):  # bicodec.py:213-221
        try:
            torch.nn.utils.remove_weight_norm(m)
        except ValueError:
            pass
    mods.apply(_rm)
    return mods


@torch.no_grad()
def reference_detokenize(mods, semantic, global_tokens, stages=None):
    """bicodec.py:183-187 on the reference's modules."""
    z_q = mods["quantizer"].detokenize(semantic)
    d = mods["speaker_encoder"].detokenize(global_tokens, onnx_export_mode=True)
    x = mods["prenet"](z_q, d)
    x = x + d.unsqueeze(-1)
    if stages is not None:
        stages.update(z_q=z_q, d_vector=d, prenet_plus_d=x)
        h = x
        outs = []
        for i, layer in enumerate(mods["decoder"].model):
            h = layer(h)
            if i <= len(mods["decoder"].model) - 4:   # conv_in and each DecoderBlock
                outs.append(h)
        stages["wavegen"] = outs
        return h
    return mods["decoder"](x)


def gen_vocoder(tag: str, cfg: C.BiCodecConfig, cases, seed=0, keep_stages=True):
    sd = W.bicodec_detok_state(cfg, seed=seed)
    mods = build_reference_vocoder(cfg, sd)
    # weight-norm fold parity: the reference's folded weights vs sparkmi.weights.fold_weight_norm
    folded = W.fold_weight_norm(sd)
    ref_sd = mods.state_dict()
    for k, v in folded.items():
        assert np.array_equal(ref_sd[k].numpy(), v), f"weight-norm fold differs at {k}"
    out = {}
    for ci, (T, sseed) in enumerate(cases):
        rng = np.random.Generator(np.random.PCG64(sseed))
        sem = rng.integers(0, cfg.codebook_size, size=(1, T), dtype=np.int64)
        glob = rng.integers(0, int(np.prod(cfg.fsq_levels)), size=(1, 1, cfg.spk_token_num), dtype=np.int64)
        stages = {}
        t0 = time.time()
        wav = reference_detokenize(mods, torch.from_numpy(sem), torch.from_numpy(glob), stages)
        dt = time.time() - t0
        out[f"c{ci}_semantic"] = sem
        out[f"c{ci}_global"] = glob
        out[f"c{ci}_wav"] = wav.numpy()
        out[f"c{ci}_d_vector"] = stages["d_vector"].numpy()
        if keep_stages and T <= 40:
            out[f"c{ci}_z_q"] = stages["z_q"].numpy()
            out[f"c{ci}_prenet_plus_d"] = stages["prenet_plus_d"].numpy()
            for si, s in enumerate(stages["wavegen"]):
                out[f"c{ci}_wavegen{si}"] = s.numpy()
        else:
            # full size / long cases: keep per-stage statistics instead of the tensors
            st = [stages["z_q"], stages["prenet_plus_d"]] + stages["wavegen"]
            out[f"c{ci}_stage_sum"] = np.array([float(s.double().sum()) for s in st])
            out[f"c{ci}_stage_abs"] = np.array([float(s.double().abs().sum()) for s in st])
        print(f"[{tag}] case {ci}: T={T} wav{tuple(wav.shape)} std={float(wav.std()):.4f} "
              f"max|.|={float(wav.abs().max()):.4f}  ({dt:.2f}s)")
    np.savez_compressed(os.path.join(HERE, f"vocoder_{tag}.npz"), **out)


def gen_ops(seed=0):
    """Per-op vectors at reduced dims straight from the reference's layer classes."""
    from sparktts.modules.blocks.layers import snake as ref_snake, ResidualUnit
    from sparktts.modules.encoder_decoder.wave_generator import DecoderBlock
    from sparktts.modules.blocks.vocos import ConvNeXtBlock
    from sparktts.modules.blocks.samper import SamplingBlock
    from sparktts.modules.fsq.finite_scalar_quantization import FSQ
    out = {}
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(2, 16, 50, generator=g)
    a = 1.0 + 0.2 * torch.randn(1, 16, 1, generator=g)
    out["snake_x"], out["snake_alpha"], out["snake_y"] = x.numpy(), a.numpy(), ref_snake(x, a).numpy()
    sb = SamplingBlock(dim=16, groups=16, upsample_scale=1)
    xin = torch.randn(2, 50, 16, generator=g)
    out["sampling_x"], out["sampling_y"] = xin.numpy(), sb(xin).detach().numpy()
    fsq = FSQ(levels=[4] * 6)
    out["fsq_codebook"] = fsq.implicit_codebook.numpy()
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **out)
    print("[ops] snake / SamplingBlock(ratio 1) / FSQ implicit codebook stored")


def _randomize(mod, g):
    """Seeded non-trivial parameters for a reference layer: every tensor ~ N(0, 0.1), gains (Snake alpha, layer scale,
    weight-norm g, LayerNorm weight) ~ 1 + N(0, 0.2)."""
    with torch.no_grad():
        for n, p in mod.named_parameters():
            r = torch.randn(p.shape, generator=g)
            if n.endswith(("alpha", "gamma", "weight_g")) or ("norm" in n and n.endswith("weight")):
                p.copy_(1.0 + 0.2 * r)
            else:
                p.copy_(0.1 * r)


def _folded_state(mod):
    """state_dict after remove_weight_norm (bicodec.py:213-221), as numpy."""
    def _rm(m):
        try:
            torch.nn.utils.remove_weight_norm(m)
        except ValueError:
            pass
    mod.apply(_rm)
    return {k: v.detach().numpy().copy() for k, v in mod.state_dict().items()}


def gen_ops_layers(seed=0):
    """SURVEY 8c's per-op fixtures, straight from the reference's layer classes at reduced dims (parameters, input, output):
    ResidualUnit (dilations 1 / 3 / 9), DecoderBlock for each (k, s) of the 0.5B decoder, ConvNeXtBlock with LayerNorm and
    with AdaLayerNorm -- and, from transformers (the LLM's arithmetic), RMSNorm, RoPE, one GQA attention step and the
    SwiGLU MLP.  tests/test_oracle_ops.py checks the oracle's functions against them one by one."""
    from sparktts.modules.blocks.layers import ResidualUnit
    from sparktts.modules.encoder_decoder.wave_generator import DecoderBlock
    from sparktts.modules.blocks.vocos import ConvNeXtBlock
    out = {}
    g = torch.Generator().manual_seed(seed + 100)

    def put(prefix, sd):
        for k, v in sd.items():
            out[f"{prefix}/{k}"] = v

    with torch.no_grad():
        for dil in (1, 3, 9):
            m = ResidualUnit(dim=24, dilation=dil).eval()
            _randomize(m, g)
            x = torch.randn(2, 24, 61, generator=g)
            y = m(x)
            put(f"resunit{dil}", _folded_state(m))
            out[f"resunit{dil}.x"], out[f"resunit{dil}.y"] = x.numpy(), y.numpy()
        for k, s in ((16, 8), (11, 5), (8, 4), (4, 2)):
            m = DecoderBlock(input_dim=32, output_dim=16, kernel_size=k, stride=s).eval()
            _randomize(m, g)
            x = torch.randn(2, 32, 13, generator=g)
            y = m(x)
            put(f"decblock{k}_{s}", _folded_state(m))
            out[f"decblock{k}_{s}.x"], out[f"decblock{k}_{s}.y"] = x.numpy(), y.numpy()
        for tag, cond_dim in (("ln", None), ("adaln", 20)):
            m = ConvNeXtBlock(dim=48, intermediate_dim=112, layer_scale_init_value=1.0, condition_dim=cond_dim).eval()
            _randomize(m, g)
            x = torch.randn(2, 48, 29, generator=g)
            cond = torch.randn(2, 20, generator=g) if cond_dim else None
            y = m(x, cond)
            put(f"convnext_{tag}", _folded_state(m))
            out[f"convnext_{tag}.x"], out[f"convnext_{tag}.y"] = x.numpy(), y.numpy()
            if cond is not None:
                out[f"convnext_{tag}.cond"] = cond.numpy()
        # ---- LLM ops from transformers' Qwen2 classes
        from transformers import Qwen2Config
        from transformers.models.qwen2 import modeling_qwen2 as MQ
        hc = Qwen2Config(vocab_size=64, hidden_size=128, intermediate_size=224, num_hidden_layers=1, num_attention_heads=4,
                         num_key_value_heads=2, rms_norm_eps=1e-6, rope_theta=1000000.0, max_position_embeddings=512,
                         attn_implementation="eager")
        norm = MQ.Qwen2RMSNorm(128, eps=1e-6)
        _randomize(norm, g)
        x = torch.randn(1, 7, 128, generator=g)
        out["rmsnorm.w"], out["rmsnorm.x"], out["rmsnorm.y"] = norm.weight.detach().numpy().copy(), x[0].numpy(), norm(x)[0].numpy()
        rot = MQ.Qwen2RotaryEmbedding(config=hc)
        pos = torch.tensor([[3, 4, 5, 130, 131]])
        q = torch.randn(1, 4, 5, 32, generator=g)
        kx = torch.randn(1, 2, 5, 32, generator=g)
        cos, sin = rot(q, pos)
        qr, kr = MQ.apply_rotary_pos_emb(q, kx, cos, sin)
        out["rope.pos"], out["rope.q"], out["rope.k"] = pos[0].numpy(), q[0].numpy(), kx[0].numpy()
        out["rope.q_rot"], out["rope.k_rot"] = qr[0].numpy(), kr[0].numpy()
        # one decode step of GQA attention: 1 query position (index 9) against 10 cached keys, 4 query / 2 kv heads
        qa = torch.randn(1, 4, 1, 32, generator=g)
        ka, va = torch.randn(1, 2, 10, 32, generator=g), torch.randn(1, 2, 10, 32, generator=g)

        class _M:   # what eager_attention_forward reads from the module
            num_key_value_groups = 2
            training = False
        ao, _ = MQ.eager_attention_forward(_M(), qa, ka, va, attention_mask=None, scaling=32 ** -0.5, dropout=0.0)
        out["attn.q"], out["attn.k"], out["attn.v"], out["attn.y"] = qa[0].numpy(), ka[0].numpy(), va[0].numpy(), ao[0].numpy()
        mlp = MQ.Qwen2MLP(hc)
        _randomize(mlp, g)
        xm = torch.randn(3, 128, generator=g)
        out["mlp.x"], out["mlp.y"] = xm.numpy(), mlp(xm).numpy()
        for k2, v2 in mlp.state_dict().items():
            out[f"mlp/{k2}"] = v2.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "ops_layers.npz"), **out)
    print(f"[ops_layers] {len(out)} arrays: ResidualUnit x3, DecoderBlock x4, ConvNeXtBlock x2, RMSNorm, RoPE, GQA step, SwiGLU")


def gen_extra(seed=0):
    """The last two fixtures SURVEY 8c lists: (1) the vocoder on a B = 4 batch through the reference's own modules (tiny
    config; equal lengths -- the reference has no length masks, a padded row would see its padding), (2) per-layer
    hidden-state checksums of the tiny LLM from transformers (output_hidden_states)."""
    out = {}
    cfg = C.tiny_bicodec()
    sd = W.bicodec_detok_state(cfg, seed=seed)
    mods = build_reference_vocoder(cfg, sd)
    rng = np.random.Generator(np.random.PCG64(4444))
    sem = rng.integers(0, cfg.codebook_size, size=(4, 19), dtype=np.int64)
    glob = rng.integers(0, int(np.prod(cfg.fsq_levels)), size=(4, 1, cfg.spk_token_num), dtype=np.int64)
    with torch.no_grad():
        wav = reference_detokenize(mods, torch.from_numpy(sem), torch.from_numpy(glob))
    out["voc4_semantic"], out["voc4_global"], out["voc4_wav"] = sem, glob, wav.numpy()
    lcfg = C.tiny_llm()
    syn = W.SyntheticLLM(lcfg, seed=seed)
    m = hf_model(lcfg, syn)
    prompt = np.random.Generator(np.random.PCG64(1234)).integers(0, lcfg.vocab_size, size=(21,), dtype=np.int64)
    with torch.no_grad():
        o = m(torch.from_numpy(prompt)[None], output_hidden_states=True)
    hs = o.hidden_states          # embeddings, then the output of every layer (the last one after the final norm in HF >= 4.x)
    out["llm_prompt"] = prompt
    out["llm_hidden_sum"] = np.array([float(h.double().sum()) for h in hs])
    out["llm_hidden_abs"] = np.array([float(h.double().abs().sum()) for h in hs])
    out["llm_hidden_first"] = np.stack([h[0, -1, :8].numpy() for h in hs])       # 8 values of the last position, per layer
    np.savez_compressed(os.path.join(HERE, "extra.npz"), **out)
    print(f"[extra] vocoder B=4 wav{tuple(wav.shape)}; {len(hs)} hidden-state checksums")


def hf_model(cfg: C.LLMConfig, syn: W.SyntheticLLM):
    from transformers import Qwen2Config, Qwen2ForCausalLM
    hc = Qwen2Config(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size,
                     intermediate_size=cfg.intermediate_size, num_hidden_layers=cfg.num_hidden_layers,
                     num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads,
                     rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
                     tie_word_embeddings=cfg.tie_word_embeddings,
                     max_position_embeddings=cfg.max_position_embeddings, use_sliding_window=False,
                     attn_implementation="eager")
    with torch.device("meta"):
        m = Qwen2ForCausalLM(hc)
    m = m.to_empty(device="cpu")
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n == "lm_head.weight" and cfg.tie_word_embeddings:
                continue
            p.copy_(torch.from_numpy(syn[n]))
    m.tie_weights()
    # non-persistent buffers (rotary inv_freq) are not restored by to_empty: rebuild them
    for mod in m.modules():
        if hasattr(mod, "inv_freq") and hasattr(mod, "rope_init_fn") is False and hasattr(mod, "compute_default_rope_parameters"):
            inv, _ = mod.compute_default_rope_parameters(mod.config)
            mod.inv_freq = inv
            mod.original_inv_freq = inv.clone()
    return m.eval()


def gen_llm(tag: str, cfg: C.LLMConfig, prompt_len: int, new_tokens: int, pseed: int, seed=0):
    syn = W.SyntheticLLM(cfg, seed=seed)
    t0 = time.time()
    m = hf_model(cfg, syn)
    rng = np.random.Generator(np.random.PCG64(pseed))
    prompt = rng.integers(0, cfg.vocab_size, size=(prompt_len,), dtype=np.int64)
    ids = torch.from_numpy(prompt)[None]
    with torch.no_grad():
        logits = m(ids).logits[0]                     # (P, V)
        gen = m.generate(ids, attention_mask=torch.ones_like(ids), max_new_tokens=new_tokens,
                         do_sample=False, eos_token_id=None, pad_token_id=0)[0, prompt_len:]
    last = logits[-1]
    top = torch.topk(last, 64)
    out = dict(prompt=prompt, greedy=gen.numpy().astype(np.int64),
               last_top_ids=top.indices.numpy().astype(np.int64), last_top_vals=top.values.numpy(),
               last_logits_sum=np.array(float(last.double().sum())),
               last_logits_abs=np.array(float(last.double().abs().sum())))
    if cfg.vocab_size * prompt_len <= 4_000_000:
        out["logits"] = logits.numpy()
    np.savez_compressed(os.path.join(HERE, f"llm_{tag}.npz"), **out)
    gap = float(top.values[0] - top.values[1])
    print(f"[llm {tag}] P={prompt_len} N={new_tokens} first tokens {gen[:8].tolist()} "
          f"top1-top2 gap {gap:.4f} unique={len(set(gen.tolist()))} ({time.time() - t0:.1f}s)")


def gen_prompts():
    """Prompt-string goldens: the reference's own two builders (cli/SparkTTS.py:53-155) cannot be
    imported (omegaconf), so the expected strings are written out from the token tables the
    reference does export (sparktts/utils/token_parser.py) and the documented layout."""
    from sparktts.utils.token_parser import TASK_TOKEN_MAP, LEVELS_MAP, GENDER_MAP
    import json
    cases = []
    for gender in GENDER_MAP:
        for pitch in LEVELS_MAP:
            for speed in ("very_low", "moderate", "very_high"):
                text = f"hello {gender} {pitch} {speed}"
                exp = "".join([TASK_TOKEN_MAP["controllable_tts"], "<|start_content|>", text, "<|end_content|>",
                               "<|start_style_label|>", f"<|gender_{GENDER_MAP[gender]}|>",
                               f"<|pitch_label_{LEVELS_MAP[pitch]}|>", f"<|speed_label_{LEVELS_MAP[speed]}|>",
                               "<|end_style_label|>"])
                assert build_control_prompt(gender, pitch, speed, text) == exp
                cases.append(dict(kind="control", gender=gender, pitch=pitch, speed=speed, text=text, expect=exp))
    glob, sem = [7, 4095, 0, 12], [3, 8191, 44]
    g = "".join(f"<|bicodec_global_{i}|>" for i in glob)
    s = "".join(f"<|bicodec_semantic_{i}|>" for i in sem)
    e1 = "".join([TASK_TOKEN_MAP["tts"], "<|start_content|>", "PT", "TXT", "<|end_content|>",
                  "<|start_global_token|>", g, "<|end_global_token|>", "<|start_semantic_token|>", s])
    e2 = "".join([TASK_TOKEN_MAP["tts"], "<|start_content|>", "TXT", "<|end_content|>",
                  "<|start_global_token|>", g, "<|end_global_token|>"])
    assert build_clone_prompt("TXT", glob, sem, "PT") == e1 and build_clone_prompt("TXT", glob, sem, None) == e2
    cases.append(dict(kind="clone", text="TXT", prompt_text="PT", glob=glob, sem=sem, expect=e1))
    cases.append(dict(kind="clone", text="TXT", prompt_text=None, glob=glob, sem=sem, expect=e2))
    with open(os.path.join(HERE, "prompts.json"), "w") as f:
        json.dump(dict(cases=cases, TASK_TOKEN_MAP=TASK_TOKEN_MAP, LEVELS_MAP=LEVELS_MAP, GENDER_MAP=GENDER_MAP), f, indent=1)
    print(f"[prompts] {len(cases)} prompt strings stored")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    torch.manual_seed(0)
    want = lambda k: not a.only or k in a.only.split(",")  # noqa: E731
    if want("ops"):
        gen_ops()
    if want("ops_layers"):
        gen_ops_layers()
    if want("extra"):
        gen_extra()
    if want("prompts"):
        gen_prompts()
    if want("voc"):
        # ragged set: lengths incl. 1 frame, a non-multiple of any tile, and a longer one
        gen_vocoder("tiny", C.tiny_bicodec(), [(37, 11), (1, 12), (8, 13), (150, 14)])
    if want("llm"):
        gen_llm("tiny", C.tiny_llm(), prompt_len=21, new_tokens=40, pseed=1234)
    if a.full:
        if want("voc"):
            gen_vocoder("full", C.spark_0p5b_bicodec(), [(150, 1235), (23, 1236)], keep_stages=False)
        if want("llm"):
            gen_llm("full", C.spark_0p5b_llm(), prompt_len=128, new_tokens=150, pseed=1234)
