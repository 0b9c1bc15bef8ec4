#!/usr/bin/env python3
"""Spark-TTS-0.5B greedy synthesis throughput on MI355X (BASELINE.json metric: audio samples/s
for the whole node + real-time factor).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one batch of `--batch` utterances through the whole hot path on one GPU:
128-token synthetic prompt -> prefill -> 149 greedy decode steps (150 new tokens, EOS
suppressed) -> host token parse (id mod codebook size, synthetic weights) -> BiCodec vocoder
(150 frames -> 48 000 samples = 3.0 s at 16 kHz).  Weights are synthetic (no checkpoint exists
offline), generated on rank 0 and broadcast over RCCL; inputs are already in HBM when the timed
region starts (prompt ids are 1 KiB of host data per utterance, passed by value at prefill).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant decode kernel, measured live with HIP
events on the launch stream; `cpu_baseline` times the CPU oracle (a port of the reference's
PyTorch-CPU arithmetic) on this box's host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "spark-tts_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1, help="concurrent utterances per GPU (BASELINE config 3 uses 32)")
    ap.add_argument("--prompt-len", type=int, default=128)
    ap.add_argument("--new-tokens", type=int, default=150)
    ap.add_argument("--kv", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probes", action="store_true")
    ap.add_argument("--cpu-tokens", type=int, default=32, help="decode tokens in the bounded CPU sample")
    ap.add_argument("--clone", action="store_true",
                    help="voice-clone path (BASELINE configs[4]): every utterance first encodes a prompt wav on the GPU "
                         "(wav2vec2 + BiCodec encoder + speaker encoder), its tokens join the LLM prompt; use with --batch 8")
    ap.add_argument("--prompt-seconds", type=float, default=6.0, help="--clone: length of each synthetic prompt wav")
    return ap.parse_args()


def host_cores() -> int:
    """CPU cores this process may actually use (affinity mask and cgroup quota, not the box total)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(llm_cfg, voc_cfg, prompt, glob, n_tokens):
    """The oracle (CPU restatement pinned to the reference by tests/golden) on the host cores."""
    from oracle.llm_ref import Qwen2Ref
    from oracle.bicodec_ref import BiCodecDetokRef
    from sparkmi import weights as W
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: building fp32 oracle weights ({cores} cores)")
    t0 = time.time()
    ref = Qwen2Ref(llm_cfg, W.SyntheticLLM(llm_cfg))
    voc = BiCodecDetokRef(voc_cfg, W.fold_weight_norm(W.bicodec_detok_state(voc_cfg)))
    build_s = time.time() - t0
    log(f"cpu_baseline: weights ready in {build_s:.1f}s; generating {n_tokens} tokens")
    t0 = time.time()
    toks = ref.generate_greedy(prompt, n_tokens)
    t_llm = time.time() - t0
    sem = torch.tensor([[t % voc_cfg.codebook_size for t in toks]])
    t0 = time.time()
    wav = voc.detokenize(sem, torch.as_tensor(glob)[None, None])
    t_voc = time.time() - t0
    samples = wav.shape[-1]
    total = t_llm + t_voc
    return {
        "value": samples / total, "unit": "audio samples/s", "cores": cores, "kind": "port",
        "rtf": total / (samples / 16000.0),
        "sample": f"1 utterance: {len(prompt)}-token prefill + {n_tokens} greedy tokens ({t_llm:.2f}s) + vocoder of "
                  f"those {n_tokens} frames ({t_voc:.2f}s), fp32 torch-CPU oracle, weights build {build_s:.1f}s excluded",
        "cpu_model": _cpu_model(), "first_tokens": toks[:8],
    }


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist
    from sparkmi import config as C, weights as W, arena as A, bicodec as BC, _lib
    from sparkmi.llm import SparkLLM
    from sparkmi.bicodec import BiCodecVocoder
    from sparkmi import dist as SD

    # rehearsal on a one-GPU box: SPARKMI_ONE_GPU=1 maps every rank to cuda:0 and uses gloo (RCCL refuses
    # two ranks on one device); the driver's real multi-GPU runs use one GPU per rank over RCCL/xGMI
    one_gpu = os.environ.get("SPARKMI_ONE_GPU") == "1"
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    arch = _lib.require_gfx950()
    torch.set_num_threads(host_cores())
    if rank == 0:
        log(f"device {arch}, world {world}, host cores {host_cores()}: building synthetic weights")

    llm_cfg, voc_cfg = C.spark_0p5b_llm(), C.spark_0p5b_bicodec()
    B, P, N = a.batch, a.prompt_len, a.new_tokens
    max_pos = P + N + 80
    if a.clone:
        from sparkmi import config_tok as T
        from sparkmi.encoder import BiCodecEncoder, get_ref_clip
        wcfg, tcfg = T.xlsr53(), T.spark_0p5b_tok()
        max_pos += wcfg.frames(int(16000 * a.prompt_seconds)) + tcfg.spk_token_num
    cs_llm = A.llm_cfg_struct(llm_cfg, B, max_pos, a.kv, not a.no_graph)
    cs_voc = BC.voc_cfg_struct(voc_cfg, B, N + 10)

    # ---- weights: rank 0 builds the arenas, everyone else receives them over RCCL/xGMI
    t0 = time.time()
    if rank == 0:
        llm_arena = torch.from_numpy(A.pack_llm_arena(llm_cfg, W.SyntheticLLM(llm_cfg), cs_llm)).to(dev)
        voc_arena = torch.from_numpy(BC.pack_voc_arena(
            voc_cfg, W.fold_weight_norm(W.bicodec_detok_state(voc_cfg)), cs_voc)).to(dev)
    else:
        llm_arena = voc_arena = None
    t_build = time.time() - t0
    llm_arena, voc_arena, bcast_ms = SD.broadcast_arenas(
        llm_arena, voc_arena, SD.arena_sizes(cs_llm, cs_voc), dev, rank, world)

    llm = SparkLLM(llm_cfg, None, dev, max_slots=B, max_positions=max_pos, kv_dtype=a.kv,
                   use_graph=not a.no_graph, arena=llm_arena)
    voc = BiCodecVocoder(voc_cfg, None, dev, max_batch=B, max_frames=N + 10, arena=voc_arena)
    enc = None
    if a.clone:
        if rank == 0:
            log("building the prompt encoder (wav2vec2-large-xlsr-53 shape, 16 layers + BiCodec tokenizer side)")
        # every rank builds its own copy of the 1 GB encoder arena from the same seeds (no second broadcast path)
        enc = BiCodecEncoder(wcfg, tcfg, W.fold_pos_conv_weight_norm(W.wav2vec2_state(wcfg)),
                             W.fold_weight_norm(W.bicodec_tok_state(tcfg, voc_cfg.vq_input_dim)), dev,
                             max_seconds=a.prompt_seconds, ref_seconds=6.0)
        tt = np.arange(int(16000 * a.prompt_seconds)) / 16000.0
        pwavs, prefs = [], []
        for i in range(B):
            f0 = 100 + 15 * i + 30 * np.sin(2 * np.pi * 0.7 * tt + i)
            ph = 2 * np.pi * np.cumsum(f0) / 16000.0
            x = (0.3 * (0.5 + 0.5 * np.sin(2 * np.pi * 2.3 * tt) ** 2) * (0.5 * np.sin(ph) + 0.3 * np.sin(2 * ph) + 0.2 * np.sin(3 * ph))
                 + 0.02 * np.random.Generator(np.random.PCG64(4000 + rank * 100 + i)).standard_normal(len(tt))).astype(np.float32)
            pwavs.append(x)
            prefs.append(get_ref_clip(x, 16000, 6.0, 320).astype(np.float32))

    # ---- synthetic inputs: prompt ids ~ U[0,V) PCG64(1234+i), global ids PCG64(1235+i)
    prompts, globs = [], []
    for i in range(B):
        s = rank * 1000 + i
        prompts.append(np.random.Generator(np.random.PCG64(1234 + 2 * s)).integers(0, llm_cfg.vocab_size, size=P).tolist())
        globs.append(np.random.Generator(np.random.PCG64(1235 + 2 * s)).integers(0, 4096, size=voc_cfg.spk_token_num))
    glob_t = torch.from_numpy(np.stack(globs)).to(dev, torch.int32).unsqueeze(1)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    decode_ms, voc_ms, enc_ms = [], [], []

    def step(timed=False):
        nonlocal glob_t
        pr = prompts
        if enc is not None:
            # prompt encode: (global, semantic) ids of each prompt wav, appended to the text prompt the way
            # process_prompt does (cli/SparkTTS.py:83-104); ids are mapped into the synthetic vocabulary
            t0 = time.perf_counter()
            toks = [enc.tokenize_arrays(pwavs[i], prefs[i]) for i in range(B)]
            glob_t = torch.cat([g for g, _ in toks], 0)
            sem_h = [s_.reshape(-1).cpu().numpy() for _, s_ in toks]
            g_h = glob_t.reshape(B, -1).cpu().numpy()
            pr = [prompts[i] + (g_h[i] % llm_cfg.vocab_size).tolist() + (sem_h[i] % llm_cfg.vocab_size).tolist() for i in range(B)]
            if timed:
                enc_ms.append((time.perf_counter() - t0) * 1e3)
        llm.prefill(pr, None)
        if timed:
            ev[0].record()
        llm.decode(N - 1)
        if timed:
            ev[1].record()
        toks = llm.tokens(N)                                 # sync + D2H: the host parses ids like the reference
        sem = torch.tensor(toks, dtype=torch.long) % voc_cfg.codebook_size
        if timed:
            ev[2].record()
        wav = voc.detokenize(sem.to(dev), glob_t)
        if timed:
            ev[3].record()
        out = wav.cpu()                                      # 192 KB per utterance back to the host
        if timed:
            decode_ms.append(ev[0].elapsed_time(ev[1]))
            voc_ms.append(ev[2].elapsed_time(ev[3]))
        return out, toks

    if rank == 0:
        log(f"weights packed in {t_build:.1f}s, broadcast {bcast_ms:.1f} ms; warm-up")
    for _ in range(a.warmup):
        wav, toks = step()
    if rank == 0:
        log("timed region")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        wav, toks = step(timed=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device="cpu" if one_gpu else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    samples_per_step = B * N * voc_cfg.hop
    value = world * a.steps * samples_per_step / el
    audio_s = world * a.steps * samples_per_step / 16000.0
    res = {
        "metric": "audio samples/sec (whole node), Spark-TTS-0.5B greedy", "value": value, "unit": "audio samples/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1000.0 * el / a.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16 weights + KV, fp32 activations/accumulate (LLM); fp32 (vocoder)" if a.kv == "bf16"
                 else "bf16 weights, fp32 KV/activations (LLM); fp32 (vocoder)",
        "data": "synthetic (seeded prompts and weights; no checkpoint or dataset offline)",
        "config": {"workload": f"Spark-TTS-0.5B, batch={B} greedy, {P}-token prompt -> {N} tokens -> "
                               f"{N * voc_cfg.hop / 16000.0:.1f} s audio per utterance (BASELINE.json configs[{4 if a.clone else (1 if B == 1 else 2)}])",
                   **({"voice_clone": f"each utterance encodes a {a.prompt_seconds:.1f} s prompt wav on the GPU first (BASELINE.json configs[4])"} if a.clone else {}),
                   "batch_per_gpu": B, "prompt_len": P, "new_tokens": N, "kv_cache": a.kv,
                   "hipgraph": not a.no_graph, "parallelism": f"utterance-parallel x{world}", "device": arch},
        "rtf": el / audio_s, "x_realtime": audio_s / el,
        "utterances_per_s": world * a.steps * B / el,
        "weights": {"build_s_rank0": t_build, "rccl_broadcast_ms": bcast_ms,
                    "llm_arena_bytes": int(llm_arena.numel()), "voc_arena_bytes": int(voc_arena.numel()) * 4},
        "stage_ms": {"decode_149_steps": float(np.median(decode_ms)), "vocoder": float(np.median(voc_ms)),
                     **({"prompt_encode_all": float(np.median(enc_ms))} if enc_ms else {})},
        "first_tokens": toks[0][:8], "wav_std": float(wav.std()),
    }

    if rank == 0:
        log(f"timed: {el:.3f}s for {a.steps} steps -> {value:.0f} samples/s; probing kernels")
        # ---- roofline: decode step is HBM-bound; algorithmic bytes = weights once + KV read + KV write
        wb = llm.step_weight_bytes()
        kvb = llm.kv_bytes_per_token()
        ctx_sum = sum(P + j for j in range(1, N))   # decode step j reads positions 0..P+j-1 (+ its own)
        step_bytes = wb + kvb * B * (ctx_sum / (N - 1) + 1) + kvb * B
        dec_step_ms = res["stage_ms"]["decode_149_steps"] / (N - 1)
        res["roofline_step"] = {
            "bound": "hbm", "achieved": step_bytes / (dec_step_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": step_bytes / (dec_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
            "bytes_per_step": step_bytes, "ms_per_step": dec_step_ms,
            "note": "whole decode step (122 kernels in one hipGraph), events inside the timed region"}
        if not a.no_probes:
            llm.prefill(prompts, None)
            llm.decode(N // 2)
            ctx = P + N // 2
            kb = llm.weight_bytes()
            per = {"qkv": kb["qkv"] + kvb // llm_cfg.num_hidden_layers * B,
                   "attn": kvb // llm_cfg.num_hidden_layers * B * ctx,
                   "o_proj": kb["o_proj"], "gate_up": kb["gate_up"], "down": kb["down"], "lm_head": kb["lm_head"]}
            count = {k: llm_cfg.num_hidden_layers for k in per}
            count["lm_head"] = 1
            ks = []
            for name in ("qkv", "attn", "o_proj", "gate_up", "down", "lm_head"):
                # layer kernels are timed where they run: (96 layers captured in a hipGraph) - (the same graph without
                # the kernel), so each finds the L2 state its producers leave (idle CUs prefetch part of the next
                # kernels' weights) and no host launch rate enters; lm_head (50 us) is looped on its own
                ms = llm.time_kernel(name, iters=96, in_sequence=True) if name != "lm_head" else llm.time_kernel(name, iters=24)
                ks.append({"kernel": name, "launches_per_step": count[name], "avg_us": ms * 1e3,
                           "bytes": per[name], "GBps": per[name] / (ms * 1e-3) / 1e9,
                           "us_per_step": ms * 1e3 * count[name]})
            dom = max(ks, key=lambda k: k["us_per_step"])
            # HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes on this build and shape
            # (profiles/r01_pmc_hbm_traffic_v13.txt; FETCH_SIZE doubled per the gfx950 rule); null for other shapes
            # (profiles/r01_pmc_hbm_traffic_v13.txt: a producer's helper-block prefetch is part of ITS launch's traffic; gate_up's own
            # in-graph traffic cannot be observed -- counter collection runs every kernel alone with the L2 invalidated)
            pmc = {"gate_up": None, "down": 11.17e6, "qkv": 10.95e6, "o_proj": 1.77e6, "lm_head": 297.63e6, "attn": 9.61e6}
            std = B == 1 and P == 128 and a.kv == "bf16"
            res["roofline"] = {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["GBps"], "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": dom["GBps"] / HBM_PEAK_GBS,
                               "traffic": pmc.get(dom["kernel"]) if std else None,
                               "bytes_per_launch": dom["bytes"], "avg_us": dom["avg_us"]}
            res["kernels"] = ks
            # the reference's default mode (temperature 0.8 / top-k 50 / top-p 0.95): decode step with the sampler in the graph
            llm.set_sampling(True, 0.8, 50, 0.95, 1234)
            llm.prefill(prompts, None)
            llm.decode(8)
            res["sampling_step_us"] = llm.time_kernel("step", iters=64) * 1e3
            llm.set_sampling(False)
            llm.prefill(prompts, None)
            llm.decode(8)
            res["greedy_step_us"] = llm.time_kernel("step", iters=64) * 1e3
            if B == 1:
                # streaming mode: wall time to the first 1.0 s chunk on the host, and to all chunks
                from sparkmi.streaming import ChunkScheduler
                sched = ChunkScheduler()
                torch.cuda.synchronize()
                ts, t_first, done_toks, nchunks, seen = time.perf_counter(), None, 1, 0, 0
                llm.prefill(prompts, None)
                while True:
                    tk = llm.tokens(N)[0]
                    ready = sched.push(tk[seen:])
                    seen = len(tk)
                    last = done_toks >= N
                    if last:
                        ready += sched.flush()
                    for ch in ready:
                        w = voc.detokenize((torch.tensor([ch], dtype=torch.long) % voc_cfg.codebook_size).to(dev), glob_t,
                                           lengths=[len(ch)]).cpu()
                        nchunks += 1
                        if t_first is None:
                            t_first = time.perf_counter() - ts
                    if last:
                        break
                    n = min(10, N - done_toks)
                    llm.decode(n)
                    done_toks += n
                res["streaming"] = {"first_chunk_ms": 1e3 * t_first, "all_chunks_ms": 1e3 * (time.perf_counter() - ts),
                                    "chunks": nchunks, "schedule": "1.0 s first chunk, x8 growth, 0.1 s overlap (reference defaults)"}
            # vocoder launches (MFMA-bound side of the path)
            voc.detokenize(torch.tensor(toks, dtype=torch.long).to(dev) % voc_cfg.codebook_size, glob_t)
            vl = []
            for i in range(voc.launches()):
                nm, ms, fl = voc.time_launch(i, iters=5)
                vl.append((nm, ms, fl))
            tot_ms = sum(m for _, m, _ in vl)
            tot_fl = sum(f for _, _, f in vl)
            res["vocoder_mfma"] = {"launches": len(vl), "sum_ms": tot_ms, "gflop": tot_fl / 1e9,
                                   "achieved_TFLOPs": tot_fl / (tot_ms * 1e-3) / 1e12, "peak_TFLOPs": 157.3,
                                   "top": [{"name": n, "ms": m, "TFLOPs": f / (m * 1e-3) / 1e12 if m > 0 else 0}
                                           for n, m, f in sorted(vl, key=lambda x: -x[1])[:6]]}
        else:
            res["roofline"] = dict(res["roofline_step"])
        if not a.no_cpu_baseline and world == 1:
            log("gpu side done; timing the CPU oracle")
            res["cpu_baseline"] = cpu_baseline(llm_cfg, voc_cfg, prompts[0], globs[0], a.cpu_tokens)
            res["gpu_over_cpu_rtf"] = res["cpu_baseline"]["rtf"] / res["rtf"]
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
