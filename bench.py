#!/usr/bin/env python3
"""Spark-TTS-0.5B greedy synthesis throughput on MI355X (BASELINE.json metric: audio samples/s
for the whole node + real-time factor).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one batch of `--batch` utterances through the whole hot path on one GPU: synthetic prompt
-> prefill -> greedy decode (EOS suppressed) -> host token parse (id mod codebook size, synthetic
weights) -> BiCodec vocoder.  Workloads:

  --batch 1  (default; BASELINE configs[1])  128-token prompt -> 150 tokens -> 48 000 samples (3.0 s)
  --batch 32 (configs[2])  ragged: prompt lengths U{96..160}, N_i U{120..180} tokens per utterance, seeds
             2000+i.  Default: in-flight retirement with KNOWN budgets -- row i leaves the batch exactly after
             its N_i tokens (SparkLLM.generate_ragged, no eos, no status polling), the step runs on the live
             rows only: the best case of in-flight batching; eos-driven serving (SparkLLM.serve) also pays a
             host round trip every `decode_stride` steps and up to that many surplus steps per finished row.
             `--static-batch`: all rows decode max(N_i) steps (a padded static batch).  `config.retire` in the
             JSON says which was timed.  The vocoder batch is padded to max(N_i) with per-row lengths.
             `--uniform` keeps 32 x (128 -> 150) for comparison with round-1 numbers.
  --clone --batch 8 (configs[4])  each utterance first encodes a 6 s prompt wav on the GPU.
  --gpus N   one rank per GPU: `value` = the same per-rank loop on every rank (weak scaling, N x the work).
             Started without WORLD_SIZE it spawns the N ranks itself (torch.distributed.run, before any
             GPU call) and exits with their code.  Every line also carries `config4` (configs[3]): the
             256-utterance set (cfg-3-style inputs, seeds 3000+i) dealt over the ranks by
             sparkmi.dist.shard_indices, batches of <= 32 per GPU, one pass = strong scaling.

Weights are synthetic (no checkpoint exists offline), generated on rank 0 and broadcast over RCCL;
inputs are already in HBM when the timed region starts (prompt ids are ~1 KiB of host data per
utterance, passed by value at prefill).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant decode kernel, measured live with HIP
events on the launch stream; `cpu_baseline` times the CPU oracle (a port of the reference's
PyTorch-CPU arithmetic) on this box's host cores: the whole configs[1] utterance, 1 warm-up + median of 3.
"""
from __future__ import annotations

import argparse
import ctypes
import glob
import hashlib
import json
import os
import subprocess
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
    ap.add_argument("--engine", action="store_true", help="one-row steps through the persistent-layer engine (csrc/smi_eng.h; opt-in A/B: slower than the launch path on MI355X, DESIGN.md 3.7)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probes", action="store_true")
    ap.add_argument("--cpu-tokens", type=int, default=150, help="tokens of the CPU-oracle utterance (SURVEY 8d: the whole configs[1] utterance)")
    ap.add_argument("--cpu-runs", type=int, default=3, help="timed CPU-oracle runs (median reported) after one warm-up run")
    ap.add_argument("--uniform", action="store_true", help="--batch > 1 with identical lengths (--prompt-len -> --new-tokens) instead of the ragged configs[2] inputs")
    ap.add_argument("--static-batch", action="store_true", help="ragged batches decode max(N_i) steps on all rows (no retirement of finished rows)")
    ap.add_argument("--no-config4", action="store_true", help="skip the 256-utterance sharded set (BASELINE configs[3])")
    ap.add_argument("--set-size", type=int, default=256, help="utterances of the configs[3] set")
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


def cpu_baseline(llm_cfg, voc_cfg, prompt, glob, n_tokens, runs):
    """The oracle (CPU restatement pinned to the reference by tests/golden) on the host cores: SURVEY 8d's protocol --
    the configs[1] utterance (prefill + n_tokens greedy tokens + their vocoder frames), 1 warm-up + `runs` timed, median."""
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
    times, toks, samples = [], None, 0
    for r in range(runs + 1):
        log(f"cpu_baseline: run {r} of 1 warm-up + {runs} ({n_tokens} tokens)")
        t0 = time.time()
        toks = ref.generate_greedy(prompt, n_tokens)
        t_llm = time.time() - t0
        sem = torch.tensor([[t % voc_cfg.codebook_size for t in toks]])
        t0 = time.time()
        wav = voc.detokenize(sem, torch.as_tensor(glob)[None, None])
        t_voc = time.time() - t0
        samples = wav.shape[-1]
        if r > 0:
            times.append((t_llm + t_voc, t_llm, t_voc))
    total, t_llm, t_voc = sorted(times)[len(times) // 2]
    return {
        "value": samples / total, "unit": "audio samples/s", "cores": cores, "kind": "port",
        "rtf": total / (samples / 16000.0),
        "sample": f"the configs[1] utterance: {len(prompt)}-token prefill + {n_tokens} greedy tokens ({t_llm:.2f}s) + vocoder of "
                  f"those {n_tokens} frames ({t_voc:.2f}s), fp32 torch-CPU oracle, 1 warm-up + median of {runs} runs "
                  f"({', '.join(f'{t[0]:.2f}' for t in times)} s), weights build {build_s:.1f}s excluded",
        "cpu_model": _cpu_model(), "first_tokens": toks[:8],
    }


def cpu_third_party(llm_cfg, prompt, n_tokens):
    """SURVEY 8d's second CPU data point: transformers' OWN Qwen2ForCausalLM.generate(do_sample=False) -- the third-party code the
    reference runs (cli/SparkTTS.py:49,197-204), not this repository's restatement -- on the same synthetic weights and prompt,
    one warm-up + one timed run of a bounded token count.  LLM half only (the reference's vocoder cannot travel to this box).
    None when transformers is not importable here."""
    try:
        from transformers import Qwen2Config, Qwen2ForCausalLM
    except Exception as e:   # noqa: BLE001 -- absence is a recorded fact, not an error
        return {"available": False, "why": f"transformers not importable: {type(e).__name__}"}
    from sparkmi import weights as W
    syn = W.SyntheticLLM(llm_cfg)
    hc = Qwen2Config(vocab_size=llm_cfg.vocab_size, hidden_size=llm_cfg.hidden_size, intermediate_size=llm_cfg.intermediate_size,
                     num_hidden_layers=llm_cfg.num_hidden_layers, num_attention_heads=llm_cfg.num_attention_heads,
                     num_key_value_heads=llm_cfg.num_key_value_heads, rms_norm_eps=llm_cfg.rms_norm_eps, rope_theta=llm_cfg.rope_theta,
                     tie_word_embeddings=llm_cfg.tie_word_embeddings, max_position_embeddings=llm_cfg.max_position_embeddings,
                     use_sliding_window=False, attn_implementation="eager")
    with torch.device("meta"):
        m = Qwen2ForCausalLM(hc)
    m = m.to_empty(device="cpu")
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n == "lm_head.weight" and llm_cfg.tie_word_embeddings:
                continue
            p.copy_(torch.from_numpy(syn[n]))
    m.tie_weights()
    for mod in m.modules():   # non-persistent rotary buffers are not restored by to_empty
        if hasattr(mod, "inv_freq") and hasattr(mod, "compute_default_rope_parameters"):
            inv, _ = mod.compute_default_rope_parameters(mod.config)
            mod.inv_freq = inv
            mod.original_inv_freq = inv.clone()
    m.eval()
    ids = torch.tensor([prompt], dtype=torch.long)
    out = None
    times = []
    for r in range(2):
        t0 = time.time()
        with torch.no_grad():
            out = m.generate(ids, attention_mask=torch.ones_like(ids), max_new_tokens=n_tokens, do_sample=False, eos_token_id=None, pad_token_id=0)
        times.append(time.time() - t0)
    import transformers
    return {"available": True, "transformers": transformers.__version__, "tokens": n_tokens, "seconds": times[1],
            "tokens_per_s": n_tokens / times[1], "ms_per_token_incl_prefill": 1e3 * times[1] / n_tokens,
            "first_tokens": out[0, len(prompt):len(prompt) + 8].tolist(),
            "sample": f"{len(prompt)}-token prefill + {n_tokens} greedy tokens through transformers' generate(), fp32, 1 warm-up + 1 timed run"}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def build_hash() -> str:
    """sha1 over the HIP sources the product library is built from: ties a PMC profile to the build it was taken on."""
    h = hashlib.sha1()
    src = os.path.join(ROOT, "spark-tts_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(src, "smi_*.hip")) + glob.glob(os.path.join(src, "smi_*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def pmc_traffic(batch: int):
    """HBM bytes per launch per decode kernel from the newest profiles/r*_pmc_traffic*.json (tools/pmc_traffic.sh:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections applied) taken on THIS build and batch;
    (None, why) when there is none -- the number is never typed in."""
    bh = build_hash()
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), key=os.path.getmtime, reverse=True)
    seen = []
    for f in cands:
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        seen.append(f"{os.path.basename(f)}: build {d.get('build')} batch {d.get('batch')}")
        if d.get("build") == bh and int(d.get("batch", -1)) == batch:
            return d, os.path.basename(f)
    return None, f"no PMC profile for build {bh} at batch {batch} under profiles/ ({'; '.join(seen) or 'none at all'})"


def cfg3_inputs(n, seed0, vocab, n_glob, uniform=None):
    """SURVEY 8d, cfg 3 / cfg 4: utterance i has a prompt of U{96..160} ids and wants U{120..180} tokens, seeded seed0 + i
    (`uniform` = (P, N): fixed lengths)."""
    out = []
    for i in range(n):
        g = np.random.Generator(np.random.PCG64(seed0 + i))
        P = int(g.integers(96, 161)) if uniform is None else uniform[0]
        N = int(g.integers(120, 181)) if uniform is None else uniform[1]
        out.append({"prompt": g.integers(0, vocab, size=P).tolist(), "n": N, "glob": g.integers(0, 4096, size=n_glob)})
    return out


def spawn_ranks(n: int) -> int:
    """`bench.py --gpus N` started by hand: become the launcher (no GPU call has happened in this process)."""
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"--gpus {n} without WORLD_SIZE: spawning the ranks: {' '.join(cmd)}")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    a = parse()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # before the first HIP call: RCCL needs dmabuf IPC on this pool
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: refusing to report a number for another rank count")
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
        os.environ.setdefault("NCCL_DEBUG", "WARN")        # BEFORE the communicator exists (device_id= creates it eagerly): RCCL warnings reach stderr
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    if a.gpus > 1 and not one_gpu and torch.cuda.device_count() < a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but this process sees {torch.cuda.device_count()} GPU(s): refusing to time fewer devices than asked for")
    pre = SD.preflight(torch.device("cpu") if one_gpu else dev, rank, world, a.gpus)   # raises on every rank if anything is off
    if rank == 0 and world > 1:
        log(f"preflight: {pre}")
    arch = _lib.require_gfx950()
    # host threads: the ranks of one node share its cores (rank 0 packs the arenas; the CPU baseline only runs at world == 1)
    torch.set_num_threads(max(1, host_cores() // max(1, world)))
    if rank == 0:
        log(f"device {arch}, world {world}, host cores {host_cores()}: building synthetic weights")

    llm_cfg, voc_cfg = C.spark_0p5b_llm(), C.spark_0p5b_bicodec()
    B, P, N = a.batch, a.prompt_len, a.new_tokens
    ragged = B > 1 and not a.uniform and not a.clone
    ntok_glob = voc_cfg.spk_token_num
    # ---- synthetic inputs.  cfg 2: prompt ids ~ U[0,V) PCG64(1234), global ids PCG64(1235); cfg 3: ragged, seeds 2000+i
    if ragged:
        utts = cfg3_inputs(B, 2000 + rank * 1000, llm_cfg.vocab_size, ntok_glob)
    else:
        utts = []
        for i in range(B):
            s_ = rank * 1000 + i
            utts.append({"prompt": np.random.Generator(np.random.PCG64(1234 + 2 * s_)).integers(0, llm_cfg.vocab_size, size=P).tolist(),
                         "n": N, "glob": np.random.Generator(np.random.PCG64(1235 + 2 * s_)).integers(0, 4096, size=ntok_glob)})
    prompts = [u["prompt"] for u in utts]
    want = [u["n"] for u in utts]
    globs = [u["glob"] for u in utts]
    Nmax, Pmax = max(want), max(len(p) for p in prompts)
    do_cfg4 = not a.no_config4 and not a.clone and not a.no_probes
    max_pos = max(Pmax + Nmax, (160 + 180) if do_cfg4 else 0) + 32
    if a.clone:
        from sparkmi import config_tok as T
        from sparkmi.encoder import BiCodecEncoder, get_ref_clip
        wcfg, tcfg = T.xlsr53(), T.spark_0p5b_tok()
        max_pos += wcfg.frames(int(16000 * a.prompt_seconds)) + tcfg.spk_token_num
    max_frames = max(Nmax, 180 if do_cfg4 else 0) + 10
    cs_llm = A.llm_cfg_struct(llm_cfg, B, max_pos, a.kv, not a.no_graph)
    cs_voc = BC.voc_cfg_struct(voc_cfg, B, max_frames)

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

    if a.engine:   # the experimental one-row engine lives in the diagnostics build (include/sparkmi_debug.h)
        os.environ["SPARKMI_ENGINE"] = "1"
    llm = SparkLLM(llm_cfg, None, dev, max_slots=B, max_positions=max_pos, kv_dtype=a.kv,
                   use_graph=not a.no_graph, arena=llm_arena, diag=bool(a.engine))
    voc = BiCodecVocoder(voc_cfg, None, dev, max_batch=B, max_frames=max_frames, arena=voc_arena)
    enc = None
    if a.clone:
        if rank == 0:
            log("building the prompt encoder (wav2vec2-large-xlsr-53 shape, 16 layers + BiCodec tokenizer side)")
        # rank 0 packs the 1 GB encoder arena, the others receive it like the other two (one more broadcast at start-up)
        from sparkmi.encoder import enc_cfg_struct, pack_enc_arena
        max_s, max_r = int(a.prompt_seconds * tcfg.sample_rate), int(6.0 * tcfg.sample_rate) + tcfg.n_fft
        cs_enc = enc_cfg_struct(wcfg, tcfg, max_s, max_r, None)
        enc_arena = None
        if rank == 0:
            enc_arena = torch.from_numpy(pack_enc_arena(tcfg, W.fold_pos_conv_weight_norm(W.wav2vec2_state(wcfg)),
                                                        W.fold_weight_norm(W.bicodec_tok_state(tcfg, voc_cfg.vq_input_dim)), cs_enc)).to(dev)
        n_enc = int(_lib.lib().smi_enc_arena_bytes(ctypes.byref(cs_enc))) // 4
        (enc_arena,), enc_bcast_ms = SD.broadcast_tensors([enc_arena], [(n_enc, torch.float32)], dev, rank, world)
        bcast_ms += enc_bcast_ms
        enc = BiCodecEncoder(wcfg, tcfg, None, None, dev, max_seconds=a.prompt_seconds, ref_seconds=6.0, arena=enc_arena)
        tt = np.arange(int(16000 * a.prompt_seconds)) / 16000.0
        pwavs, prefs = [], []
        for i in range(B):
            f0 = 100 + 15 * i + 30 * np.sin(2 * np.pi * 0.7 * tt + i)
            ph = 2 * np.pi * np.cumsum(f0) / 16000.0
            x = (0.3 * (0.5 + 0.5 * np.sin(2 * np.pi * 2.3 * tt) ** 2) * (0.5 * np.sin(ph) + 0.3 * np.sin(2 * ph) + 0.2 * np.sin(3 * ph))
                 + 0.02 * np.random.Generator(np.random.PCG64(4000 + rank * 100 + i)).standard_normal(len(tt))).astype(np.float32)
            pwavs.append(x)
            prefs.append(get_ref_clip(x, 16000, 6.0, 320).astype(np.float32))

    glob_t = torch.from_numpy(np.stack(globs)).to(dev, torch.int32).unsqueeze(1)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    decode_ms, voc_ms, enc_ms = [], [], []

    def run_batch(eng, vocoder, pr, wants, glob_dev, timed=False):
        """prefill -> decode (ragged budgets: row i retired after its own N_i tokens, known in advance -- no eos polling;
        --static-batch / equal budgets: max(N_i) - 1 steps on all rows) -> ragged vocoder batch -> host."""
        nmax = max(wants)
        if len(pr) > 1 and min(wants) < nmax and not a.static_batch:
            # ragged budgets: rows are retired as they finish (in-flight batching) -- the step runs on the live rows
            toks = eng.generate_ragged(pr, wants, None, on_prefilled=(lambda: ev[0].record()) if timed else None)
            if timed:
                ev[1].record()
        else:
            eng.prefill(pr, None)
            if timed:
                ev[0].record()
            eng.decode(nmax - 1)
            if timed:
                ev[1].record()
            toks = eng.tokens(nmax)                              # sync + D2H: the host parses ids like the reference
        sem = torch.zeros((len(pr), nmax), dtype=torch.long)
        for b, t in enumerate(toks):
            sem[b, : wants[b]] = torch.tensor(t[: wants[b]], dtype=torch.long) % voc_cfg.codebook_size
        if timed:
            ev[2].record()
        wav = vocoder.detokenize(sem.to(dev), glob_dev, lengths=wants)
        if timed:
            ev[3].record()
        out = wav.cpu()                                      # 192 KB per 3 s utterance back to the host
        if timed:
            decode_ms.append(ev[0].elapsed_time(ev[1]))
            voc_ms.append(ev[2].elapsed_time(ev[3]))
        return out, toks

    def step(timed=False):
        nonlocal glob_t
        pr = prompts
        if enc is not None:
            # prompt encode: (global, semantic) ids of each prompt wav, appended to the text prompt the way
            # process_prompt does (cli/SparkTTS.py:83-104); ids are mapped into the synthetic vocabulary
            t0 = time.perf_counter()
            toks = enc.tokenize_many(pwavs, prefs)       # the B prompts side by side on their own HIP streams
            glob_t = torch.cat([g for g, _ in toks], 0)
            sem_h = [s_.reshape(-1).cpu().numpy() for _, s_ in toks]
            g_h = glob_t.reshape(B, -1).cpu().numpy()
            pr = [prompts[i] + (g_h[i] % llm_cfg.vocab_size).tolist() + (sem_h[i] % llm_cfg.vocab_size).tolist() for i in range(B)]
            if timed:
                enc_ms.append((time.perf_counter() - t0) * 1e3)
        return run_batch(llm, voc, pr, want, glob_t, timed)

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

    samples_per_step = sum(want) * voc_cfg.hop
    value = world * a.steps * samples_per_step / el
    audio_s = world * a.steps * samples_per_step / 16000.0
    cfg_idx = 4 if a.clone else (1 if B == 1 else 2)
    shape = (f"ragged: prompts {min(len(p) for p in prompts)}..{Pmax} ids, {min(want)}..{Nmax} tokens per utterance (SURVEY 8d cfg 3, seeds 2000+i)"
             if ragged else f"{P}-token prompt -> {N} tokens -> {N * voc_cfg.hop / 16000.0:.1f} s audio per utterance")
    res = {
        "metric": "audio samples/sec (whole node), Spark-TTS-0.5B greedy", "value": value, "unit": "audio samples/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1000.0 * el / a.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("bf16 weights + KV, fp32 activations/accumulate (LLM); " if a.kv == "bf16"
                  else "bf16 weights, fp32 KV/activations (LLM); ") +
                 ("fp32 (vocoder, exact-fp32 matrix pipe)" if voc.exact_fp32
                  else "vocoder: fp32 operands split into 2 bf16 planes, 3 products on the bf16 matrix pipe, fp32 accumulate"),
        "data": "synthetic (seeded prompts and weights; no checkpoint or dataset offline)",
        "config": {"workload": f"Spark-TTS-0.5B, batch={B} greedy, {shape} (BASELINE.json configs[{cfg_idx}])",
                   **({"voice_clone": f"each utterance encodes a {a.prompt_seconds:.1f} s prompt wav on the GPU first (BASELINE.json configs[4])"} if a.clone else {}),
                   "batch_per_gpu": B, "prompt_len": [len(p) for p in prompts] if ragged else P, "new_tokens": want if ragged else N,
                   "kv_cache": a.kv, "hipgraph": not a.no_graph,
                   "retire": ("n/a (one row)" if B == 1 else "static batch: every row decodes max(N_i) steps" if (a.static_batch or min(want) == Nmax)
                              else "in-flight, oracle budgets: row i retired after exactly N_i tokens, no eos polling"), "parallelism": f"utterance-parallel x{world} (same per-rank loop on every rank)",
                   "device": arch, "build": build_hash()},
        "rtf": el / audio_s, "x_realtime": audio_s / el,
        "utterances_per_s": world * a.steps * B / el,
        "multi_gpu_preflight": pre,
        "weights": {"build_s_rank0": t_build, "rccl_broadcast_ms": bcast_ms,
                    "llm_arena_bytes": int(llm_arena.numel()), "voc_arena_bytes": int(voc_arena.numel()) * 4},
        "stage_ms": {f"decode_{Nmax - 1}_steps": float(np.median(decode_ms)), "vocoder": float(np.median(voc_ms)),
                     **({"prompt_encode_all": float(np.median(enc_ms))} if enc_ms else {})},
        "first_tokens": toks[0][:8], "wav_std": float(wav.std()),
    }

    # ---- BASELINE configs[3]: the 256-utterance set, sharded over the ranks, batches of <= 32 per GPU (strong scaling)
    if do_cfg4:
        SET, GB = a.set_size, 32
        reqs = cfg3_inputs(SET, 3000, llm_cfg.vocab_size, ntok_glob)
        mine = SD.shard_indices([r["n"] for r in reqs], rank, world)
        if B == GB:
            llm32, voc32 = llm, voc
        else:
            llm32 = SparkLLM(llm_cfg, None, dev, max_slots=GB, max_positions=max_pos, kv_dtype=a.kv, use_graph=not a.no_graph, arena=llm_arena)
            voc32 = BiCodecVocoder(voc_cfg, None, dev, max_batch=GB, max_frames=max_frames, arena=voc_arena)

        def run_shard():
            got = 0
            for i in range(0, len(mine), GB):
                grp = [reqs[j] for j in mine[i: i + GB]]
                g_dev = torch.from_numpy(np.stack([r["glob"] for r in grp])).to(dev, torch.int32).unsqueeze(1)
                run_batch(llm32, voc32, [r["prompt"] for r in grp], [r["n"] for r in grp], g_dev)
                got += sum(r["n"] for r in grp)
            return got

        if rank == 0:
            log(f"config4: {SET} utterances, {len(mine)} on this rank, batches of {GB}")
        grp0 = [reqs[j] for j in mine[:GB]]
        run_batch(llm32, voc32, [r["prompt"] for r in grp0], [r["n"] for r in grp0],
                  torch.from_numpy(np.stack([r["glob"] for r in grp0])).to(dev, torch.int32).unsqueeze(1))     # warm-up (graph capture at this batch)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_shard()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el4 = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el4], device="cpu" if one_gpu else dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el4 = float(t.item())
        tot = sum(r["n"] for r in reqs) * voc_cfg.hop
        res["config4"] = {"workload": f"BASELINE.json configs[3]: {SET} utterances (prompts U{{96..160}} ids, U{{120..180}} tokens, seeds 3000+i) "
                                      f"dealt over {world} rank(s) by sparkmi.dist.shard_indices (longest first, serpentine), batches of <= {GB} per GPU, one pass",
                          "value": tot / el4, "unit": "audio samples/s", "scaling": "strong", "seconds": el4,
                          "utterances_per_s": SET / el4, "x_realtime": tot / 16000.0 / el4, "n_gpus": world,
                          "utterances_per_rank": len(mine), "rccl_broadcast_ms": bcast_ms}
        if llm32 is not llm:
            llm32.close()

    if rank == 0:
        log(f"timed: {el:.3f}s for {a.steps} steps -> {value:.0f} samples/s; probing kernels")
        # ---- roofline: decode step is HBM-bound; algorithmic bytes = weights once + KV read + KV write.  Every row steps
        # through all Nmax - 1 decode steps (rows past their own length are the padding of a static batch).
        wb = llm.step_weight_bytes()
        kvb = llm.kv_bytes_per_token()
        # (a ragged batch retires rows at their own length N_i: row i takes part in N_i - 1 of the Nmax - 1 steps; with
        # --static-batch every row steps through all of them, rows past their length being padding)
        retiring = B > 1 and min(want) < Nmax and not a.static_batch
        eff = want if retiring else [Nmax] * B
        ctx_sum = sum(len(p) + j for p, w in zip(prompts, eff) for j in range(1, w))   # decode step j of a row reads its positions 0..P+j-1
        row_steps = sum(w - 1 for w in eff)
        step_bytes = wb + kvb * (ctx_sum + row_steps) / (Nmax - 1) + kvb * row_steps / (Nmax - 1)   # KV read (cached + own) + KV write
        dec_step_ms = res["stage_ms"][f"decode_{Nmax - 1}_steps"] / (Nmax - 1)
        res["roofline_step"] = {
            "bound": "hbm", "achieved": step_bytes / (dec_step_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": step_bytes / (dec_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
            "bytes_per_step": step_bytes, "ms_per_step": dec_step_ms,
            "note": "whole decode step of this workload (one hipGraph replay), events inside the timed region"}
        # HBM traffic of a whole step from the PMC profile of THIS build at this batch, when there is one: layers x the layer
        # kernels' read + write bytes per launch + lm_head's (the profile is taken at full, uniform rows: an upper bound for a
        # ragged batch whose rows retire)
        tr_step, tr_src = pmc_traffic(B)
        if tr_step is not None:
            tk = tr_step.get("kernels", {})
            need = [k for k in ("qkv", "attn", "o_proj", "gate_up", "down") if k in tk]
            if {"qkv", "attn", "gate_up", "down", "lm_head"} <= set(tk):
                res["roofline_step"]["traffic"] = (llm_cfg.num_hidden_layers * sum(tk[k]["hbm_bytes_per_launch"] for k in need)
                                                   + tk["lm_head"]["hbm_bytes_per_launch"])
                res["roofline_step"]["traffic_source"] = tr_src
        if not a.no_probes:
            # per-kernel probes, stamps and the sampling-step timing are diagnostics (include/sparkmi_debug.h): they run on a
            # second engine on libsparkmi_diag.so -- the same sources and kernels, same arena -- never on the product library
            # that the timed region above ran on
            pl = llm if llm._lib.is_diag else SparkLLM(llm_cfg, None, dev, max_slots=B, max_positions=max_pos, kv_dtype=a.kv,
                                                      use_graph=not a.no_graph, arena=llm_arena, diag=True)
            pl.prefill(prompts, None)
            pl.decode(Nmax // 2)
            kb = pl.weight_bytes()
            ctx_now = sum(len(p) + Nmax // 2 for p in prompts)
            per = {"qkv": kb["qkv"] + kvb // llm_cfg.num_hidden_layers * B,
                   "attn": kvb // llm_cfg.num_hidden_layers * ctx_now,
                   "o_proj": kb["o_proj"], "gate_up": kb["gate_up"], "down": kb["down"], "lm_head": kb["lm_head"]}
            count = {k: llm_cfg.num_hidden_layers for k in per}
            count["lm_head"] = 1
            eng = pl.engine_info() if B == 1 else {"enabled": False}
            ks = []
            if eng["enabled"]:
                # one live row: all layers of a step are ONE persistent launch (csrc/smi_eng.h); the launch-path layer kernels
                # below are what batches run -- probed for comparison, not part of this workload's step
                pl.prefill(prompts, None)
                pl.decode(Nmax // 2)
                ms = pl.time_kernel("layers", iters=48)
                eb = llm_cfg.num_hidden_layers * (kb["qkv"] + kb["o_proj"] + kb["gate_up"] + kb["down"]) + kvb * (ctx_now + B)
                ks.append({"kernel": "engine", "what": f"all {llm_cfg.num_hidden_layers} layers of the step, one persistent launch",
                           "launches_per_step": 1, "in_step": True, "avg_us": ms * 1e3, "bytes": eb,
                           "GBps": eb / (ms * 1e-3) / 1e9, "us_per_step": ms * 1e3,
                           "us_per_layer": ms * 1e3 / llm_cfg.num_hidden_layers})
                pl.prefill(prompts, None)
                pl.decode(Nmax // 2)
            for name in ("qkv", "attn", "o_proj", "gate_up", "down", "lm_head"):
                # layer kernels are timed where they run: (96 layers captured in a hipGraph) - (the same graph without
                # the kernel), so each finds the L2 state its producers leave (idle CUs prefetch part of the next
                # kernels' weights) and no host launch rate enters; lm_head (50 us) is looped on its own
                ms = pl.time_kernel(name, iters=96, in_sequence=True) if name != "lm_head" else pl.time_kernel(name, iters=24)
                in_step = name == "lm_head" or not eng["enabled"]
                fused = name == "o_proj" and B == 1 and ms * 1e3 < 0.5    # one row: done inside the attention kernel (k_attn<.., FUSE>)
                ent = {"kernel": name, "launches_per_step": count[name], "in_step": in_step, "avg_us": ms * 1e3,
                       "bytes": per[name], "GBps": per[name] / (ms * 1e-3) / 1e9,
                       "us_per_step": ms * 1e3 * count[name]}
                if fused:   # the leave-one-out difference is noise around zero: no rate
                    ent.update({"fused": True, "avg_us": 0.0, "GBps": None, "us_per_step": 0.0})
                ks.append(ent)
            dom = max((k for k in ks if k["in_step"]), key=lambda k: k["us_per_step"])
            prof, src = pmc_traffic(B)
            traffic = None
            if prof is not None and dom["kernel"] in prof.get("kernels", {}):
                traffic = prof["kernels"][dom["kernel"]]["hbm_bytes_per_launch"]
            res["roofline"] = {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["GBps"], "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": dom["GBps"] / HBM_PEAK_GBS, "traffic": traffic,
                               "traffic_source": src, "bytes_per_launch": dom["bytes"], "avg_us": dom["avg_us"]}
            if eng["enabled"]:
                res["roofline"]["note"] = ("dominant kernel of the one-row step = the persistent-layer engine: algorithmic bytes = "
                                           "24 layers of weights + KV read / write of this context, over its launch duration")
            res["engine"] = eng
            res["kernels"] = ks
            # the reference's default mode (temperature 0.8 / top-k 50 / top-p 0.95): decode step with the sampler in the graph
            pl.set_sampling(True, 0.8, 50, 0.95, 1234)
            pl.prefill(prompts, None)
            pl.decode(8)
            res["sampling_step_us"] = pl.time_kernel("step", iters=64) * 1e3
            pl.set_sampling(False)
            pl.prefill(prompts, None)
            pl.decode(8)
            res["greedy_step_us"] = pl.time_kernel("step", iters=64) * 1e3
            if pl is not llm:
                pl.close()
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
            run_batch(llm, voc, prompts, want, glob_t)
            vl = []
            for i in range(voc.launches()):
                nm, ms, fl = voc.time_launch(i, iters=5)
                vl.append((nm, ms, fl))
            tot_ms = sum(m for _, m, _ in vl)
            tot_fl = sum(f for _, _, f in vl)
            res["vocoder_mfma"] = {"launches": len(vl), "sum_ms": tot_ms, "gflop": tot_fl / 1e9,
                                   "achieved_TFLOPs": tot_fl / (tot_ms * 1e-3) / 1e12,
                                   # fp32-equivalent FLOPs on the bf16 pipe with three bf16 products per fp32 product
                                   # (w_hi x_hi + w_hi x_mid + w_mid x_hi): peak = 2.5 PFLOP/s dense bf16 / 3
                                   "peak_TFLOPs": 2500.0 / 3, "peak_note": "bf16 MFMA 2.5 PF dense / 3 split products per fp32 product",
                                   "top": [{"name": n, "ms": m, "TFLOPs": f / (m * 1e-3) / 1e12 if m > 0 else 0}
                                           for n, m, f in sorted(vl, key=lambda x: -x[1])[:6]]}
        else:
            res["roofline"] = dict(res["roofline_step"])
        if not a.no_cpu_baseline and world == 1:
            log("gpu side done; timing the CPU oracle")
            cprompt = np.random.Generator(np.random.PCG64(1234)).integers(0, llm_cfg.vocab_size, size=128).tolist()
            cglob = np.random.Generator(np.random.PCG64(1235)).integers(0, 4096, size=ntok_glob)
            res["cpu_baseline"] = cpu_baseline(llm_cfg, voc_cfg, cprompt, cglob, a.cpu_tokens, a.cpu_runs)
            res["gpu_over_cpu_rtf"] = res["cpu_baseline"]["rtf"] / res["rtf"]
            log("timing transformers' own generate() on the host (third-party data point)")
            tp = cpu_third_party(llm_cfg, cprompt, min(32, a.cpu_tokens))
            if tp.get("available"):   # the restated path and the third-party code must walk the same tokens
                tp["same_first_tokens_as_oracle"] = tp["first_tokens"] == res["cpu_baseline"]["first_tokens"][:len(tp["first_tokens"])]
            res["cpu_baseline"]["third_party"] = tp
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
