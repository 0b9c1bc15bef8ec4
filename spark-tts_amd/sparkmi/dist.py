"""Multi-GPU: one process per MI355X, utterances sharded across ranks, no data-path collective.

The only collective is the start-up broadcast of the two weight arenas from rank 0 (RCCL over
xGMI when the backend is "nccl"; the same code runs over gloo on CPU tensors in the tests).
Utterances are independent (the reference has no cross-utterance state: cli/SparkTTS.py:157-236),
so each rank synthesises its own shard and rank 0 gathers the waveforms over host memory.
"""
from __future__ import annotations

import ctypes as C
import time
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def arena_sizes(cs_llm, cs_voc) -> Tuple[int, int]:
    """(LLM arena bytes, vocoder arena floats) from the library's own layout functions."""
    from . import _lib
    l = _lib.lib()
    return int(l.smi_llm_arena_bytes(C.byref(cs_llm))), int(l.smi_voc_arena_bytes(C.byref(cs_voc))) // 4


def physical_device_id(device: torch.device) -> Tuple[str, str]:
    """(host name, physical id of the card): uuid where the runtime reports one, else the PCI bus id, else the visible
    index qualified by HIP_VISIBLE_DEVICES -- equal on two ranks only when they really share a card."""
    import os
    import socket
    host = socket.gethostname()
    if device.type != "cuda":
        return host, f"{device.type}:{os.getpid()}"
    idx = device.index if device.index is not None else torch.cuda.current_device()
    p = torch.cuda.get_device_properties(idx)
    for attr in ("uuid", "pci_bus_id"):
        v = getattr(p, attr, None)
        if v is not None and str(v) not in ("", "0", "00000000-0000-0000-0000-000000000000"):
            extra = f"{getattr(p, 'pci_domain_id', 0)}:{getattr(p, 'pci_device_id', 0)}" if attr == "pci_bus_id" else ""
            return host, f"{attr}={v}{(':' + extra) if extra else ''}"
    return host, f"visible={os.environ.get('HIP_VISIBLE_DEVICES', os.environ.get('ROCR_VISIBLE_DEVICES', '*'))}#{idx}"


def preflight(device: torch.device, rank: int, world: int, expect_world: int) -> dict:
    """Fail loudly and early on a multi-GPU launch that is not what was asked for -- nobody can rehearse the 8-GPU run on a
    one-GPU lease, so everything that can be checked before the 1.4 GB broadcast is: the rank count, one distinct device per
    rank, a 1 MB broadcast that must arrive intact (timed), and the collective library's version for the record.
    Every rank calls it; raises RuntimeError on every rank when any check fails."""
    info = {"world": world, "backend": dist.get_backend() if world > 1 else None, "rccl_version": None,
            "small_broadcast_ms": None, "devices": None}
    if world != expect_world:
        raise RuntimeError(f"preflight: {world} ranks are running, {expect_world} were asked for")
    if world == 1:
        return info
    if dist.get_world_size() != expect_world:
        raise RuntimeError(f"preflight: process group has {dist.get_world_size()} ranks, {expect_world} were asked for")
    if device.type == "cuda":
        if torch.cuda.device_count() < 1:
            raise RuntimeError("preflight: no GPU visible to this rank")
        try:
            info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:   # noqa: BLE001 -- the version is for the record only
            info["rccl_version"] = "unknown"
    # one distinct PHYSICAL device per rank (a launcher that maps two ranks to one card would halve the node silently).  The
    # identity is (host name, the card's uuid / PCI bus id): device indices alone say nothing across nodes, nor under a
    # launcher that gives every rank its own HIP_VISIBLE_DEVICES (each then sees its card as cuda:0)
    mine = physical_device_id(device)
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    info["devices"] = [f"{h}/{d}" for h, d in everyone]
    if device.type == "cuda" and dist.get_backend() == "nccl" and len(set(everyone)) != world:
        raise RuntimeError(f"preflight: ranks share devices: {everyone}")
    # a small broadcast before the big one: pattern checked on every rank, time recorded
    n = 1 << 20
    stage = device if (device.type != "cuda" or dist.get_backend() == "nccl") else torch.device("cpu")
    buf = (torch.arange(n, dtype=torch.int64, device=stage) % 251).to(torch.uint8) if rank == 0 else torch.zeros(n, dtype=torch.uint8, device=stage)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    dist.barrier()
    t0 = time.perf_counter()
    dist.broadcast(buf, src=0)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    info["small_broadcast_ms"] = (time.perf_counter() - t0) * 1e3
    want = (torch.arange(n, dtype=torch.int64, device=stage) % 251).to(torch.uint8)
    ok = torch.tensor([1 if torch.equal(buf, want) else 0], dtype=torch.int32, device=stage)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) != 1:
        raise RuntimeError("preflight: the 1 MB test broadcast did not arrive intact on every rank")
    return info


def broadcast_tensors(tensors: Sequence[Optional[torch.Tensor]], shapes: Sequence[Tuple[int, torch.dtype]],
                      device: torch.device, rank: int, world: int):
    """Rank 0 passes its flat arenas, the others pass None (they allocate [n] of the given dtype); returns (list, ms).
    One broadcast per arena -- the only collective of the whole path."""
    out = list(tensors)
    if rank != 0:
        out = [torch.empty(n, dtype=dt, device=device) for n, dt in shapes]
    if world == 1:
        return out, 0.0
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    dist.barrier()
    t0 = time.perf_counter()
    if dist.get_backend() == "gloo" and device.type == "cuda":
        # gloo rehearsal path: stage through host memory
        for a in out:
            h = a.cpu()
            dist.broadcast(h, src=0)
            if rank != 0:
                a.copy_(h)
    else:
        for a in out:
            dist.broadcast(a, src=0)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    dist.barrier()
    return out, (time.perf_counter() - t0) * 1e3


def broadcast_arenas(llm_arena: Optional[torch.Tensor], voc_arena: Optional[torch.Tensor], sizes: Tuple[int, int],
                     device: torch.device, rank: int, world: int):
    """Rank 0 passes its arenas, the others pass None; returns (llm_arena, voc_arena, milliseconds)."""
    (llm_arena, voc_arena), ms = broadcast_tensors([llm_arena, voc_arena], [(sizes[0], torch.uint8), (sizes[1], torch.float32)],
                                                   device, rank, world)
    return llm_arena, voc_arena, ms


def shard_indices(lengths: Sequence[int], rank: int, world: int) -> List[int]:
    """Utterance indices for `rank`: longest first, dealt round-robin (serpentine) so every rank
    gets a similar amount of audio."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    mine = []
    for pos, idx in enumerate(order):
        rnd, k = divmod(pos, world)
        owner = k if rnd % 2 == 0 else world - 1 - k
        if owner == rank:
            mine.append(idx)
    return mine


def synthesize_sharded(requests: Sequence[dict], est_lengths: Sequence[int], synth: Callable[[List[dict]], List[np.ndarray]],
                       rank: int, world: int, batch: int = 1) -> Optional[List[np.ndarray]]:
    """Run `synth` on this rank's shard (in groups of `batch`); rank 0 returns every waveform in
    request order, other ranks return None."""
    mine = shard_indices(est_lengths, rank, world)
    out = []
    for i in range(0, len(mine), batch):
        grp = mine[i: i + batch]
        wavs = synth([requests[j] for j in grp])
        out.extend(zip(grp, wavs))
    if world == 1:
        res = [None] * len(requests)
        for j, w in out:
            res[j] = w
        return res
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(out, gathered, dst=0)
    if rank != 0:
        return None
    res = [None] * len(requests)
    for part in gathered:
        for j, w in part:
            res[j] = w
    return res
