"""The N>1 path on CPU: two gloo ranks run the same start-up broadcast and utterance sharding
code bench.py / the pipeline use with RCCL on the GPUs (sparkmi/dist.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sparkmi import dist as SD


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cpu")
        pre = SD.preflight(dev, rank, world, world)          # rank count, 1 MB test broadcast checked on every rank
        assert pre["world"] == world and pre["small_broadcast_ms"] >= 0 and len(pre["devices"]) == world
        try:
            SD.preflight(dev, rank, world, world + 1)        # asked for another rank count: every rank refuses
            raise AssertionError("preflight accepted a wrong rank count")
        except RuntimeError:
            pass
        sizes = (4096 + 17, 999)
        if rank == 0:
            a = torch.arange(sizes[0], dtype=torch.int64).to(torch.uint8)
            b = torch.linspace(-1, 1, sizes[1])
        else:
            a = b = None
        a, b, ms = SD.broadcast_arenas(a, b, sizes, dev, rank, world)
        ok = bool((a == torch.arange(sizes[0], dtype=torch.int64).to(torch.uint8)).all()) and \
            bool(torch.equal(b, torch.linspace(-1, 1, sizes[1]))) and ms >= 0
        reqs = [dict(id=i) for i in range(11)]
        lens = [5, 9, 1, 7, 3, 8, 2, 6, 4, 10, 11]
        calls = []

        def synth(group):
            calls.append([r["id"] for r in group])
            return [np.full(lens[r["id"]], r["id"], np.float32) for r in group]

        res = SD.synthesize_sharded(reqs, lens, synth, rank, world, batch=2)
        if rank == 0:
            ok = ok and all(res[i].shape == (lens[i],) and (res[i] == i).all() for i in range(11))
        else:
            ok = ok and res is None
        mine = SD.shard_indices(lens, rank, world)
        ok = ok and sorted(sum(calls, [])) == sorted(mine) and all(len(c) <= 2 for c in calls)
        q.put((rank, ok, sum(lens[i] for i in mine)))
    finally:
        dist.destroy_process_group()


def test_two_rank_broadcast_and_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in got)
    loads = [l for _, _, l in got]
    assert abs(loads[0] - loads[1]) <= 11      # serpentine dealing balances the audio per rank


def test_shards_partition_the_requests():
    lens = list(np.random.default_rng(0).integers(1, 200, size=37))
    for world in (1, 2, 4, 8):
        parts = [SD.shard_indices(lens, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(37))
        tot = [sum(lens[i] for i in p) for p in parts]
        assert max(tot) - min(tot) <= max(lens)
