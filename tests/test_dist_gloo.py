"""N > 1 path on CPU: two gloo ranks shard images image-per-rank (no data-path collective), agree on disjoint seeds,
and aggregate throughput with a barrier + MAX-reduce exactly like bench.py does over RCCL."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from jpeglibrary_amd import sharding
    from jpeglibrary_amd.decoder import JpegDecoder
    from tools import jpegsynth

    dist.init_process_group("gloo", rank=rank, world_size=world)
    # each rank builds and parses its own shard: distinct seeds, host logic only (no GPU here)
    seeds = [sharding.rank_seed_base(rank) + i for i in range(3)]
    pixels = 0
    for s in seeds:
        data = jpegsynth.encode(64, 48, "420", 75, 4, seed=s)
        d = JpegDecoder(host_only=True)
        d.SetInput(data)
        d.Identify()
        pixels += d.Width * d.Height
    dist.barrier()
    # bench.py's host-side group: per-rank generation times gathered as objects, and the parking barrier the ranks wait in
    # while rank 0 measures the CPU baseline
    host_group = dist.new_group(backend="gloo")
    gen = [None] * world
    dist.all_gather_object(gen, round(0.1 * (rank + 1), 1), group=host_group)
    assert gen == [round(0.1 * (r + 1), 1) for r in range(world)]
    dist.barrier(group=host_group)
    elapsed = 0.5 + rank  # pretend rank 1 is slower
    emax = sharding.max_over_ranks(dist, elapsed)
    value = sharding.aggregate_throughput(pixels, world, steps=2, elapsed_max=emax)
    q.put((rank, seeds, sharding.shard_indices(10, rank, world), emax, value))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_aggregation():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, seeds0, idx0, emax0, v0), (r1, seeds1, idx1, emax1, v1) = results
    assert set(seeds0).isdisjoint(seeds1)
    assert sorted(idx0 + idx1) == list(range(10)) and set(idx0).isdisjoint(idx1)
    assert emax0 == emax1 == 1.5
    assert v0 == v1 == pytest.approx(3 * 64 * 48 * 2 * 2 / 1.5 / 1e6)


def test_bench_launches_its_own_ranks_when_started_without_torchrun():
    """`python bench.py --gpus 2` (the form the driver uses for --gpus 1) used to measure ONE GPU and print n_gpus: 1 when no
    torchrun had set WORLD_SIZE.  It now starts `python -m torch.distributed.run --nproc-per-node 2 bench.py ...` as a child before
    touching any device and relays rank 0's line.  --dry-run keeps the launch's plumbing (rendezvous, barrier, MAX-reduce of the
    slowest rank's time, one line from rank 0) and leaves the device work out, which is all a box without GPUs can run."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--dry-run"], capture_output=True,
                       text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["value"] is None
    assert out["slowest_rank_s"] == 0.75  # rank 1 pretends to be the slower one: the MAX over the ranks
    assert "self-spawned" in out["launch"]
