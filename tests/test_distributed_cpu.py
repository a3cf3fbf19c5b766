"""World-size-2 gloo tests (CPU) of the multi-GPU plumbing: partitioning, the max-over-ranks step
time, and the optional input scatter / output gather around the (collective-free) decode."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from lzma_rs_amd import distributed as D  # noqa: E402


def test_partition_planner_through_the_c_symbol():
    """milzma_partition (the planner behind milzma_multi_* and bench.py --scatter), no GPU needed: every item exactly once,
    the heaviest part within one item of the lightest, grouped items together, deterministic."""
    import lzma_rs_amd as M
    sizes = [(i * 7919) % 1000 + 1 for i in range(500)]
    for world in (1, 2, 3, 4, 8):
        parts = D.shard_by_bytes(sizes, world)
        assert sorted(i for p in parts for i in p) == list(range(500))
        loads = [sum(sizes[i] for i in p) for p in parts]
        assert max(loads) - min(loads) <= max(sizes)
        assert D.shard_by_bytes(sizes, world) == parts
    # the blocks of one .xz file / the units of one LZMA2 group are never split
    groups = [1 + i // 4 if i < 400 else 0 for i in range(500)]
    part = M.partition(sizes, 8, groups)
    for g in range(1, 101):
        assert len({part[i] for i in range(500) if groups[i] == g}) == 1
    loads = [sum(s for s, p in zip(sizes, part) if p == k) for k in range(8)]
    assert max(loads) - min(loads) <= 4 * max(sizes)
    assert M.partition([], 3) == [] and M.partition([7], 1) == [0]
    with pytest.raises(M.InfraError):
        M.partition([1, 2], 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w = D.init(backend="gloo")
    assert (r, w) == (rank, world)
    dev = torch.device("cpu")
    # step time agreed by all ranks = the slowest rank's
    t = D.max_over_ranks(1.0 + rank, dev)
    total = D.sum_over_ranks(10 * (rank + 1), dev)
    # rank 0 scatters one compressed chunk per rank; every rank "decodes" (here: transforms) its own
    # chunk without talking to anyone; rank 0 gathers the variable-length outputs
    chunks = None
    if rank == 0:
        chunks = [torch.arange(100 + 50 * k, dtype=torch.uint8) for k in range(world)]
    mine = D.scatter_inputs(chunks if rank == 0 else [None] * world, dev)
    out = (mine.to(torch.int16) * 2 % 251).to(torch.uint8).repeat(rank + 1)
    D.barrier_sync(None)
    gathered = D.gather_outputs(out, dev)
    ok = True
    if rank == 0:
        for k in range(world):
            want = (torch.arange(100 + 50 * k, dtype=torch.int16) % 256 * 2 % 251).to(torch.uint8).repeat(k + 1)
            ok = ok and torch.equal(gathered[k], want)
    q.put((rank, t, total, mine.numel(), ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_gloo(world):
    """world 2 and 3 (unequal chunk / output sizes per rank; rank 0 posts every receive of the gather before it waits for any)"""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, t, total, n, ok in res:
        assert t == float(world) and total == 10.0 * world * (world + 1) / 2 and n == 100 + 50 * rank and ok


def test_bench_rank_logic_two_ranks_gloo():
    """bench.py's multi-rank path up to (not including) the first decode, world size 2 over gloo on CPU: generation
    (rank 0's pool in --scatter mode, cut per rank by the library's partition planner), process-group rendezvous, input and
    descriptor scatter, barrier and the max-over-ranks / sum-over-ranks reductions."""
    import json
    import subprocess
    env = dict(os.environ, MILZMA_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--scatter", "--streams", "8",
                        "--size", "65536", "--distinct", "4"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["dry_run"] and line["n_gpus"] == 2 and line["units_all_ranks"] == 16 and line["max_time"] == 0.002
    assert line["distinct_all_ranks"] == 8       # rank 0's pool of 2 x 4 different streams, every one on exactly one rank


def test_bench_refuses_more_gpus_than_present():
    import subprocess
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("MILZMA_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 2)], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)


def test_bench_traffic_only_from_records_of_this_kernel_source(tmp_path, monkeypatch):
    """bench.py's roofline.traffic: the recorded PMC figure only if the record was taken on exactly the kernel source being run;
    anything else is null -- a record cannot vouch for another source (round 3's `also_valid_for` is gone)."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    rec = {"kernel_source_sha256": "aaaa", "derived": {"hbm_bytes_per_launch": 123.0}, "also_valid_for": {"bbbb": "no memory instruction changed"}}
    (prof / (bench.PROFILE_ROUND + "_pmc_lzma64k.json")).write_text(json.dumps(rec))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.pmc_traffic("lzma64k", "aaaa") == (123.0, None)
    assert bench.pmc_traffic("lzma64k", "bbbb") == (None, None)
    assert bench.pmc_traffic("lzma64k", "cccc") == (None, None)
    assert bench.pmc_traffic("dict8m", "aaaa") == (None, None)
