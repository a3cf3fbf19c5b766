#!/usr/bin/env python3
"""bench.py -- batched LZMA decode throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): 4096 independent 1 MiB .lzma streams per GPU, lc3/lp0/pb2,
64 KiB dictionary, "text" class plaintext (seed 0xC0FFEE ^ i), compressed with liblzma preset 6
on the host before the timed region.  A step = one call of milzma_decode_units over the whole
batch with compressed input and output slices resident in HBM (descriptor upload and result
download included).  N > 1: every rank decodes its own 4096 streams (weak scaling), no collective
on the data path; time = max over ranks between barriers.

Prints ONE JSON line (rank 0).  `roofline` is the decode kernel vs the HBM roofline using
algorithmic bytes (compressed read once + output written once); `cpu_baseline` is the CPU oracle
(a C port of the reference's decode path, oracle/) on the host cores over a bounded sample.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import lzma_rs_amd as M  # noqa: E402
from lzma_rs_amd import distributed as D  # noqa: E402
from lzma_rs_amd import workloads as W  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured copy


def effective_cores():
    """CPUs this process can actually use: affinity mask capped by the cgroup CPU quota (the GPU
    boxes expose 256 hardware threads but a 16-CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _compress_range(job):
    """Worker: compress streams [lo, hi) and park them in one /dev/shm file (returning bulk data
    through the pool's pipes would serialise on the parent)."""
    kind, size, dict_size, lo, hi, path = job
    lens = []
    with open(path, "wb") as f:
        for i in range(lo, hi):
            comp = W._one_stream_compressed((kind, size, i, dict_size, True))
            f.write(comp)
            lens.append(len(comp))
    return lens


def compress_streams(n_streams, size, kind, dict_size, first_index, processes):
    """n_streams complete .lzma streams (seed 0xC0FFEE ^ index) compressed on `processes` cores."""
    import multiprocessing
    import tempfile
    procs = max(1, min(processes, n_streams))
    tmpdir = tempfile.mkdtemp(prefix="milzma_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    per = (n_streams + procs * 4 - 1) // (procs * 4)  # ~4 jobs per worker for balance
    jobs = []
    for k, lo in enumerate(range(0, n_streams, per)):
        hi = min(n_streams, lo + per)
        jobs.append((kind, size, dict_size, first_index + lo, first_index + hi, os.path.join(tmpdir, "%d.bin" % k)))
    if procs > 1:
        with multiprocessing.get_context("fork").Pool(procs) as pool:
            all_lens = pool.map(_compress_range, jobs, chunksize=1)
    else:
        all_lens = [_compress_range(j) for j in jobs]
    comps = []
    for job, lens in zip(jobs, all_lens):
        with open(job[5], "rb") as f:
            data = f.read()
        os.unlink(job[5])
        o = 0
        for ln in lens:
            comps.append(data[o:o + ln])
            o += ln
    os.rmdir(tmpdir)
    return comps


def build_batch(n_streams, size, kind, dict_size, first_index, processes):
    """Returns (units ctypes array, host input bytes, total compressed payload bytes, seconds)."""
    t0 = time.time()
    res = compress_streams(n_streams, size, kind, dict_size, first_index, processes)
    units = (M.Unit * n_streams)()
    blobs, in_off = [], 0
    comp_total = 0
    for k, comp in enumerate(res):
        u, hl = M.lzma_read_header(comp)
        payload = comp[hl:]
        u.in_off, u.in_len = in_off, len(payload)
        u.out_off, u.out_cap = k * size, size
        units[k] = u
        pad = (-len(payload)) % 256
        blobs.append(payload)
        if pad:
            blobs.append(bytes(pad))
        in_off += len(payload) + pad
        comp_total += len(payload)
    return units, b"".join(blobs), comp_total, time.time() - t0


def cpu_baseline(sample_streams, size, kind, dict_size, threads):
    """Times the CPU oracle (port of the reference decode path) on `threads` host threads."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py as orc
    comps = compress_streams(sample_streams, size, kind, dict_size, 1 << 20, threads)
    blob = b"".join(comps)
    offs, lens, o = [], [], 0
    for c in comps:
        offs.append(o)
        lens.append(len(c))
        o += len(c)
    n = len(comps)
    a_off = (ctypes.c_uint64 * n)(*offs)
    a_len = (ctypes.c_uint64 * n)(*lens)
    chk = ctypes.c_uint32()
    buf = ctypes.create_string_buffer(blob, len(blob))
    lib = orc.lib()
    lib.orc_bench_lzma_batch(buf, a_off, a_len, min(n, threads), threads, ctypes.byref(chk))  # warm
    best = None
    for _ in range(2):
        t0 = time.time()
        total = lib.orc_bench_lzma_batch(buf, a_off, a_len, n, threads, ctypes.byref(chk))
        dt = time.time() - t0
        best = dt if best is None else min(best, dt)
    dt = best
    assert total == n * size, "oracle failed on the CPU baseline sample"
    # an independent (and faster) CPU decoder on the same sample and the same threads, so that the comparison
    # with the port of the reference is not flattering by construction: liblzma through Python's lzma module
    # (which releases the GIL while decoding)
    import lzma
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(threads) as ex:
        t0 = time.time()
        # (liblzma refuses a known-size header on a stream that also carries the end marker, SURVEY A.8: give it
        #  the header the encoder wrote, size field all ones)
        got = sum(ex.map(lambda c: len(lzma.decompress(c[:5] + b"\xff" * 8 + c[13:], format=lzma.FORMAT_ALONE)), comps))
        dt_xz = time.time() - t0
    assert got == n * size
    return {"value": round(total / dt / 1e9, 4), "unit": "GB/s decompressed", "cores": threads, "kind": "port",
            "sample": "%d x %d B %s streams, dict %d, oracle/lzma_oracle.c (C restatement of the reference; "
                      "no Rust toolchain to build the crate), one stream per thread, %.2f s wall"
                      % (n, size, kind, dict_size, dt),
            "liblzma": {"value": round(got / dt_xz / 1e9, 4), "unit": "GB/s decompressed", "cores": threads,
                        "note": "liblzma via Python lzma.decompress on the same sample (not the reference; an "
                                "independent, faster CPU decoder)"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--streams", type=int, default=4096, help="streams per GPU")
    ap.add_argument("--size", type=int, default=1 << 20, help="plaintext bytes per stream")
    ap.add_argument("--dict", type=int, default=1 << 16, help="LZMA dictionary size")
    ap.add_argument("--kind", default="text", choices=["text", "random", "repeat", "zeros"])
    ap.add_argument("--distinct", type=int, default=512,
                    help="distinct streams compressed per GPU (0 = all; fewer are tiled over the slots, each "
                         "slot still reads its own copy of the input and writes its own output slice)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="streams in the CPU baseline sample (0 = auto)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--pcie", action="store_true",
                    help="also time H2D(compressed) + decode + D2H(output) through pinned host buffers and report it as "
                         "pcie_inclusive (never as value)")
    args = ap.parse_args()

    rank, local_rank, world = D.env_world()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    cores = effective_cores()
    procs = max(1, cores // world)
    n = args.streams
    distinct = args.distinct or n
    distinct = min(distinct, n)
    # host-side generation first (forks worker processes): before any HIP/RCCL state exists
    units_d, blob_d, comp_d, gen_s = build_batch(distinct, args.size, args.kind, args.dict,
                                                 first_index=rank * n, processes=procs)
    cpu_line = None
    if world == 1 and not args.no_cpu_baseline:
        sample = args.cpu_sample or min(2048, cores * 64)  # ~20 s of CPU work
        cpu_line = cpu_baseline(sample, args.size, args.kind, args.dict, cores)

    D.init()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible (no CPU fallback exists)")
    dev_index = local_rank % torch.cuda.device_count()  # (== local_rank on a node with one GPU per rank)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    ctx = M.Context(dev_index)
    # tile the distinct streams over the n slots (each slot still reads its own copy from HBM)
    units = (M.Unit * n)()
    reps = (n + distinct - 1) // distinct
    d_in_one = torch.frombuffer(bytearray(blob_d), dtype=torch.uint8)
    d_in = d_in_one.repeat(reps).to(dev) if reps > 1 else d_in_one.to(dev)
    stride = len(blob_d)
    comp_total = 0
    for k in range(n):
        src = units_d[k % distinct]
        u = M.Unit()
        ctypes.memmove(ctypes.byref(u), ctypes.byref(src), ctypes.sizeof(M.Unit))
        u.in_off = src.in_off + (k // distinct) * stride
        u.out_off = k * args.size
        units[k] = u
        comp_total += src.in_len
    d_out = torch.empty(n * args.size, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        return ctx.decode_units(units, d_in.data_ptr(), d_out.data_ptr(), stream)

    for _ in range(args.warmup):
        step()
    D.barrier_sync(dev)
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res, ms, launches = step()
        kernel_ms.append(ms)
    torch.cuda.synchronize(dev)
    D.barrier_sync(dev)
    elapsed = D.max_over_ranks(time.perf_counter() - t0, dev)

    # correctness of what was timed: every unit OK and full length; sampled streams bit-exact
    bad = sum(1 for r in res if r.status != M.ST_OK or r.out_len != args.size)
    verified = 0
    if not args.no_verify:
        check = sorted(set([0, 1, n // 2, n - 1] + list(range(0, n, max(1, n // 16)))))
        for k in check:
            plain = W.make_plain(args.kind, args.size, W.SEED0 ^ (rank * n + (k % distinct)))
            got = d_out[k * args.size:(k + 1) * args.size].cpu().numpy().tobytes()
            if got != plain:
                bad += 1
            verified += 1
    bad_total = int(D.sum_over_ranks(bad, dev))

    pcie = None
    if args.pcie:
        h_in = torch.empty(d_in.numel(), dtype=torch.uint8, pin_memory=True)
        h_in.copy_(d_in)
        h_out = torch.empty(d_out.numel(), dtype=torch.uint8, pin_memory=True)
        torch.cuda.synchronize(dev)
        reps_p = 2
        t1 = time.perf_counter()
        for _ in range(reps_p):
            d_in.copy_(h_in, non_blocking=True)
            step()
            h_out.copy_(d_out, non_blocking=True)
            torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t1) / reps_p
        pcie = {"value": round(n * args.size / dt / 1e9, 4), "unit": "GB/s decompressed", "ms_per_batch": round(dt * 1e3, 1),
                "note": "pinned host buffers: H2D of the compressed bytes, decode, D2H of the output, serialised"}

    out_bytes_rank = n * args.size
    total_out = out_bytes_rank * world
    step_s = elapsed / args.steps
    value = total_out / step_s / 1e9
    k_ms = sum(kernel_ms) / len(kernel_ms)
    # HBM-side traffic cannot be read from inside the process: it comes from a separate
    # `rocprofv3 --pmc FETCH_SIZE` pass of this same command, summarised under profiles/.
    traffic, traffic_note = None, None
    pmc_path = os.path.join(ROOT, "profiles", "r01_asm_pmc_summary.json")
    if os.path.exists(pmc_path) and args.kind == "text" and n == 4096 and args.size == 1 << 20 and args.dict == 1 << 16:
        with open(pmc_path) as f:
            pmc = json.load(f)
        traffic = pmc["derived"]["fetch_bytes_per_launch"]
        traffic_note = ("L2->fabric read bytes per launch (FETCH_SIZE, rocprofv3 --pmc pass recorded in "
                        "profiles/r01_asm_pmc_summary.json; the WRITE_SIZE pass hangs rocprofv3 on the box)")
    alg_bytes = comp_total + out_bytes_rank  # per launch on this rank: compressed read once + output written once
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9

    if rank == 0:
        line = {
            "metric": "decompressed GB/s (whole node), %d x %d B LZMA streams per GPU" % (n, args.size),
            "value": round(value, 4),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(step_s * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "streams_per_s": round(n * world / step_s, 1),
            "bit_exact": bad_total == 0,
            "config": {
                "workload": "configs[1]: %d independent %d-byte .lzma streams per GPU, lc3/lp0/pb2, dict %d, "
                            "class %s, liblzma preset 6, known-size headers" % (n, args.size, args.dict, args.kind),
                "streams_per_gpu": n, "distinct_streams_per_gpu": distinct, "stream_bytes": args.size,
                "dict_size": args.dict, "class": args.kind,
                "compressed_bytes_per_gpu": comp_total, "parallelism": "streams sharded, %d per GPU" % n,
                "generation_s": round(gen_s, 1), "verified_streams_per_gpu": verified,
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_note": traffic_note,
                "kernel": "decode kernel(s) of one milzma_decode_units call", "kernel_ms": round(k_ms, 3),
                "launches_per_step": launches, "algorithmic_bytes_per_launch": alg_bytes,
                "note": "serial range-decoder dependency chain bounds this path, not HBM (DESIGN.md)",
            },
        }
        line["cpu_baseline"] = cpu_line
        if pcie is not None:
            line["pcie_inclusive"] = pcie
        print(json.dumps(line))
    ctx.close()
    if bad_total:
        raise SystemExit("bench: %d units/streams failed verification" % bad_total)


if __name__ == "__main__":
    main()
