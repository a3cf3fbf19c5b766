"""The WHOLE host side of the library (lzma_rs_amd/csrc/host.cpp: staging, pools, streamed launches and their consumer threads, park /
regrow / resume rounds, lanes, the asynchronous halves, the multi-device entry points incl. the one-ingest-point one) under
AddressSanitizer + UndefinedBehaviorSanitizer and under ThreadSanitizer -- without a GPU.

tests/san/fake_hip.cpp is a HIP runtime made of host memory and host threads (a stream = an in-order chain of tasks on threads of their own, so
work on different streams really overlaps), tests/san/fake_kernels.cpp lets the CPU oracle decode a unit where the kernels would and keeps the
kernels' contract with the host code (results, parking, span counters, the input-ready word).  tests/san/pipeline_fuzz.cpp drives the public
entry points over valid, truncated, lying and mutated .lzma / LZMA2 / .xz inputs and compares EVERY result with the oracle (the stand-in kernels
report each error site with the real kernels' status).  What this checks is the HOST
logic -- memory safety, data races, and that its many paths all hand the caller the right bytes; parity of the real kernels is `-m gpu`'s business.
The product library never contains any of this: the fake runtime, the stand-in kernels and the oracle are linked into the test binary only."""
import lzma
import os
import random
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SAN = os.path.join(ROOT, "tests", "san")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"

from lzma_rs_amd import workloads as W  # noqa: E402


@pytest.fixture(scope="module")
def inputs(tmp_path_factory):
    d = tmp_path_factory.mktemp("pipeline_inputs")
    rnd = random.Random(5)
    for kind in ("text", "random", "repeat", "zeros"):
        for size in (0, 1, 700, 20000, 150000):
            plain = W.make_plain(kind, size, seed=rnd.randrange(1 << 30))
            for known in (True, False):
                lc, lp, pb = rnd.choice(((3, 0, 2), (0, 0, 0), (4, 0, 4), (1, 2, 3), (2, 1, 1)))
                (d / ("%s_%d_%d.lzma" % (kind, size, known))).write_bytes(
                    W.compress_alone(plain, dict_size=rnd.choice((1 << 12, 1 << 16, 1 << 20)), lc=lc, lp=lp, pb=pb, known_size=known))
            # (lc + lp = 4: an LZMA2 unit of the fast class is sent back and decoded again in another launch class -- a streamed launch's
            #  host destinations then hold nothing of it; round 4's GPU fuzz found the hand-over taking them at their word)
            for lc, lp in ((3, 0), (2, 2)):
                flt = [{"id": lzma.FILTER_LZMA2, "dict_size": 1 << 16, "lc": lc, "lp": lp, "pb": 2}]
                (d / ("%s_%d_%d%d.lzma2" % (kind, size, lc, lp))).write_bytes(lzma.compress(plain, format=lzma.FORMAT_RAW, filters=flt))
            for bs, chk in ((1 << 16, "crc64"), (1 << 14, "crc32"), (1 << 20, "none")):
                (d / ("%s_%d_%d_%s.xz" % (kind, size, bs, chk))).write_bytes(W.compress_xz_blocks(plain, block_size=bs, check=chk))
            flt = [{"id": lzma.FILTER_LZMA2, "dict_size": 1 << 16, "lc": 1, "lp": 3, "pb": 0}]
            (d / ("%s_%d_lclp4.xz" % (kind, size))).write_bytes(lzma.compress(plain, format=lzma.FORMAT_XZ, check=lzma.CHECK_CRC64, filters=flt))
    # literal-row classes liblzma cannot write (lc + lp > 4: symbol encoder of the tests) next to lc + lp = 4 streams of unknown size: in one
    # GROW batch the former finish while the latter park, and the slab's stride must outlive the launch that chose it (pipeline_fuzz: 3c)
    import lzma_enc as E
    for (lc, lp, pb), known in (((8, 0, 2), True), ((5, 2, 4), True), ((6, 0, 0), False), ((4, 0, 2), False), ((1, 3, 0), False)):
        plain = W.make_plain("text", 24000, seed=700 + lc * 10 + lp) + bytes(range(256)) * 8
        enc = E.LzmaSymbolEncoder(lc, lp, pb)
        enc.encode(E.lz_parse(plain, dict_size=1 << 16))
        if not known:
            enc.encode([("marker",)])
        (d / ("rows_%d%d%d_%d.lzma" % (lc, lp, pb, known))).write_bytes(E.lzma_header(lc, lp, pb, 1 << 16, len(plain) if known else None) + enc.finish())
    # .xz files whose Index lies about a block's size (by a little: the block still fits its slice but not its place; by a lot: its unit
    # runs out of room and the block is decoded on demand -- then COPIED into a buffer whose later places already hold the next blocks:
    # round 4's GPU fuzz found that copy running over them; ASan's memcpy-param-overlap finds it here)
    import test_xz_literals as X
    blocks = [W.make_plain("text", 60000, seed=900 + i) for i in range(3)]
    for lie in (8, 4000, 60000, -5000):
        bl = [X.block(b, check=4) for b in blocks]
        idx = X.index([(bl[0][1], bl[0][2]), (bl[1][1], bl[1][2] - lie), (bl[2][1], bl[2][2])])
        (d / ("liar_%d.xz" % lie)).write_bytes(X.xz_file(check=4, blocks=bl, idx=idx))
    # the reference's own fixtures and the 34 malformed .xz files of tests/test_xz_literals.py
    gold = os.path.join(ROOT, "tests", "golden")
    for name in os.listdir(gold):
        if name.endswith((".xz", ".lzma")):
            (d / ("golden_" + name)).write_bytes(open(os.path.join(gold, name), "rb").read())
    for name, (data, _kind, _msg) in X.CASES.items():
        (d / ("case_" + "".join(c if c.isalnum() else "_" for c in name) + ".xz")).write_bytes(data)
    return str(d)


@pytest.fixture(scope="module")
def binaries(tmp_path_factory):
    if not os.path.exists(CLANG):
        pytest.skip("no clang with sanitizer runtimes in this image")
    out = str(tmp_path_factory.mktemp("pipeline_build"))
    r = subprocess.run(["make", "-C", SAN, "-s", "-j4", "OUT=" + out, out + "/pipeline_asan", out + "/pipeline_tsan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return {"asan": out + "/pipeline_asan", "tsan": out + "/pipeline_tsan"}


# what each run sends the calls through (the library reads these once per process)
STREAMED = {"MILZMA_STREAM_MIN": "1,1,1"}          # every batch of the fast class through the streamed launch, whatever its shape
MATRIX = {
    "default": {},
    "streamed": STREAMED,
    "streamed-pageable": dict(STREAMED, MILZMA_PINNED_OUT="0"),
    "streamed-two-part-xz": dict(STREAMED, MILZMA_TWO_PART="1"),
    "streamed-three-devices": dict(STREAMED, FAKE_HIP_DEVICES="3"),
    "replicas-copy-home": {"MILZMA_MULTI_REPLICAS": "3", "MILZMA_ROOTED_STREAM": "0"},
    "groups-over-lanes": {"PIPELINE_BIG": "1", "MILZMA_LANES": "3"},
    "groups-streamed": dict(STREAMED, PIPELINE_BIG="1"),
    "generic-kernel-classes": {"MILZMA_KERNEL": "generic"},
    "always-sliced": {"MILZMA_STREAM": "0", "MILZMA_SLICE": "2"},
    # three mutations of every input on top: damaged containers are the host code's own business and are compared with the oracle
    "mutations-streamed": dict(STREAMED, PIPELINE_MUTATIONS="3"),
    "mutations-default": {"PIPELINE_MUTATIONS": "3"},
}


# (ThreadSanitizer runs several times slower: it gets the settings where threads meet -- consumer threads, lanes, devices, on-demand
#  decodes beside the parallel walks -- and one round; the rest is AddressSanitizer's)
TSAN_SKIPS = {"mutations-default", "mutations-streamed", "groups-streamed", "generic-kernel-classes", "default", "always-sliced"}
ASAN_SKIPS = {"streamed-three-devices", "replicas-copy-home", "groups-streamed"}   # (ThreadSanitizer's, or covered by a neighbour)


def _skipped(san, path):
    return (san == "tsan" and path in TSAN_SKIPS) or (san == "asan" and path in ASAN_SKIPS)


FAULT_RUNS = [("asan", "default"), ("asan", "streamed"), ("asan", "multi-two-devices"), ("tsan", "streamed")]


def _clean_env():
    env = {k: v for k, v in os.environ.items() if not k.startswith("MILZMA_")}
    env.update(ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", TSAN_OPTIONS="halt_on_error=1")
    return env


@pytest.fixture(scope="module")
def runs(binaries, inputs):
    """Every run of the matrix is a process of its own that shares nothing with the others: they are all started here, four at a time (the
    container has 8 CPUs; a run keeps one to three of them busy), and each test below only waits for its own and judges it -- the module
    takes a third of the time the runs take one after another."""
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=int(os.environ.get("MILZMA_TEST_SAN_JOBS", "4")))
    jobs = {}

    def start(key, argv, env):
        jobs[key] = pool.submit(subprocess.run, argv, capture_output=True, text=True, env=env, timeout=1800)

    # (the slow ones first: ThreadSanitizer's)
    for san in ("tsan", "asan"):
        for path in sorted(MATRIX):
            if _skipped(san, path):
                continue
            env = _clean_env()
            env.update(MATRIX[path])
            rounds = "1" if san == "tsan" or "PIPELINE_BIG" in MATRIX[path] or "PIPELINE_MUTATIONS" in MATRIX[path] else "2"
            start(("matrix", san, path), [binaries[san], inputs, rounds, "11"], env)
    for san, path in FAULT_RUNS:
        env = _clean_env()
        # (multi-two-devices: also the multi-device whole-file batch and the one-ingest-point unit call -- there fault injection found a
        #  launch still writing the caller's buffer after its failed call had returned: the wait half drains the device on every way out now)
        env.update(MATRIX.get(path, {"FAKE_HIP_DEVICES": "2", "PIPELINE_FAULTS_MULTI": "1"}))
        env["PIPELINE_FAULTS"] = "180"
        if san == "tsan":
            env["PIPELINE_FAULT_STRIDE"] = "3"     # (every third call: ThreadSanitizer's runs are the slow ones)
        start(("faults", san, path), [binaries[san], inputs, "1", "5"], env)
    yield jobs
    pool.shutdown(wait=True, cancel_futures=True)


@pytest.mark.parametrize("san", ["asan", "tsan"])
@pytest.mark.parametrize("path", sorted(MATRIX))
def test_host_pipeline_under_sanitizers(runs, san, path):
    if _skipped(san, path):
        pytest.skip("the other sanitizer's share of the matrix")
    r = runs[("matrix", san, path)].result()
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    assert "WARNING: ThreadSanitizer" not in r.stderr and "runtime error" not in r.stderr, tail
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("ok "), tail
    stats = dict(kv.split("=") for kv in last.split()[1:])
    assert int(stats["compared"]) >= 200


@pytest.mark.parametrize("san,path", FAULT_RUNS)
def test_host_pipeline_fault_injection(runs, san, path):
    """Every fallible runtime call of a batch -- allocations, copies, stream and event creation, kernel launches: about a hundred per call --
    fails once (tests/san/fake_hip.cpp: fake_hip_fail_at), one run per call and entry point.  Whatever fails, every file comes back either as
    the oracle has it or with an infrastructure error and a text; no crash, no hang (the waves' input-ready word is set on failed uploads
    too), and LeakSanitizer finds nothing left behind (it found an event leaked by milzma_create's own failure path).  Round 6: the push-mode
    calls too -- open, four writes, finish, close of a dozen streams and of enough streams for the waves to deliver into the result buffers
    (about 170 fallible calls per sequence)."""
    # (ThreadSanitizer's run: a failed piece of the second upload used to leave the earlier pieces in flight -- writing the input buffer
    #  the files decoded on their own were about to use --, and a launch that could not be made a streamed one ran on input that was
    #  not there yet: both found here, both fixed)
    r = runs[("faults", san, path)].result()
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0 and "runtime error" not in r.stderr and "WARNING: ThreadSanitizer" not in r.stderr, tail
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("ok faults=180"), tail
    stats = dict(kv.split("=") for kv in last.replace("<=", "=").split()[1:])
    assert int(stats["files_with_infra_error"]) > 30 and int(stats["compared"]) > 150
