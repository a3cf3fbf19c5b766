"""lzma_rs_amd/csrc/host.cpp under AddressSanitizer + UndefinedBehaviorSanitizer (no GPU).

The host side is 2 600 lines that read hostile bytes: .lzma headers, the XZ container (stream header, block headers, index,
footer), option handling, the planners, the output pool.  tests/san builds exactly that file with -fsanitize=address,undefined
(kernel launchers stubbed: nothing is launched) and drives the GPU-free entry points -- and, through a test hook, the whole XZ
walk with the LZMA2 payloads decoded by the CPU oracle -- over the reference's fixtures, the 34 malformed .xz files of
tests/test_xz_literals.py and ten thousand mutations of them.  Wherever no payload failed, the walk's verdict (kind, message,
bytes, reader position) must equal the oracle's own XZ decoder: a differential check of the container logic on every case.
Also: milzma_free on foreign / double-freed pointers, milzma_pool_trim, the multi-device calls without a handle."""
import os
import shutil
import subprocess

import pytest

import test_xz_literals as X

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "tests", "san")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.fixture(scope="module")
def fuzzer(tmp_path_factory):
    if not os.path.exists(CLANG):
        pytest.skip("no clang with sanitizer runtimes in this image")
    out = tmp_path_factory.mktemp("san_build")
    r = subprocess.run(["make", "-C", SAN, "-s", "OUT=" + str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return os.path.join(str(out), "host_fuzz")


def _run(fuzzer, seeds, per_seed, rng_seed):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([fuzzer, str(seeds), str(per_seed), str(rng_seed)], capture_output=True, text=True, env=env, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("ok "), tail
    return dict(kv.split("=") for kv in last.split()[1:])


def test_sanitized_host_side_on_fixtures_and_malformed_files(fuzzer, tmp_path):
    seeds = tmp_path / "seeds"
    seeds.mkdir()
    gold = os.path.join(ROOT, "tests", "golden")
    for name in os.listdir(gold):
        if name.endswith((".xz", ".lzma")):
            shutil.copy(os.path.join(gold, name), seeds / name)
    for name, (data, _kind, _msg) in X.CASES.items():
        (seeds / ("case_" + "".join(c if c.isalnum() else "_" for c in name) + ".xz")).write_bytes(data)
    stats = _run(fuzzer, seeds, 0, 1)
    assert int(stats["cases"]) >= 13 + len(X.CASES) and int(stats["walks_compared"]) >= 30


def test_sanitized_host_side_mutation_fuzz(fuzzer, tmp_path):
    seeds = tmp_path / "seeds"
    seeds.mkdir()
    gold = os.path.join(ROOT, "tests", "golden")
    for name in os.listdir(gold):
        if name.endswith(".xz") or name in ("hello.txt.lzma", "foo.txt.lzma"):
            shutil.copy(os.path.join(gold, name), seeds / name)
    for name, (data, _kind, _msg) in list(X.CASES.items())[::3]:
        (seeds / ("case_" + "".join(c if c.isalnum() else "_" for c in name) + ".xz")).write_bytes(data)
    n_seeds = len(os.listdir(seeds))
    stats = _run(fuzzer, seeds, 10000 // n_seeds + 1, 20260927)
    assert int(stats["cases"]) >= 10000 and int(stats["walks_compared"]) >= 3000
