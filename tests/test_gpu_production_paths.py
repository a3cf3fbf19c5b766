"""GPU parity at PRODUCTION size through the DEFAULT paths of the whole-file batch calls.

The streamed launch is what every `*_decompress_batch` call of >= 256 same-sized files takes; until round 5 the suite reached it
only through MILZMA_STREAM_MIN on small inputs -- and that switch was cached per process, so in a full-suite run those tests had silently
taken the classic path.  Here: >= 512 x 1 MiB `.lzma` files and >= 128 x 4 MiB `.xz` files with NO MILZMA_STREAM* setting, the path each call
took asserted through milzma_last_call_paths, and kind / message / bytes / reader position compared with the oracle for every file.
"""
import lzma
import os
import random
import struct
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

import lzma_enc as E
import lzma_rs_amd as M
import oracle_py as orc
import test_xz_literals as X
from lzma_rs_amd import workloads as W

pytestmark = pytest.mark.gpu

MIB = 1 << 20


@pytest.fixture(scope="module")
def ctx():
    for k in [k for k in os.environ if k.startswith("MILZMA_STREAM") or k in ("MILZMA_KERNEL", "MILZMA_PINNED_OUT", "MILZMA_TWO_PART", "MILZMA_SPAN")]:
        os.environ.pop(k)
    c = M.Context(0)
    yield c
    c.close()


def _check(files, refs_of, decs, what):
    assert len(decs) == len(files)
    for i, d in enumerate(decs):
        r = refs_of(i)
        assert (d.kind, d.msg) == (r.kind, r.msg), (what, i, d.kind, d.msg, r.msg)
        assert d.data == r.out, (what, i, len(d.data), len(r.out))
        assert d.in_consumed == r.in_consumed, (what, i, d.in_consumed, r.in_consumed)


def _oracle(fn, items):
    with ThreadPoolExecutor(8) as ex:
        return list(ex.map(fn, items))


def test_default_lzma_batches_at_production_size(ctx):
    """512 x 1 MiB .lzma files per call, default settings.  (a) known sizes, with a truncated and a bit-flipped member: ONE streamed launch
    with its input in two parts; (b) liblzma's native headers (no size, end marker), a few of them so compressible that the first guess of
    their output slice is wrong by orders of magnitude: streamed, then parked / moved / resumed; (c) the known-size batch with one lc + lp = 4
    member: two launch classes, hence the classic path -- the same bytes either way."""
    with ThreadPoolExecutor(8) as ex:
        plains = list(ex.map(lambda k: W.make_plain("text", MIB, seed=W.SEED0 ^ (5000 + k)), range(40)))
        known = list(ex.map(lambda p: W.compress_alone(p, dict_size=1 << 16, known_size=True), plains))
        unknown = list(ex.map(lambda p: W.compress_alone(p, dict_size=1 << 16, known_size=False), plains[:24]))
    cut = known[3][:len(known[3]) * 3 // 5]                                           # truncated: io error, what was decoded is delivered
    mid = len(known[5]) // 2
    flipped = known[5][:mid] + bytes([known[5][mid] ^ 0x10]) + known[5][mid + 1:]       # one bit: whatever the oracle makes of it
    # (a)
    distinct = known + [cut, flipped]
    refs = _oracle(orc.lzma_decompress, distinct)
    files = [distinct[i % len(distinct)] for i in range(512)]
    decs = ctx.lzma_batch(files)
    paths = ctx.last_call_paths()
    assert paths & M.PATH_STREAMED and paths & M.PATH_TWO_PART_INPUT, paths
    _check(files, lambda i: refs[i % len(distinct)], decs, "known sizes")
    assert not refs[40].ok and sum(1 for d in decs if d.ok) >= 512 - 2 * 13
    # (b)
    dense = [W.compress_alone(W.make_plain("zeros", MIB + 4096 * k, seed=k), dict_size=1 << 16, known_size=False) for k in range(3)] + \
            [W.compress_alone(W.make_plain("repeat", MIB, seed=70 + k), dict_size=1 << 16, known_size=False) for k in range(3)]
    distinct = unknown + dense
    refs = _oracle(orc.lzma_decompress, distinct)
    files = [unknown[i % len(unknown)] for i in range(506)] + dense
    decs = ctx.lzma_batch(files)
    assert ctx.last_call_paths() & M.PATH_STREAMED, ctx.last_call_paths()
    _check(files, lambda i: refs[i % len(unknown)] if i < 506 else refs[len(unknown) + i - 506], decs, "unknown sizes")
    assert all(d.ok for d in decs)
    # (c)
    plain4 = W.make_plain("text", MIB, seed=4444)
    lclp4 = W.compress_alone(plain4, dict_size=1 << 16, lc=2, lp=2, pb=2, known_size=True)
    ref4 = orc.lzma_decompress(lclp4)
    refs = _oracle(orc.lzma_decompress, known)
    files = [known[i % len(known)] for i in range(511)] + [lclp4]
    decs = ctx.lzma_batch(files)
    assert ctx.last_call_paths() & M.PATH_CLASSIC and not ctx.last_call_paths() & M.PATH_STREAMED, ctx.last_call_paths()
    _check(files, lambda i: refs[i % len(known)] if i < 511 else ref4, decs, "one lc + lp = 4 member")
    assert decs[511].ok and decs[511].data == plain4


def test_default_xz_batch_at_production_size(ctx):
    """130 x 4 MiB .xz files (four 1 MiB blocks, LZMA2 with stored chunks, CRC-64): 520 units, one streamed launch, the blocks landing in
    their files' buffers; among them a file whose Index understates a block (its unit runs past its place: fetched from the device, and
    the walk reports the Index) and a file with a damaged block check.  Every verdict is the oracle's."""
    def xz_plain(k):
        return W.make_plain("text", 2 * MIB - 100_000, seed=900 + k) + W.make_plain("random", 200_000, seed=1900 + k) + \
            W.make_plain("text", 2 * MIB - 100_000, seed=2900 + k)
    with ThreadPoolExecutor(8) as ex:
        plains = list(ex.map(xz_plain, range(8)))
        good = list(ex.map(lambda p: W.compress_xz_blocks(p, block_size=MIB, dict_size=1 << 16, check="crc64"), plains))
    # a lying Index and a bad check, built block by block (tests/test_xz_literals.py's container writer)
    blocks = [W.make_plain("text", MIB, seed=3900 + j) for j in range(4)]
    bl = [X.block(b, check=4) for b in blocks]
    liar = X.xz_file(check=4, blocks=bl, idx=X.index([(bl[0][1], bl[0][2]), (bl[1][1], bl[1][2] - 4096), (bl[2][1], bl[2][2]), (bl[3][1], bl[3][2])]))
    blb = list(bl)
    blb[1] = X.block(blocks[1], check=4, check_bytes=struct.pack("<Q", 0x0123456789ABCDEF))     # a CRC-64 that is not the block's
    bad = X.xz_file(check=4, blocks=blb, idx=X.index([(b[1], b[2]) for b in blb]))
    distinct = good + [liar, bad]
    refs = _oracle(orc.xz_decompress, distinct)
    assert refs[0].ok and not refs[8].ok and not refs[9].ok, (refs[8].msg, refs[9].msg)
    files = [good[i % 8] for i in range(128)] + [liar, bad]
    decs = ctx.xz_batch(files)
    assert ctx.last_call_paths() & M.PATH_STREAMED, ctx.last_call_paths()
    for i, d in enumerate(decs):
        r = refs[i % 8] if i < 128 else refs[8 + i - 128]
        assert (d.kind, d.msg) == (r.kind, r.msg), (i, d.msg, r.msg)
        assert d.data == r.out, (i, len(d.data), len(r.out))


@pytest.mark.parametrize("mode", ["default", "streamed", "streamed-pageable"])
def test_differential_fuzz_of_the_batch_calls(ctx, mode, monkeypatch):
    """The essential cases of experiments/parity_fuzz.py inside the suite, fixed seeds: header edits (props, dictionary size, declared size),
    byte flips, truncation, trailing garbage, concatenation on .lzma / LZMA2 / .xz inputs of many property sets -- through the default paths
    and with EVERY batch sent through the streamed launch (page-locked and pageable result buffers), which is where round 4's out-of-suite
    fuzz found wrong bytes.  Kind, message, bytes, reader position against the oracle."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "experiments"))
    import parity_fuzz as F
    if mode != "default":
        monkeypatch.setenv("MILZMA_STREAM_MIN", "1,1,1")
    if mode == "streamed-pageable":
        monkeypatch.setenv("MILZMA_PINNED_OUT", "0")
    rng = random.Random({"default": 501, "streamed": 502, "streamed-pageable": 503}[mode])
    lz = F.pool_lzma(rng)
    l2, xz = F.pool_lzma2_xz(rng)
    lclp = lambda c: (c[0] % 9) + (c[0] // 9) % 5 if c and c[0] < 225 else 0
    if mode != "default":   # (a .lzma batch is streamed only if all its members are of the lc + lp <= 3 launch class: the streamed runs keep to it)
        lz = [c for c in lz if lclp(c) <= 3]
    cases = []
    for _ in range(400):
        c = rng.choice(lz)
        if rng.random() < 0.5:
            e = F.edit_lzma_header(rng, c)
            c = e if mode == "default" or lclp(e) <= 3 else c
        if rng.random() < 0.15:
            c = c + rng.choice(lz)
        d = F.damage(rng, c, 13)
        cases.append(d if mode == "default" or lclp(d) <= 3 else c)
    streamed_seen = 0
    decs = ctx.lzma_batch(cases)
    streamed_seen += bool(ctx.last_call_paths() & M.PATH_STREAMED)
    refs = _oracle(orc.lzma_decompress, cases)
    _check(cases, lambda i: refs[i], decs, "lzma fuzz")
    cases = [F.damage(rng, rng.choice(l2), 0) for _ in range(200)]
    decs = ctx.lzma2_batch(cases)
    streamed_seen += bool(ctx.last_call_paths() & M.PATH_STREAMED)
    refs = _oracle(orc.lzma2_decompress, cases)
    _check(cases, lambda i: refs[i], decs, "lzma2 fuzz")
    cases = [F.damage(rng, rng.choice(xz), 0) for _ in range(200)]
    decs = ctx.xz_batch(cases)
    streamed_seen += bool(ctx.last_call_paths() & M.PATH_STREAMED)
    refs = _oracle(orc.xz_decompress, cases)
    for i, d in enumerate(decs):
        assert (d.kind, d.msg, d.data) == (refs[i].kind, refs[i].msg, refs[i].out), ("xz fuzz", i, d.msg, refs[i].msg)
    if mode != "default":
        assert streamed_seen == 3, "only %d of the three batches took the streamed launch" % streamed_seen


def test_rccl_first_contact_at_world_size_one():
    """no multi-GPU node has been available to any round, so RCCL itself had never run.  bench.py's rank path -- the process
    group over the real `nccl` backend (= RCCL), barriers, the MAX / SUM all-reduces and the all-gather on DEVICE tensors, the scatter of the
    compressed input from rank 0, the gather of the decoded output and its CRC check -- at world size 1 on the one GPU there is
    (MILZMA_DIST_FORCE=1): the first contact with an 8-GPU node is then not also the first contact with RCCL."""
    import json
    import socket
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MILZMA_DIST_FORCE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("MILZMA_KERNEL", "MILZMA_SPILL", "MILZMA_SLICE", "MILZMA_DIST_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--scatter", "--steps", "2", "--warmup", "1", "--streams", "96",
                        "--size", "65536", "--distinct", "24", "--no-cpu-baseline", "--other-configs", "none"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])   # (RCCL prints its own lines around it)
    assert line["ranks"]["backend"].startswith("nccl") and line["ranks"]["world"] == 1, line["ranks"]
    sg = line["scatter_gather"]
    assert sg["gathered_units_bad"] == 0 and line["value"] > 0, sg
    assert line["config"].get("verified_streams_per_gpu", line.get("verified_streams_per_gpu", 0)) == 96, line
