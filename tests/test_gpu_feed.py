"""GPU parity of fed input (MILZMA_DECODE_FEED): the input-side counterpart of growable output.

The reference's streaming front end (src/decode/stream.rs:223-283 `impl Write for Stream`, lzma.rs:435-524 `process_mode(Partial)`) takes the compressed
bytes piece by piece.  Here every unit's descriptor names a VIEW of its stream; a unit that comes within 20 bytes of the view's end parks
(MILZMA_ST_NEED_INPUT) and resumes on a view that starts at its first unused byte.  The tests cut streams of every kind at random places,
move the unused tails around in the input buffer (any alignment), mix input parks with room parks, and compare the end result -- verdict,
message, bytes, reader position -- with the oracle's one-shot decode of the whole stream.
"""
import lzma
import os
import random

import pytest

import lzma_enc as E
import lzma_rs_amd as M
import oracle_py as orc
import test_gpu_parity as P
from lzma_rs_amd import workloads as W

pytestmark = pytest.mark.gpu

# MILZMA_TEST_EXTRA_SEEDS=k: k more seeds per parametrized test (out-of-suite fuzz runs: profiles/r05_parity_fuzz.txt)
EXTRA = [1000 + 37 * j for j in range(int(os.environ.get("MILZMA_TEST_EXTRA_SEEDS", "0")))]


@pytest.fixture(scope="module")
def ctx():
    if os.environ.get("MILZMA_TEST_KEEP_ENV") != "1":   # (stress runs keep MILZMA_SLICE=2 / MILZMA_QUANTUM: every unit parked at every quantum)
        for k in ("MILZMA_KERNEL", "MILZMA_SPILL", "MILZMA_SLICE"):
            os.environ.pop(k, None)
    c = M.Context(0)
    yield c
    c.close()


def _feed_until_done(ctx, comps, kinds, heads, out_caps, rng, piece, relocate=True, max_rounds=4000):
    """Decodes `comps` (payload = comps[i][heads[i]:]) with fed input: every round each stream gets up to piece() more bytes; the views
    are rebuilt from scratch in a new input buffer (relocate: at random alignments) from every unit's first unused byte.  Units that park
    for room get a larger slice (their bytes moved).  Returns (results, output bytes per unit, total reader position per unit, counts)."""
    import torch
    n = len(comps)
    payload = [c[h:] for c, h in zip(comps, heads)]
    units = (M.Unit * n)()
    total = 0
    for i in range(n):
        if kinds[i] == M.KIND_RAW_LZMA:
            u, hl = M.lzma_read_header(comps[i])
            assert hl == heads[i]
        else:
            u = M.Unit()
            u.kind = M.KIND_LZMA2
        u.out_cap = out_caps[i]
        u.out_off = total
        total += (u.out_cap + 255) & ~255
        units[i] = u
    make = lambda nbytes: torch.zeros(nbytes + 512, dtype=torch.uint8, device="cuda")
    d_out = make(total)
    pos = [0] * n          # first unused byte of the stream
    avail = [0] * n        # bytes of the stream that have "arrived"
    res = None
    counts = {"rounds": 0, "input_parks": 0, "room_parks": 0, "empty_views": 0}
    live = set(range(n))   # units still parked (or not started)
    keep = []              # input tensors stay alive while a launch may read them
    while True:
        counts["rounds"] += 1
        assert counts["rounds"] < max_rounds
        blobs, off = [], 0
        for i in range(n):
            if i in live:
                if avail[i] < len(payload[i]):
                    avail[i] = min(len(payload[i]), avail[i] + piece())
                pad = rng.randrange(64) if relocate else 0
                blobs.append(bytes(pad) + payload[i][pos[i]:avail[i]])
                units[i].in_off, units[i].in_len = off + pad, avail[i] - pos[i]
                counts["empty_views"] += avail[i] == pos[i]
                last = avail[i] == len(payload[i])
                units[i].kind = kinds[i] | (M.KIND_LAST_VIEW if last else 0)
                off += len(blobs[-1])
            else:
                units[i].kind = kinds[i]
        d_in = torch.frombuffer(bytearray(b"".join(blobs) + bytes(512)), dtype=torch.uint8).cuda()
        keep = [d_in]
        flags = M.DECODE_FEED | (M.DECODE_RESUME if res is not None else 0)
        res, _, _ = ctx.decode_units_ex(units, d_in.data_ptr(), d_out.data_ptr(), flags, results=res)
        parked = [i for i in live if res[i].err_a == M.PARKED and res[i].status in (M.ST_NEED_INPUT, M.ST_OUT_FULL)]
        for i in live - set(parked):
            pos[i] += res[i].in_consumed       # (the final result: reader position within the last view)
        live = set(parked)
        if not live:
            break
        room = []
        for i in parked:
            assert res[i].in_consumed <= units[i].in_len, (i, res[i].in_consumed, units[i].in_len)
            if res[i].status == M.ST_NEED_INPUT:
                counts["input_parks"] += 1
                # within 20 bytes of the view's end (an LZMA2 unit: or in front of a packet that is not inside it), and never on the last view
                assert avail[i] < len(payload[i]) or units[i].in_len - res[i].in_consumed <= 0xFFFF + 32, i
                if kinds[i] == M.KIND_RAW_LZMA:
                    assert units[i].in_len - res[i].in_consumed < 32, (i, units[i].in_len, res[i].in_consumed)
            else:
                counts["room_parks"] += 1
                room.append(i)
            pos[i] += res[i].in_consumed
        if room:
            old = [(units[i].out_off, units[i].out_cap) for i in range(n)]
            total = 0
            for i in range(n):
                cap = units[i].out_cap
                if i in room:
                    assert res[i].out_len <= cap
                    cap = cap * 3 + 512
                units[i].out_off, units[i].out_cap = total, cap
                total += (cap + 255) & ~255
            new_out = make(total)
            lens = [min(res[i].out_len, old[i][1]) for i in range(n)]
            ctx.move_units(d_out.data_ptr(), [o[0] for o in old], new_out.data_ptr(), [units[i].out_off for i in range(n)], lens)
            d_out = new_out
    host = d_out.cpu().numpy().tobytes()
    outs = [host[units[i].out_off:units[i].out_off + min(res[i].out_flushed, units[i].out_cap)] for i in range(n)]
    del keep
    return res, outs, pos, counts


def _compare(res, outs, pos, comps, kinds, heads, refs):
    for i, ref in enumerate(refs):
        kind, msg = M.result_message(res[i], kinds[i])
        assert (kind, msg) == (ref.kind, ref.msg), (i, msg, ref.msg)
        assert outs[i] == ref.out, (i, len(outs[i]), len(ref.out))
        assert pos[i] + heads[i] == ref.in_consumed, (i, pos[i] + heads[i], ref.in_consumed, len(comps[i]))


@pytest.mark.parametrize("seed", [81, 181, 281] + EXTRA)
def test_fed_raw_streams_of_every_class(ctx, seed):
    """RAW .lzma payloads: every property class of the loop's four variants (lc + lp >= 4 with their rows in the slab), known and unknown
    sizes, a truncated and a damaged one, text / zeros / random data; pieces of 1 .. 3000 bytes, the views moved to a new place with a new
    alignment every round."""
    rng = random.Random(seed)
    comps, refs = [], []
    props = [(3, 0, 2), (0, 2, 0), (1, 1, 4), (4, 0, 2), (2, 2, 3), (8, 0, 2), (0, 4, 4), (3, 0, 2)]
    for i in range(24):
        lc, lp, pb = props[i % len(props)]
        if lc + lp > 4:     # (liblzma cannot write these: the tests' own encoder)
            c = P._rows_stream(lc, lp, pb, rng.randint(3000, 60000), seed * 10 + i, i % 3 == 0)
        else:
            p = W.make_plain(rng.choice(["text", "text", "zeros", "repeat", "random"]), rng.randint(1, 60000), seed=seed * 10 + i)
            c = W.compress_alone(p, dict_size=rng.choice([4096, 1 << 16]), known_size=(i % 3 == 0), lc=lc, lp=lp, pb=pb)
        if i % 7 == 5:
            c = c[:13 + (len(c) - 13) * 2 // 3]                     # truncated: UnexpectedEof on the LAST view, not before
        if i % 7 == 6:
            k = 13 + (len(c) - 13) // 2
            c = c[:k] + bytes([c[k] ^ 0x41]) + c[k + 1:]            # damaged in the middle: whatever the oracle makes of it
        comps.append(c)
        refs.append(orc.lzma_decompress(c))
    comps.append(E.lzma_header(3, 0, 2, 1 << 16, 0) + bytes(5))     # an empty stream of known size: five bytes of range coder
    refs.append(orc.lzma_decompress(comps[-1]))
    kinds = [M.KIND_RAW_LZMA] * len(comps)
    heads = [13] * len(comps)
    caps = [max(len(r.out), 1) + 600 for r in refs]
    res, outs, pos, counts = _feed_until_done(ctx, comps, kinds, heads, caps, rng, lambda: rng.choice([1, 7, 40, 300, 3000]))
    _compare(res, outs, pos, comps, kinds, heads, refs)
    assert counts["input_parks"] > 200, counts
    assert sum(1 for r in refs if not r.ok) >= 3


@pytest.mark.parametrize("seed", [82, 182, 282] + [e + 1 for e in EXTRA])
def test_fed_lzma2_units_with_every_packet_kind(ctx, seed):
    """LZMA2 units: liblzma's streams (compressed chunks, stored chunks for random data) and hand-made packet sequences (every control
    byte, property switches up to lc + lp = 4, dictionary resets, stored chunks, chunks a few symbols long); the views end inside packet
    headers, inside stored chunks, inside the five bytes of a range-coder start and inside compressed chunks."""
    rng = random.Random(seed)
    comps = [lzma.compress(W.make_plain("text", 150000, seed=19) + W.make_plain("random", 70000, seed=seed) + b"xyz" * 20000, format=lzma.FORMAT_RAW,
                           filters=[{"id": lzma.FILTER_LZMA2, "dict_size": 1 << 16}]),
             lzma.compress(W.make_plain("random", 100000, seed=seed + 1), format=lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA2, "dict_size": 1 << 12}]),
             lzma.compress(W.make_plain("text", 90000, seed=77), format=lzma.FORMAT_RAW,
                           filters=[{"id": lzma.FILTER_LZMA2, "dict_size": 1 << 16, "lc": 2, "lp": 2, "pb": 2}])]
    comps += [P.random_lzma2_stream(rng) for _ in range(40)]
    c = comps[0]
    comps.append(c[:len(c) // 2])                                   # ends inside a chunk
    comps.append(c[:len(c) - 1])                                    # the end byte is missing
    comps.append(comps[5] + b"trailing bytes behind the end byte")  # the walk stops at the end byte: reader position in front of them
    refs = [orc.lzma2_decompress(c) for c in comps]
    kinds = [M.KIND_LZMA2] * len(comps)
    heads = [0] * len(comps)
    caps = [max(len(r.out), 1) + 600 for r in refs]
    res, outs, pos, counts = _feed_until_done(ctx, comps, kinds, heads, caps, rng, lambda: rng.choice([1, 2, 5, 17, 100, 700, 5000]))
    _compare(res, outs, pos, comps, kinds, heads, refs)
    assert counts["input_parks"] > 300, counts


@pytest.mark.parametrize("seed", [83, 183, 283] + [e + 2 for e in EXTRA])
def test_fed_input_and_growable_output_together(ctx, seed):
    """Both parking reasons in one batch: slices of a few hundred bytes and pieces of a few hundred bytes, RAW and LZMA2 units.  A unit
    parked for room by a FEED call resumes on a re-based view like one parked for input."""
    rng = random.Random(seed)
    comps, kinds, heads, refs = [], [], [], []
    for i in range(12):
        lc, lp, pb = [(3, 0, 2), (4, 0, 2), (1, 2, 3), (3, 0, 2)][i % 4]
        p = W.make_plain(["text", "repeat", "zeros"][i % 3], rng.randint(2000, 50000), seed=seed * 11 + i)
        comps.append(W.compress_alone(p, dict_size=1 << 16, known_size=(i % 2 == 0), lc=lc, lp=lp, pb=pb))
        kinds.append(M.KIND_RAW_LZMA)
        heads.append(13)
        refs.append(orc.lzma_decompress(comps[-1]))
    for i in range(8):
        comps.append(P.random_lzma2_stream(rng, max_chunks=12, max_syms=400))
        kinds.append(M.KIND_LZMA2)
        heads.append(0)
        refs.append(orc.lzma2_decompress(comps[-1]))
    caps = [[300, 1000, 4096][i % 3] for i in range(len(comps))]
    res, outs, pos, counts = _feed_until_done(ctx, comps, kinds, heads, caps, rng, lambda: rng.choice([50, 400, 1500]))
    _compare(res, outs, pos, comps, kinds, heads, refs)
    assert counts["input_parks"] > 30 and counts["room_parks"] > 20, counts


def test_fed_views_in_place_and_the_flag_less_last_call(ctx):
    """The other way to feed: ONE input buffer, longer views of it (no relocation); the batch's last call carries no FEED flag at all
    (every view is then the last).  Also: a view that brings nothing new (empty) parks the unit again without harm, and a FEED call on
    a context made for the generic kernel is refused."""
    import torch
    rng = random.Random(84)
    plain = [W.make_plain("text", 40000 + 5000 * i, seed=950 + i) for i in range(6)]
    comps = [W.compress_alone(p, dict_size=1 << 16, known_size=False) for p in plain]
    payload = [c[13:] for c in comps]
    n = len(comps)
    units = (M.Unit * n)()
    offs, off, total = [], 0, 0
    for i in range(n):
        u, _ = M.lzma_read_header(comps[i])
        u.out_off, u.out_cap = total, len(plain[i]) + 600
        total += (u.out_cap + 255) & ~255
        units[i] = u
        offs.append(off)
        off += (len(payload[i]) + 255) & ~255
    d_in = torch.frombuffer(bytearray(b"".join(p + bytes((-len(p)) % 256) for p in payload) + bytes(512)), dtype=torch.uint8).cuda()
    d_out = torch.zeros(total + 512, dtype=torch.uint8, device="cuda")
    pos, res = [0] * n, None
    for step, frac in enumerate([0.0, 0.1, 0.1, 0.5, 0.9]):        # (0.0: an empty first view; 0.1 twice: a view with nothing new)
        for i in range(n):
            end = int(len(payload[i]) * frac)
            units[i].in_off, units[i].in_len = offs[i] + pos[i], max(end, pos[i]) - pos[i]
        res, _, _ = ctx.decode_units_ex(units, d_in.data_ptr(), d_out.data_ptr(), M.DECODE_FEED | (M.DECODE_RESUME if res is not None else 0),
                                        results=res)
        for i in range(n):
            assert (res[i].status, res[i].err_a) == (M.ST_NEED_INPUT, M.PARKED), (step, i, res[i].status)
            pos[i] += res[i].in_consumed
    for i in range(n):
        units[i].in_off, units[i].in_len = offs[i] + pos[i], len(payload[i]) - pos[i]
    res, _, _ = ctx.decode_units_ex(units, d_in.data_ptr(), d_out.data_ptr(), M.DECODE_RESUME, results=res)
    host = d_out.cpu().numpy().tobytes()
    for i in range(n):
        assert res[i].status == M.ST_OK, (i, res[i].status)
        assert host[units[i].out_off:units[i].out_off + res[i].out_len] == plain[i], i
        assert pos[i] + res[i].in_consumed == len(payload[i]), i
    os.environ["MILZMA_KERNEL"] = "generic"
    try:
        g = M.Context(0)
        with pytest.raises(M.InfraError):
            g.decode_units_ex(units, d_in.data_ptr(), d_out.data_ptr(), M.DECODE_FEED)
        g.close()
    finally:
        os.environ.pop("MILZMA_KERNEL")


def test_fed_batch_of_many_units_in_place(ctx):
    """More units than the chip holds waves for at once would be 4096; 1200 x 256 KiB here (the queue, the parking lot and the result upload of
    a RESUME at a size where units share waves): every unit's views end at its own random places, longer views of ONE buffer; a third
    of the units get output slices that are too small as well.  Bytes against the plain text."""
    import torch
    rng = random.Random(85)
    plains = [W.make_plain("text", 256 << 10, seed=990 + k) for k in range(24)]
    comps = [W.compress_alone(p, dict_size=1 << 16, known_size=(k % 2 == 0)) for k, p in enumerate(plains)]
    payload = [c[13:] for c in comps]
    reader_end = [orc.lzma_decompress(c).in_consumed - 13 for c in comps]   # (a stream of known size ends in front of its end marker)
    n = 1200
    blob_offs, off = [], 0
    for p in payload:
        blob_offs.append(off)
        off += (len(p) + 255) & ~255
    d_in = torch.frombuffer(bytearray(b"".join(p + bytes((-len(p)) % 256) for p in payload) + bytes(512)), dtype=torch.uint8).cuda()
    units = (M.Unit * n)()
    total = 0
    for i in range(n):
        u, _ = M.lzma_read_header(comps[i % 24])
        u.out_cap = (256 << 10) + 512 if i % 3 else 100_000
        u.out_off = total
        total += (u.out_cap + 255) & ~255
        units[i] = u
    d_out = torch.zeros(total + 512, dtype=torch.uint8, device="cuda")
    used, res, live, rounds = [0] * n, None, list(range(n)), 0
    cuts = [sorted(rng.randint(0, len(payload[i % 24])) for _ in range(3)) + [len(payload[i % 24])] for i in range(n)]
    parks = {M.ST_NEED_INPUT: 0, M.ST_OUT_FULL: 0}
    while live:
        assert rounds < 12
        for i in live:
            end = cuts[i][min(rounds, 3)]
            units[i].in_off, units[i].in_len = blob_offs[i % 24] + used[i], max(end, used[i]) - used[i]
            units[i].kind = M.KIND_RAW_LZMA | (M.KIND_LAST_VIEW if rounds >= 3 else 0)
        res, _, _ = ctx.decode_units_ex(units, d_in.data_ptr(), d_out.data_ptr(), M.DECODE_FEED | (M.DECODE_RESUME if res is not None else 0),
                                        results=res)
        rounds += 1
        live = [i for i in live if res[i].err_a == M.PARKED and res[i].status in parks]
        room = [i for i in live if res[i].status == M.ST_OUT_FULL]
        for i in live:
            used[i] += res[i].in_consumed
            parks[res[i].status] += 1
        if room:    # (every slice in a new place, the parked units' bytes moved: as in the GROW tests)
            old = [(units[i].out_off, units[i].out_cap) for i in range(n)]
            total = 0
            for i in range(n):
                cap = (256 << 10) + 512 if i in set(room) else units[i].out_cap
                units[i].out_off, units[i].out_cap = total, cap
                total += (cap + 255) & ~255
            new_out = torch.zeros(total + 512, dtype=torch.uint8, device="cuda")
            lens = [min(res[i].out_len, old[i][1]) if (res[i].status == M.ST_OK or res[i].err_a == M.PARKED) else 0 for i in range(n)]
            ctx.move_units(d_out.data_ptr(), [o[0] for o in old], new_out.data_ptr(), [units[i].out_off for i in range(n)], lens)
            d_out = new_out
    host = d_out.cpu().numpy().tobytes()
    for i in range(n):
        assert res[i].status == M.ST_OK, (i, res[i].status)
        assert host[units[i].out_off:units[i].out_off + res[i].out_len] == plains[i % 24], i
        assert used[i] + res[i].in_consumed == reader_end[i % 24], i
    assert parks[M.ST_NEED_INPUT] > 2 * n and parks[M.ST_OUT_FULL] >= n // 3 - 5, parks
