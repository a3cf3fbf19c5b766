"""Synthetic LZMA workloads (BASELINE.md section 4 / SURVEY.md section 8d).

The reference's own encoder cannot emit matches (src/encode/dumbencoder.rs:71-82),
so -- exactly like the reference's big fixtures (tests/files/foo.txt.lzma) -- inputs
with real match structure are produced with liblzma (Python `lzma`, `xz` CLI).

Stream i of a batch uses seed 0xC0FFEE ^ i.  Plaintext classes:
  text    4000-word random lowercase vocabulary, words of 2-9 letters, space separated
  random  uniformly random bytes (worst case: ~9 binary decisions per output byte)
  repeat  a 997-byte random block tiled (best case: length-273 matches)
  zeros   all-zero (every match overlaps its own output)
"""
import lzma
import multiprocessing
import os
import random
import struct
import subprocess

SEED0 = 0xC0FFEE


def make_plain(kind, size, seed):
    rng = random.Random(seed)
    if kind == "text":
        letters = "abcdefghijklmnopqrstuvwxyz"
        vocab = ["".join(rng.choice(letters) for _ in range(rng.randint(2, 9))) for _ in range(4000)]
        words = []
        n = 0
        while n < size:
            chunk = rng.choices(vocab, k=4096)
            words.extend(chunk)
            n += sum(len(w) + 1 for w in chunk)
        return " ".join(words).encode("ascii")[:size]
    if kind == "random":
        return rng.randbytes(size)
    if kind == "repeat":
        block = rng.randbytes(997)
        return (block * (size // 997 + 1))[:size]
    if kind == "zeros":
        return bytes(size)
    raise ValueError(kind)


def lzma1_filter(dict_size=65536, lc=3, lp=0, pb=2, preset=6):
    return [{"id": lzma.FILTER_LZMA1, "preset": preset, "dict_size": dict_size,
             "lc": lc, "lp": lp, "pb": pb}]


def compress_alone(plain, dict_size=65536, lc=3, lp=0, pb=2, preset=6, known_size=False):
    """A complete .lzma (FORMAT_ALONE) stream.  liblzma writes the 0xFF..FF size sentinel and
    an end marker; known_size=True patches the real size into the header (the reference then
    stops at that size without reading the marker)."""
    comp = lzma.compress(plain, format=lzma.FORMAT_ALONE,
                         filters=lzma1_filter(dict_size, lc, lp, pb, preset))
    if known_size:
        comp = comp[:5] + struct.pack("<Q", len(plain)) + comp[13:]
    return comp


def compress_xz_blocks(plain, block_size=1 << 20, dict_size=65536, check="crc64"):
    """Multi-block .xz via the xz CLI (Python's lzma cannot set a block size)."""
    args = ["xz", "-c", "--format=xz", "--check=" + check, "--block-size=%d" % block_size,
            "--lzma2=dict=%d,lc=3,lp=0,pb=2" % dict_size]
    return subprocess.run(args, input=plain, stdout=subprocess.PIPE, check=True).stdout


def _one_stream(args):
    kind, size, index, dict_size, known_size = args
    plain = make_plain(kind, size, SEED0 ^ index)
    return compress_alone(plain, dict_size=dict_size, known_size=known_size), plain


def _one_stream_compressed(args):
    return _one_stream(args)[0]


def make_lzma_batch(n_distinct, size=1 << 20, kind="text", dict_size=65536, known_size=True,
                    processes=None, keep_plain=False):
    """n_distinct .lzma streams (seed 0xC0FFEE ^ i), compressed on `processes` host cores.
    Returns (list of compressed streams, list of plaintexts or None)."""
    jobs = [(kind, size, i, dict_size, known_size) for i in range(n_distinct)]
    processes = processes or min(len(jobs), os.cpu_count() or 1)
    if processes > 1 and len(jobs) > 1:
        with multiprocessing.get_context("fork").Pool(processes) as pool:
            res = pool.map(_one_stream, jobs, chunksize=1)
    else:
        res = [_one_stream(j) for j in jobs]
    comps = [c for c, _ in res]
    plains = [p for _, p in res] if keep_plain else None
    return comps, plains
