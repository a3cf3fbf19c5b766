import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import lzma_rs_amd as M, oracle_py as orc
from lzma_rs_amd import workloads as W
p = W.make_plain("text", 300_000, seed=607)
c = W.compress_alone(p, dict_size=1 << 16, known_size=True)
c7 = c[:len(c) // 2]
r = orc.lzma_decompress(c7)
print("oracle", r.kind, r.in_consumed, len(c7))
ctx = M.Context(0)
d = ctx.lzma(c7); print("single", d.kind, d.in_consumed)
d = ctx.lzma_batch([c7, c])[0]; print("batch2", d.kind, d.in_consumed)
for cut in (100, 1000, 5000, 20000, 56629):
    d = ctx.lzma(c[:cut]); r = orc.lzma_decompress(c[:cut]); print("cut", cut, d.in_consumed, r.in_consumed, d.msg == r.msg)
ctx.close()
os.environ["MILZMA_STREAM_MIN"] = "2,1"
ctx = M.Context(0)
for rep in range(3):
    ds = ctx.lzma_batch([c7] * 20 + [c] * 20)
    print("streamed", sorted(set(x.in_consumed for x in ds[:20])), sorted(set(x.in_consumed for x in ds[20:])))
ctx.close()
os.environ["MILZMA_SLICE"] = "2"; os.environ["MILZMA_QUANTUM"] = "4096"; os.environ["MILZMA_STREAM"] = "0"
ctx = M.Context(0)
d = ctx.lzma(c7); print("sliced single", d.in_consumed)
ctx.close()
