#!/bin/bash
# configs[4]'s whole batch (32768 x 1 MiB streams, 8 chip-fulls) on the ONE GPU there is: device-resident bench line, every unit CRC-checked;
# then 16384 files through one whole-file call (4 groups over 2 contexts)
cd $GRAFT_REPO_ROOT
G=gpurun_out/r3_large; mkdir -p $G
free -g | head -2
timeout 600 python bench.py --streams 32768 --steps 3 --warmup 1 --no-cpu-baseline --other-configs none > $G/bench_32768.json 2> $G/bench_32768.err; echo "rc=$?"; tail -c 300 $G/bench_32768.err
python - <<PY
import json
l=json.loads(open("$G/bench_32768.json").read().strip().splitlines()[-1])
print(l["value"], l["ms_per_step"], l["roofline"]["kernel_ms"], l.get("bit_exact"), l["config"])
PY
timeout 600 python - > $G/batch_16384.txt 2>&1 <<PY
import sys, time, ctypes
sys.path.insert(0, ".")
avail = [int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0] / 2**20
if avail < 80:
    print("skipped: %.0f GiB of host memory available, the probe holds ~40" % avail); sys.exit(0)
import lzma_rs_amd as M
from lzma_rs_amd import workloads as W
plains = [W.make_plain("text", 1 << 20, seed=500 + i) for i in range(16)]
comps = [W.compress_alone(p, dict_size=65536, known_size=True) for p in plains]
ctx = M.Context(0)
files = [comps[i % 16] for i in range(16384)]
for rep in range(2):
    t0 = time.time(); decs = ctx.lzma_batch(files); dt = time.time() - t0
    ok = sum(1 for i, d in enumerate(decs) if d.ok and d.data == plains[i % 16])
    print("rep %d: 16384 files in one call: %d bit-exact, %.3f s = %.2f GB/s (includes the Python binding's copies of 16 GiB)" % (rep, ok, dt, 16384 * 2**20 / dt / 1e9))
    del decs
PY
cat $G/batch_16384.txt | tail -3
