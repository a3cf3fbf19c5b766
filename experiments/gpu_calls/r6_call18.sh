#!/bin/bash
# Round 6: the kernel with 1, 2, 3 and 4 waves per SIMD (1024 .. 4096 streams of configs[1]'s recipe): the queueing model's N
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6_call18; rm -rf $O; mkdir -p $O
for n in 1024 2048 3072 4096; do
  timeout 600 python bench.py --streams $n --distinct 256 --steps 4 --warmup 1 --no-cpu-baseline --other-configs none 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print($n, d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])" | tee -a $O/waves_per_simd.txt
done
