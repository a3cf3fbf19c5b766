#!/bin/bash
# whole-file batch figures of the final library (results download behind the kernels; default grouping)
cd $GRAFT_REPO_ROOT
G=gpurun_out/r3_batch; mkdir -p $G
timeout 300 python experiments/batch_api_bench.py 4096 32 lzma 6 > $G/lzma.txt 2>/dev/null; echo "rc=$?"; grep -v "generated\|run [01]" $G/lzma.txt
timeout 300 python experiments/batch_api_bench.py 1024 32 xz 6 > $G/xz.txt 2>/dev/null; echo "rc=$?"; grep -v "generated\|run [01]" $G/xz.txt
python bench.py --pcie --other-configs none --no-cpu-baseline > $G/bench_pcie.json 2> $G/bench_pcie.err; echo "pcie rc=$?"; python -c "
import json; l=json.loads(open('$G/bench_pcie.json').read().strip().splitlines()[-1]); print(l['value'], l['pcie_inclusive'])"
