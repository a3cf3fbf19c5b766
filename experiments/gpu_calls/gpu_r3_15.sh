#!/bin/bash
# round 3, call 15: whole-file batch calls cut into groups over four lanes (one call alone now overlaps its own phases): timing A/B
# against one launch per call, the grouped / async GPU tests
cd $GRAFT_REPO_ROOT
G=gpurun_out/r3_15; mkdir -p $G
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "grouped or async or batch" > $G/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $G/tests.txt
timeout 500 python experiments/batch_api_bench.py 4096 32 lzma 6 > $G/batch_lzma.txt 2>&1; echo "lzma rc=$?"; grep -v generated $G/batch_lzma.txt
timeout 500 python experiments/batch_api_bench.py 1024 32 xz 6 > $G/batch_xz.txt 2>&1; echo "xz rc=$?"; grep -v generated $G/batch_xz.txt
# the one PMC pass that hung twice in call 14 (configs[2], FETCH_SIZE); merged with that call's WRITE_SIZE / SQ passes afterwards
export TMPDIR=/tmp
O=gpurun_out/r3_pmc
for attempt in 1 2 3; do
  rm -rf $O/dict8m/pass_1
  timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/dict8m/pass_1 -- python bench.py --config dict8m --steps 1 --warmup 0 --no-cpu-baseline --no-verify --other-configs none > $O/dict8m_pass_1.log 2>&1
  rc=$?
  n=$(find $O/dict8m/pass_1 -name "*counter_collection.csv" 2>/dev/null | wc -l)
  echo "dict8m pass 1 attempt $attempt rc=$rc csv=$n"
  [ "$n" -gt 0 ] && break
done
rm -rf $O/*/pass_*/*/*.db 2>/dev/null
