#!/bin/bash
# round 3, call 16: do the four lanes' kernels run concurrently?  (call 15: one grouped call 0.494 s vs 0.354 s as one launch, group kernels
# 193 ms each: two at a time at best.)  Same probe with more hardware queues than the runtime's default of 4; then the PMC pass again.
cd $GRAFT_REPO_ROOT
G=gpurun_out/r3_16; mkdir -p $G
for q in 8 16; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python experiments/batch_api_bench.py 4096 16 lzma 2 > $G/batch_lzma_q$q.txt 2>&1; echo "q=$q rc=$?"; grep "one call, groups\|two in flight\|one call x" $G/batch_lzma_q$q.txt
done
export TMPDIR=/tmp
O=gpurun_out/r3_pmc
mkdir -p $O/dict8m
for attempt in 1 2 3; do
  rm -rf $O/dict8m/pass_1
  timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/dict8m/pass_1 -- python bench.py --config dict8m --steps 1 --warmup 0 --no-cpu-baseline --no-verify --other-configs none > $O/dict8m_pass_1.log 2>&1
  rc=$?
  n=$(find $O/dict8m/pass_1 -name "*counter_collection.csv" 2>/dev/null | wc -l)
  echo "dict8m pass 1 attempt $attempt rc=$rc csv=$n"
  [ "$n" -gt 0 ] && break
done
rm -rf $O/*/pass_*/*/*.db 2>/dev/null
