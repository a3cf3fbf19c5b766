#!/bin/bash
# Round 6: .xz batches with their input in two parts (MILZMA_TWO_PART=1) on the final library
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6_call11; rm -rf $O; mkdir -p $O
echo "## default" | tee -a $O/xz.txt
MILZMA_TRACE=1 timeout 600 python experiments/batch_api_bench.py 1024 32 xz 0 2>$O/trace_default.txt | grep "one call" | head -3 | tee -a $O/xz.txt
echo "## MILZMA_TWO_PART=1" | tee -a $O/xz.txt
MILZMA_TRACE=1 MILZMA_TWO_PART=1 timeout 600 python experiments/batch_api_bench.py 1024 32 xz 0 2>$O/trace_two_part.txt | grep "one call" | head -3 | tee -a $O/xz.txt
grep milzma $O/trace_default.txt | sed -n 8,16p | sed 's/^\[milzma [^ ]* group 0\]//'
echo ---
grep milzma $O/trace_two_part.txt | sed -n 8,16p | sed 's/^\[milzma [^ ]* group 0\]//'
