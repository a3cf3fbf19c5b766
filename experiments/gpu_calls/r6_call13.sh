#!/bin/bash
# Round 6: two whole-file calls in flight on two contexts -- where the second call's time goes on the final library
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6_call13; rm -rf $O; mkdir -p $O
MILZMA_TRACE=1 timeout 600 python experiments/batch_api_bench.py 4096 64 lzma 2 > $O/two.txt 2>$O/two_trace.txt; grep "calls x" $O/two.txt
grep milzma $O/two_trace.txt | tail -45 | cut -c1-110
echo "## MILZMA_POOL_BYTES=20G"
MILZMA_POOL_BYTES=21474836480 timeout 600 python experiments/batch_api_bench.py 4096 64 lzma 2 2>/dev/null | grep "calls x"
