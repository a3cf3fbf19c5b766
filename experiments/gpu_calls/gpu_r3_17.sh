#!/bin/bash
# round 3, call 17: where a grouped call's time goes (MILZMA_TRACE marks per lane)
cd $GRAFT_REPO_ROOT
G=gpurun_out/r3_17; mkdir -p $G
MILZMA_TRACE=1 timeout 300 python experiments/batch_api_bench.py 4096 16 lzma 2 > $G/batch_lzma.txt 2> $G/trace.txt; echo "rc=$?"; grep "one call" $G/batch_lzma.txt | head -8
grep -n "milzma" $G/trace.txt | sed -n 1,400p | awk 'NR>=1' | tail -130
