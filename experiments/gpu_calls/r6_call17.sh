#!/bin/bash
# Round 6: the last cheap switches once more on the shipped loop (instructions per shadow 5 / 8, the priority rotation's period 2^17 / 2^19 / 2^23 cycles), alternating A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6_call17; rm -rf $O; mkdir -p $O
L=lzma_rs_amd/libmilzma.so
V=""
for v in sh8 sh5 pt19 pt23 pt17; do V="$V lzma_rs_amd/variants/libmilzma_$v.so"; done
timeout 1500 python experiments/ab_bench.py --steps 4 $L $V $L $V | tee $O/ab_text.txt
