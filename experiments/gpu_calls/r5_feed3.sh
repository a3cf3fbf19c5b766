#!/bin/bash
# Round 5: the final loop text (FEED margin fix: bit 5, lim - off) at every phase of its code in the fetch lines (q<k>; the library itself = 3);
# `nofeed` = the loop without the FEED test
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5_feed3; rm -rf $O; mkdir -p $O
L=lzma_rs_amd/libmilzma.so
V=""
for v in nofeed q0 q1 q2 q4 q5 q6 q7; do V="$V lzma_rs_amd/variants/libmilzma_$v.so"; done
timeout 1500 python experiments/ab_bench.py --steps 4 $L $V $L $V | tee $O/ab_text.txt
