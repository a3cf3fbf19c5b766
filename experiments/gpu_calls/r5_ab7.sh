#!/bin/bash
# Round 5: where the loop's code falls in its 64-byte fetch lines (align8 was +1.4 %: the phase matters): the loop 64-byte aligned + 0 .. 14 dwords
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5_ab7; rm -rf $O; mkdir -p $O
L=lzma_rs_amd/libmilzma.so
V=""
for p in 0 2 4 6 8 10 12 14; do V="$V lzma_rs_amd/variants/libmilzma_a6p$p.so"; done
timeout 1500 python experiments/ab_bench.py --steps 4 $L $V $L $V | tee $O/ab_text.txt
