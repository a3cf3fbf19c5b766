#!/bin/bash
# Round 5, fifth call: on top of m0 + six queued instructions per shadow (m0sh6: 221.5 ms) -- form B for single decisions / literal levels (the
# second v_readlane costs 4 cycles of the vector pipe now, the two scalar instructions it replaces 8 of the scalar one), form A for the
# tree walks, range >> 11 on the scalar ALU, eight instructions per shadow
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5_ab4; rm -rf $O; mkdir -p $O
L=lzma_rs_amd/variants/libmilzma_m0sh6.so
V=""
for v in m6fball m6fbs m6fbl m6fa2 m0sh8 m6r11s; do V="$V lzma_rs_amd/variants/libmilzma_$v.so"; done
timeout 1200 python experiments/ab_bench.py --steps 4 $L $V $L $V | tee $O/ab_text.txt
