#!/bin/bash
# Round 6: QDIRECT (direct bits behind a normalisation by a lane-parallel quotient, from 5 / 4 / 6 bits on) against the shipped loop, alternating; text at both dictionary sizes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6_call19; rm -rf $O; mkdir -p $O
L=lzma_rs_amd/libmilzma.so
V=""
for v in q5 q4 q6; do V="$V lzma_rs_amd/variants/libmilzma_$v.so"; done
timeout 1500 python experiments/ab_bench.py --steps 4 $L $V $L $V | tee $O/ab_text.txt
timeout 900 python experiments/ab_bench.py --steps 3 --dict 8388608 $L $V $L $V | tee $O/ab_dict8m.txt
