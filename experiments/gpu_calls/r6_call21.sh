#!/bin/bash
# Round 6: QDIRECT from 2 bits on against 3
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6_call21; rm -rf $O; mkdir -p $O
A=lzma_rs_amd/variants/libmilzma_q3e.so
B=lzma_rs_amd/variants/libmilzma_q2e.so
timeout 1500 python experiments/ab_bench.py --steps 4 $A $B $A $B | tee $O/ab_text.txt
