#!/bin/bash
# Round 5: the normalisation stubs (reached only by a taken branch: 0.37 per byte in, as many out) aligned to 16 / 32 / 64 bytes -- the padding is never executed
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5_ab9; rm -rf $O; mkdir -p $O
L=lzma_rs_amd/libmilzma.so
V=""
for v in as4 as5 as6; do V="$V lzma_rs_amd/variants/libmilzma_$v.so"; done
timeout 1500 python experiments/ab_bench.py --steps 4 $L $V $L $V | tee $O/ab_text.txt
