#!/bin/bash
# Round 5: the loop's alignment phase pinned (32-byte aligned + 5 dwords = libmilzma.so) against the unpinned build and its neighbours
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5_ab8; rm -rf $O; mkdir -p $O
L=lzma_rs_amd/libmilzma.so
V=""
for v in unaligned a5p4 a5p6 a5p7; do V="$V lzma_rs_amd/variants/libmilzma_$v.so"; done
timeout 1500 python experiments/ab_bench.py --steps 4 $L $V $L $V | tee $O/ab_text.txt
timeout 600 python experiments/ab_bench.py --steps 3 --dict 8388608 $L lzma_rs_amd/variants/libmilzma_unaligned.so $L lzma_rs_amd/variants/libmilzma_unaligned.so | tee $O/ab_dict8m.txt
