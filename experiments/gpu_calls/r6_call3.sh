#!/bin/bash
# Round 6, third call: ALIGN_FREE = 5 (every label that cannot be fallen into starts a 32-byte fetch line) x the loop's global phase (ALIGN_PAD 0 .. 7;
# the shipped loop is phase 3 without ALIGN_FREE)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export MILZMA_BENCH_CACHE=/tmp/milzma_bench_cache
O=gpurun_out/r6_call3; rm -rf $O; mkdir -p $O
L=lzma_rs_amd/libmilzma.so
V=""
for v in af5p0 af5p1 af5p2 af5p4 af5p5 af5p6 af5p7; do V="$V lzma_rs_amd/variants/libmilzma_$v.so"; done
timeout 1500 python experiments/ab_bench.py --steps 4 $L $V $L $V | tee $O/ab_text.txt
