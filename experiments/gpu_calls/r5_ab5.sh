#!/bin/bash
# Round 5, seventh call: the scalar pipe binds now (scalar + branch 93 % busy, vector 79 %: tools/emu/profile.py --pipes) -- the range < 2^24
# test of the tree walks / literal levels on the vector ALU (v_cmp + s_cbranch_vccnz instead of s_cmp + s_cbranch_scc1)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5_ab5; rm -rf $O; mkdir -p $O
L=lzma_rs_amd/libmilzma.so
V=""
for v in nvt nvl nvtl; do V="$V lzma_rs_amd/variants/libmilzma_$v.so"; done
timeout 900 python experiments/ab_bench.py --steps 4 $L $V $L $V | tee $O/ab_text.txt
