#!/bin/bash
# Round 5, another A/B on the shipped loop: presymv = the update constant of immediately updated tree levels on the vector ALU (one scalar
# instruction less per literal level 6 / 7, two vector more); prio20 / prio22 = priority rotation every 2^20 / 2^22 cycles (default 2^21);
# align8 = the loop's first instruction 256-byte aligned; dsingle / dtree = only the single decisions' / only the tree walks' updates deferred
# into shadows; stnt = non-temporal output stores
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5_ab6; rm -rf $O; mkdir -p $O
L=lzma_rs_amd/libmilzma.so
V=""
for v in presymv prio20 prio22 align8 dsingle dtree stnt; do [ -f lzma_rs_amd/variants/libmilzma_$v.so ] && V="$V lzma_rs_amd/variants/libmilzma_$v.so"; done
timeout 1200 python experiments/ab_bench.py --steps 4 $L $V $L $V | tee $O/ab_text.txt
