#!/bin/bash
# Round 4, third kernel A/B: + DISP2 (three dependent vector instructions in front of the dispatch's v_readlane instead of six, one hop), scalar
# bookkeeping in the next decision's shadow (SSHADOW), a matched literal's LDS row requested early (EARLYLDS).  old = none of this round's second
# batch, prev = the library of r4_ab2.sh, noss = + DISP2 only, nolds = + DISP2 + SSHADOW.  Parity first (the error-site / property / stress tests).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_ab3; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/suite.txt
L=lzma_rs_amd/libmilzma.so
V=""
for v in old prev noss nolds; do V="$V lzma_rs_amd/variants/libmilzma_$v.so"; done
timeout 900 python experiments/ab_bench.py --steps 4 $L $V $L $V | tee $O/ab_text.txt
