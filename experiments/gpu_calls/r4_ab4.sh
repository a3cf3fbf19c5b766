#!/bin/bash
# Round 4, fourth kernel A/B: + NBPRE (the next input byte read ahead into an SGPR; a normalisation's v_readlane is its last instruction).
# old = none of this round's second batch; nonb = the library without NBPRE; wb = + the pos_slot tree's way back in an align shadow.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_ab4; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/suite.txt
L=lzma_rs_amd/libmilzma.so
V=""
for v in old nonb wb; do V="$V lzma_rs_amd/variants/libmilzma_$v.so"; done
timeout 900 python experiments/ab_bench.py --steps 4 $L $V $L $V $L $V | tee $O/ab_text.txt
