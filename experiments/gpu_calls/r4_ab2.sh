#!/bin/bash
# Round 4, second kernel A/B: cheaper row swap (SWAP2), dispatch by v_mad (DISPMAD), one guard for a matched literal's distance checks (MLGUARD) --
# the shipped library -- against the loop without them (old), SWAP2 alone (swaponly) and range >> 11 on the scalar ALU (r11s); then random data.
# The GPU suite first: the new loop must be bit-exact before it is timed.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_ab2; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/suite.txt
L=lzma_rs_amd/libmilzma.so
V=""
for v in old r11s swaponly; do V="$V lzma_rs_amd/variants/libmilzma_$v.so"; done
timeout 900 python experiments/ab_bench.py --steps 4 $L $V $L $V | tee $O/ab_text.txt
timeout 600 python experiments/ab_bench.py --steps 3 --kind random $L lzma_rs_amd/variants/libmilzma_old.so $L lzma_rs_amd/variants/libmilzma_old.so | tee $O/ab_random.txt
