#!/bin/bash
# Round 4, fifth kernel A/B: the low length tree's update in the shadows of the pos_slot walk's head (LENDEFER), the align tree's update made
# lazily in the next pos_slot walk's shadows (ALIGNLAZY).  nodefer = neither (r4_ab4's library), nolazy = LENDEFER only.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_ab5; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size or stress or error_sites or sliced or property" 2>&1 | tail -5 | tee $O/subset.txt
L=lzma_rs_amd/libmilzma.so
V=""
for v in nodefer nolazy; do V="$V lzma_rs_amd/variants/libmilzma_$v.so"; done
timeout 900 python experiments/ab_bench.py --steps 4 $L $V $L $V $L $V | tee $O/ab_text.txt
