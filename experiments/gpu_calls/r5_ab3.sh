#!/bin/bash
# Round 5, fourth call: the running symbol of a tree walk in m0 (a v_readlane whose lane select is m0 takes the vector pipe for 4 cycles, one
# whose lane select is an SGPR for 8: profiles/r05_pipe_peaks.txt), alone and with six queued instructions per shadow; parity subset on the variant
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5_ab3; rm -rf $O; mkdir -p $O
MILZMA_LIB=$PWD/lzma_rs_amd/variants/libmilzma_m0sh6.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "asm and (full_size or stress or error_sites or sliced or property or lclp or fixtures or truncated)" 2>&1 | tail -5 | tee $O/subset_m0sh6.txt
L=lzma_rs_amd/libmilzma.so
V=""
for v in m0 m0sh6 sh6; do V="$V lzma_rs_amd/variants/libmilzma_$v.so"; done
timeout 900 python experiments/ab_bench.py --steps 4 $L $V $L $V $L $V | tee $O/ab_text.txt
timeout 600 python experiments/ab_bench.py --steps 3 --dict 8388608 $L $V $L $V | tee $O/ab_dict8m.txt
