#!/bin/bash
# Round 6: QDIRECT from 3 / 4 / 5 bits on, the chain's end reached straight from the quotient block; alternating against the shipped loop; then the GPU suite on the 4-bit build
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6_call20; rm -rf $O; mkdir -p $O
L=lzma_rs_amd/libmilzma.so
V=""
for v in q3e q4e q5e; do V="$V lzma_rs_amd/variants/libmilzma_$v.so"; done
timeout 1500 python experiments/ab_bench.py --steps 4 $L $V $L $V | tee $O/ab_text.txt
timeout 900 python experiments/ab_bench.py --steps 3 --dict 8388608 $L $V | tee $O/ab_dict8m.txt
timeout 600 python experiments/ab_bench.py --steps 3 --kind random $L lzma_rs_amd/variants/libmilzma_q4e.so | tee $O/ab_random.txt
MILZMA_LIB=$PWD/lzma_rs_amd/variants/libmilzma_q4e.so timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/suite_q4e.txt
