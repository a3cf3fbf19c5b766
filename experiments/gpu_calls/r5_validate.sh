#!/bin/bash
# Round 5, sixth call: the new default loop (symbol in m0, six instructions per shadow): whole GPU suite, the default bench line, and two more
# knobs on top (scalar bookkeeping in shadows; seven instructions per shadow)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export MILZMA_BENCH_CACHE=/tmp/milzma_bench_cache
O=gpurun_out/r5_validate; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/suite.txt
L=lzma_rs_amd/libmilzma.so
V=""
for v in ss sh7; do V="$V lzma_rs_amd/variants/libmilzma_$v.so"; done
timeout 600 python experiments/ab_bench.py --steps 4 $L $V $L $V | tee $O/ab_text.txt
timeout 900 python bench.py > $O/bench_default.json 2>$O/bench_default.err; tail -c 1500 $O/bench_default.json; tail -3 $O/bench_default.err
