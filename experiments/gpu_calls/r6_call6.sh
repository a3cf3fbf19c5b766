#!/bin/bash
# Round 6, sixth call: the HB0 loop variant (lc >= 4 with lp == 0, pb <= 2): the suite, the property classes, the headline (must not move),
# and where one whole-file call's time goes (MILZMA_TRACE)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export MILZMA_BENCH_CACHE=/tmp/milzma_bench_cache
O=gpurun_out/r6_call6; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $O/suite.txt
timeout 900 python experiments/lclp_bench.py --size 1048576 --distinct 64 3,0,2 4,0,2 8,0,2 6,0,0 2,2,4 8,4,4 > $O/lclp_classes.txt 2>$O/lclp_classes.err; cat $O/lclp_classes.txt
timeout 600 python bench.py --steps 8 --warmup 2 --other-configs none --no-cpu-baseline > $O/bench_quick.json 2>$O/bench_quick.err; tail -c 300 $O/bench_quick.json | head -c 300; echo
MILZMA_TRACE=1 timeout 600 python experiments/batch_api_bench.py 4096 64 lzma 1 > $O/batch_api_lzma.txt 2>$O/batch_api_trace.txt; tail -8 $O/batch_api_lzma.txt; grep milzma $O/batch_api_trace.txt | tail -40
