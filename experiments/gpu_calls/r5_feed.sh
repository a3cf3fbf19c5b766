#!/bin/bash
# Round 5: fed input (MILZMA_DECODE_FEED): its GPU tests, the park / resume tests it shares code with, the default bench line (is the loop's
# extra window-refill test free?)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export MILZMA_BENCH_CACHE=/tmp/milzma_bench_cache
O=gpurun_out/r5_feed; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_feed.py -x -q 2>&1 | tail -25 | tee $O/feed.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "grow or park or wrong_guesses or time_sliced or lzma2_random" 2>&1 | tail -6 | tee $O/park.txt
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench.json 2>$O/bench.err; tail -c 400 $O/bench.json; tail -2 $O/bench.err
