#!/bin/bash
# Round 6, first call: the push-mode changes (Partial mode behind an end marker, FAILED -> WriteZero, side batches, get_output) on the GPU:
# the streams tests first, then the whole suite, the default bench line (this box's reference figure) and the push-mode bench.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export MILZMA_BENCH_CACHE=/tmp/milzma_bench_cache
O=gpurun_out/r6_call1; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_streams.py -x -q 2>&1 | tail -30 | tee $O/streams_tests.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $O/suite.txt
timeout 600 python bench.py --steps 8 --warmup 2 > $O/bench_default.json 2>$O/bench_default.err; tail -c 600 $O/bench_default.json; tail -3 $O/bench_default.err
timeout 600 python experiments/streams_bench.py 2>&1 | tail -20 | tee $O/streams_bench.txt
