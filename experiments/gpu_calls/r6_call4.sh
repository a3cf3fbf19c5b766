#!/bin/bash
# Round 6, fourth call: push-mode streams whose waves deliver into the result buffers while they decode; RCCL's first contact (world size 1);
# the whole suite
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export MILZMA_BENCH_CACHE=/tmp/milzma_bench_cache
O=gpurun_out/r6_call4; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_streams.py tests/test_gpu_reader.py -q -x 2>&1 | tail -30 | tee $O/streams_tests.txt
timeout 600 python experiments/streams_bench.py 2>&1 | tail -6 | tee $O/streams_bench.txt
MILZMA_PINNED_OUT=0 timeout 600 python experiments/streams_bench.py 2>&1 | tail -2 | tee $O/streams_bench_nopin.txt
timeout 600 python -m pytest tests/test_gpu_production_paths.py -q -x -k rccl 2>&1 | tail -30 | tee $O/rccl_test.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $O/suite.txt
