#!/bin/bash
# Round 6, fifth call: push mode with the written data taken straight from the caller's buffers; where a write call's time goes (MILZMA_TRACE)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6_call5; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_streams.py tests/test_gpu_reader.py -q -x 2>&1 | tail -30 | tee $O/streams_tests.txt
timeout 600 python experiments/streams_bench.py 2>&1 | tail -6 > $O/streams_bench.txt; tail -1 $O/streams_bench.txt | cut -c330-520
MILZMA_TRACE=1 timeout 600 python experiments/streams_bench.py 2> $O/trace.txt | tail -1 | cut -c330-520
grep "milzma" $O/trace.txt | tail -60
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $O/suite.txt
