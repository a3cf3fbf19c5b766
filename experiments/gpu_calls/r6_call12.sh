#!/bin/bash
# Round 6: push mode after the regrow's result buffers come out of the pool under one lock
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6_call12; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_streams.py tests/test_gpu_reader.py -q -x 2>&1 | tail -3 | tee $O/streams_tests.txt
timeout 600 python experiments/streams_bench.py 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print({k: d[k] for k in ('GBps', 'seconds', 'writes_s', 'finish_s', 'bad')})" | tee $O/streams_bench.txt
