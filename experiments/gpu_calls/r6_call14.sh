#!/bin/bash
# Round 6: push-mode units that park for input deliver their last turn with their NEXT launch (lazy tails); copies that start behind an
# unaligned park position go out 16 bytes per lane after a byte-wise head
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6_call14; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_streams.py tests/test_gpu_reader.py tests/test_gpu_feed.py -q -x 2>&1 | tail -3 | tee $O/streams_tests.txt
timeout 600 python experiments/streams_bench.py 2>$O/streams_bench.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print({k: d[k] for k in ('GBps', 'seconds', 'writes_s', 'finish_s', 'bad')})" | tee $O/streams_bench.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $O/suite.txt
( MILZMA_TEST_EXTRA_SEEDS=12 timeout 1200 python -m pytest tests/test_gpu_streams.py -q 2>&1 | tail -2 ) | tee $O/extra_seeds.txt
timeout 600 python bench.py --steps 6 --warmup 2 --other-configs none --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel_source_sha256'])" | tee $O/bench_quick.txt
