#!/bin/bash
# Round 6, seventh call: push mode with the delivery cut into staggered 64 KiB spans; smoke()
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6_call7; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_streams.py tests/test_gpu_reader.py tests/test_gpu_feed.py -q -x 2>&1 | tail -5 | tee $O/streams_tests.txt
timeout 600 python experiments/streams_bench.py 2>&1 | tail -6 > $O/streams_bench.txt; grep '^{' $O/streams_bench.txt | cut -c330-520
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python experiments/streams_bench.py > $O/streams_trace.json 2>$O/streams_trace.err
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/r06_streams_kernel_trace_stats.csv \;
rm -rf $O/trace
head -3 $O/r06_streams_kernel_trace_stats.csv | cut -c1-160
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3 | tee $O/smoke.txt
