#!/bin/bash
# Round 6, on the final sources (kernel hash c05122cda1fd8ccc -- the quotient blocks --, new seeds): the differential fuzz (experiments/parity_fuzz.py: default paths, streamed, parked every 4 KiB, generic kernel),
# the push-mode / fed / reader tests on 24 more seeds, the whole GPU suite with every fast launch time-sliced and every unit parked every 4 KiB,
# and a kernel trace of the push-mode bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6_fuzz3; rm -rf $O; mkdir -p $O
run() { name=$1; shift; ( env "$@" timeout 400 python experiments/parity_fuzz.py --seed $SEED --rounds 3 2>&1 | tail -1 | sed "s/^/[$name] /" ) | tee -a $O/fuzz.txt; }
SEED=691; run default MILZMA_X=0
SEED=692; run default MILZMA_X=0
SEED=693; run streamed MILZMA_STREAM_MIN=1,1,1
SEED=694; run streamed-pageable MILZMA_STREAM_MIN=1,1,1 MILZMA_PINNED_OUT=0
SEED=696; run parked-every-4KiB MILZMA_SLICE=2 MILZMA_QUANTUM=4096
SEED=697; ( timeout 400 python experiments/parity_fuzz.py --seed 697 --rounds 2 --kernel generic 2>&1 | tail -1 | sed "s/^/[generic kernel] /" ) | tee -a $O/fuzz.txt
( MILZMA_TEST_EXTRA_SEEDS=24 timeout 1200 python -m pytest tests/test_gpu_streams.py tests/test_gpu_feed.py tests/test_gpu_reader.py -q 2>&1 | tail -3 | sed "s/^/[push mode, fed input, reader mode: 24 more seeds] /" ) | tee -a $O/fuzz.txt
( MILZMA_SLICE=2 MILZMA_QUANTUM=4096 MILZMA_TEST_KEEP_ENV=1 timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -3 | sed "s/^/[whole suite, MILZMA_SLICE=2 MILZMA_QUANTUM=4096] /" ) | tee -a $O/fuzz.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python experiments/streams_bench.py > $O/streams_trace.json 2>$O/streams_trace.err
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/r06_streams_kernel_trace_stats.csv \;
rm -rf $O/trace
head -8 $O/r06_streams_kernel_trace_stats.csv | cut -c1-200
