#!/bin/bash
# Round 6, second call: the cursor model of Stream::write (WriteZero by the crate's partial-input buffer, headers read through Stream.tmp)
# on the GPU; then the kernel A/Bs of the round's first batch.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export MILZMA_BENCH_CACHE=/tmp/milzma_bench_cache
O=gpurun_out/r6_call2; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_streams.py tests/test_gpu_feed.py tests/test_gpu_reader.py -q 2>&1 | tail -40 | tee $O/streams_tests.txt
MILZMA_TEST_EXTRA_SEEDS=6 timeout 900 python -m pytest tests/test_gpu_streams.py -q -k "random_chunkings" 2>&1 | tail -15 | tee $O/streams_extra_seeds.txt
L=lzma_rs_amd/libmilzma.so
V=""
for v in af5 af6 as5 mixv stt; do V="$V lzma_rs_amd/variants/libmilzma_$v.so"; done
timeout 1500 python experiments/ab_bench.py --steps 4 $L $V $L $V | tee $O/ab_text.txt
