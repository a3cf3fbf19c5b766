#!/bin/bash
# Round 6, after the last host-side changes (pool taken under one lock, leads in four pieces, streamed launches that announce one span; the kernel
# sources and their hash are those of r6_final.sh's evidence): the whole GPU suite, smoke(), the driver's bench command, push mode, the whole-file calls
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export MILZMA_BENCH_CACHE=/tmp/milzma_bench_cache
O=gpurun_out/r6_final2; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/suite.txt
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $O/smoke.txt
timeout 900 python bench.py > $O/bench_default.json 2>$O/bench_default.err; tail -c 300 $O/bench_default.json; tail -2 $O/bench_default.err
timeout 600 python experiments/streams_bench.py > $O/streams_bench.json 2>$O/streams_bench.err; tail -c 200 $O/streams_bench.json
timeout 600 python experiments/batch_api_bench.py 4096 64 lzma 4 > $O/batch_api_lzma.txt 2>&1; tail -10 $O/batch_api_lzma.txt
timeout 600 python experiments/batch_api_bench.py 1024 32 xz 4 > $O/batch_api_xz.txt 2>&1; tail -10 $O/batch_api_xz.txt
( MILZMA_STREAM_MIN=1,1,1 timeout 400 python experiments/parity_fuzz.py --seed 391 --rounds 3 2>&1 | tail -1 | sed "s/^/[streamed] /" ) | tee -a $O/fuzz.txt
( MILZMA_STREAM_MIN=1,1,1 MILZMA_PINNED_OUT=0 timeout 400 python experiments/parity_fuzz.py --seed 392 --rounds 3 2>&1 | tail -1 | sed "s/^/[streamed-pageable] /" ) | tee -a $O/fuzz.txt
( timeout 400 python experiments/parity_fuzz.py --seed 393 --rounds 3 2>&1 | tail -1 | sed "s/^/[default] /" ) | tee -a $O/fuzz.txt
