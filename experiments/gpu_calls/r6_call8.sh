#!/bin/bash
# Round 6, eighth call: where ONE whole-file call's 262 ms go (MILZMA_TRACE marks; page-locked result buffers vs the staging buffer)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6_call8; rm -rf $O; mkdir -p $O
MILZMA_TRACE=1 timeout 600 python experiments/batch_api_bench.py 4096 64 lzma 0 > $O/lzma_default.txt 2>$O/lzma_default_trace.txt; grep "one call" $O/lzma_default.txt | head -3; grep milzma $O/lzma_default_trace.txt | head -40 | cut -c38-120
MILZMA_PINNED_OUT=0 timeout 600 python experiments/batch_api_bench.py 4096 64 lzma 0 > $O/lzma_pageable.txt 2>&1; grep "one call" $O/lzma_pageable.txt | head -3
MILZMA_SPAN=131072 timeout 600 python experiments/batch_api_bench.py 4096 64 lzma 0 > $O/lzma_span128k.txt 2>&1; grep "one call" $O/lzma_span128k.txt | head -3
