#!/bin/bash
# Round 4: the differential fuzz with every batch REALLY through the streamed launch (MILZMA_STREAM_MIN=1,1,1: any number of units, any number of
# bytes, any mix of sizes -- r4_fuzz.sh / r4_fuzz2.sh set only the first field, which left the 256 MiB floor in place: their "streamed" runs took
# the classic path).
cd $GRAFT_REPO_ROOT
G=gpurun_out/r4_fuzz; mkdir -p $G
MILZMA_STREAM_MIN=1,1,1 timeout 100 python experiments/parity_fuzz.py --seed 71 --rounds 4 > $G/fuzz_streamed_71.txt 2>&1; echo "streamed rc=$?"; tail -1 $G/fuzz_streamed_71.txt
MILZMA_STREAM_MIN=1,1,1 MILZMA_PINNED_OUT=0 timeout 100 python experiments/parity_fuzz.py --seed 72 --rounds 3 > $G/fuzz_streamed_pageable_72.txt 2>&1; echo "streamed pageable rc=$?"; tail -1 $G/fuzz_streamed_pageable_72.txt
MILZMA_STREAM_MIN=1,1,1 MILZMA_TWO_PART=1 MILZMA_TRACE=1 timeout 100 python experiments/parity_fuzz.py --seed 73 --rounds 3 > $G/fuzz_streamed_twopart_73.txt 2>$G/trace73.txt; echo "streamed two-part rc=$?"; tail -1 $G/fuzz_streamed_twopart_73.txt; grep -c "streamed: launch" $G/trace73.txt
