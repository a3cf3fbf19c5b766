#!/bin/bash
# Round 4: more of r4_fuzz.sh: longer runs, the streamed launch with pageable result buffers, the generic kernel, small spans.
cd $GRAFT_REPO_ROOT
G=gpurun_out/r4_fuzz; mkdir -p $G
timeout 170 python experiments/parity_fuzz.py --seed 61 --rounds 10 > $G/fuzz_default_61.txt 2>&1; echo "default rc=$?"; tail -1 $G/fuzz_default_61.txt
MILZMA_STREAM_MIN=1 MILZMA_PINNED_OUT=0 timeout 170 python experiments/parity_fuzz.py --seed 62 --rounds 8 > $G/fuzz_streamed_pageable_62.txt 2>&1; echo "streamed pageable rc=$?"; tail -1 $G/fuzz_streamed_pageable_62.txt
MILZMA_STREAM_MIN=1 MILZMA_TWO_PART=1 timeout 170 python experiments/parity_fuzz.py --seed 63 --rounds 8 > $G/fuzz_streamed_twopart_63.txt 2>&1; echo "streamed two-part rc=$?"; tail -1 $G/fuzz_streamed_twopart_63.txt
timeout 120 python experiments/parity_fuzz.py --seed 64 --rounds 4 --kernel generic > $G/fuzz_generic_64.txt 2>&1; echo "generic rc=$?"; tail -1 $G/fuzz_generic_64.txt
