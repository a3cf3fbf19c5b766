#!/bin/bash
# Round 4: the streamed fuzz seed that still had one mismatch (a block decoded on demand copied over its successors' places), after the fix
cd $GRAFT_REPO_ROOT
G=gpurun_out/r4_fuzz; mkdir -p $G
MILZMA_STREAM_MIN=1,1,1 timeout 80 python experiments/parity_fuzz.py --seed 71 --rounds 4 > $G/fuzz_streamed_71.txt 2>&1; echo "streamed rc=$?"; tail -1 $G/fuzz_streamed_71.txt
