#!/bin/bash
# Round 4: differential fuzz of the FINAL library against the oracle with the stronger comparison (reader position on every error path too):
# default paths, every batch through the streamed launch (MILZMA_STREAM_MIN=1), every unit parked at every 4 KiB (MILZMA_SLICE=2).
cd $GRAFT_REPO_ROOT
G=gpurun_out/r4_fuzz; mkdir -p $G
timeout 140 python experiments/parity_fuzz.py --seed 51 --rounds 4 > $G/fuzz_default_51.txt 2>&1; echo "default rc=$?"; tail -2 $G/fuzz_default_51.txt
MILZMA_STREAM_MIN=1 timeout 140 python experiments/parity_fuzz.py --seed 52 --rounds 4 > $G/fuzz_streamed_52.txt 2>&1; echo "streamed rc=$?"; tail -2 $G/fuzz_streamed_52.txt
MILZMA_SLICE=2 MILZMA_QUANTUM=4096 timeout 140 python experiments/parity_fuzz.py --seed 53 --rounds 3 > $G/fuzz_sliced_53.txt 2>&1; echo "sliced rc=$?"; tail -2 $G/fuzz_sliced_53.txt
