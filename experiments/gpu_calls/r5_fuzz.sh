#!/bin/bash
# Round 5: the differential fuzz (experiments/parity_fuzz.py) on the final library: default paths, every batch through the streamed launch
# (page-locked / pageable result buffers, two-part .xz input), every unit parked every few KiB, the generic kernel
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5_fuzz; rm -rf $O; mkdir -p $O
run() { name=$1; shift; ( env "$@" timeout 400 python experiments/parity_fuzz.py --seed $SEED --rounds 3 2>&1 | tail -1 | sed "s/^/[$name] /" ) | tee -a $O/fuzz.txt; }
SEED=91; run default MILZMA_X=0
SEED=92; run default MILZMA_X=0
SEED=93; run streamed MILZMA_STREAM_MIN=1,1,1
SEED=94; run streamed-pageable MILZMA_STREAM_MIN=1,1,1 MILZMA_PINNED_OUT=0
SEED=95; run streamed-two-part-xz MILZMA_STREAM_MIN=1,1,1 MILZMA_TWO_PART=1
SEED=96; run parked-every-4KiB MILZMA_SLICE=2 MILZMA_QUANTUM=4096
SEED=97; ( timeout 400 python experiments/parity_fuzz.py --seed 97 --rounds 2 --kernel generic 2>&1 | tail -1 | sed "s/^/[generic kernel] /" ) | tee -a $O/fuzz.txt
