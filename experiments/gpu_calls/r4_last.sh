#!/bin/bash
# Round 4, last call: the GPU suite on the final host library (exception-safe helper threads, rooted entry with the streamed way back) and the rooted
# bench with three shares on the one GPU (bit-exact; the pack / peer-copy in / streamed way back path).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_last; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/suite.txt
( MILZMA_MULTI_REPLICAS=3 timeout 200 python bench.py --gpus 1 --inproc --scatter --steps 3 --warmup 1 --no-cpu-baseline 2>$O/err3.txt | cut -c1-1300;
  MILZMA_MULTI_REPLICAS=3 MILZMA_ROOTED_STREAM=0 timeout 200 python bench.py --gpus 1 --inproc --scatter --steps 3 --warmup 1 --no-cpu-baseline 2>>$O/err3.txt | cut -c1-1300 ) | tee $O/rooted.txt
tail -3 $O/err3.txt
