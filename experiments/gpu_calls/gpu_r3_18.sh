#!/bin/bash
# round 3, call 18: grouped calls with 8 hardware queues (the runtime's default of 4 makes two of the four lanes' kernels share one)
cd $GRAFT_REPO_ROOT
G=gpurun_out/r3_18; mkdir -p $G
GPU_MAX_HW_QUEUES=8 MILZMA_TRACE=1 timeout 300 python experiments/batch_api_bench.py 4096 16 lzma 6 > $G/batch_lzma.txt 2> $G/trace.txt; echo "rc=$?"; cat $G/batch_lzma.txt
GPU_MAX_HW_QUEUES=8 timeout 300 python experiments/batch_api_bench.py 1024 16 xz 6 > $G/batch_xz.txt 2> /dev/null; echo "rc=$?"; cat $G/batch_xz.txt
