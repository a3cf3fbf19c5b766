#!/bin/bash
# round 3, call 5: spill class with full launches, whole-file batch API with two calls in flight, GPU suite on the new library
O=gpurun_out/r3_05
mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/gputests.txt 2>&1
tail -5 $O/gputests.txt
timeout 600 python experiments/lclp_bench.py 8,0,2 4,4,0 5,2,4 > $O/lclp.txt 2>&1; echo "lclp rc=$?"; tail -3 $O/lclp.txt
timeout 900 python experiments/batch_api_bench.py 4096 64 lzma 6 > $O/batch_lzma.txt 2>&1; echo "batch lzma rc=$?"; tail -5 $O/batch_lzma.txt
timeout 900 python experiments/batch_api_bench.py 1024 32 xz 6 > $O/batch_xz.txt 2>&1; echo "batch xz rc=$?"; tail -5 $O/batch_xz.txt
