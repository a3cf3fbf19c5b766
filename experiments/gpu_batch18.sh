#!/bin/bash
O=gpurun_out/b18
mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/gputests.txt 2>&1
tail -4 $O/gputests.txt
python experiments/batch_api_bench.py 4096 64 lzma > $O/batch_lzma.txt 2>&1; tail -2 $O/batch_lzma.txt
python experiments/batch_api_bench.py 1024 32 xz > $O/batch_xz.txt 2>&1; tail -2 $O/batch_xz.txt
