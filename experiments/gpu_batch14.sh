#!/bin/bash
mkdir -p gpurun_out/b14
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > gpurun_out/b14/gputests.txt 2>&1
tail -5 gpurun_out/b14/gputests.txt
python experiments/ab_bench.py --steps 3 lzma_rs_amd/libmilzma.so > gpurun_out/b14/ab.txt 2>&1
python experiments/ab_bench.py --steps 2 --props 4,0,4 lzma_rs_amd/libmilzma.so >> gpurun_out/b14/ab.txt 2>&1
python experiments/ab_bench.py --steps 2 --props 4,0,2 lzma_rs_amd/libmilzma.so >> gpurun_out/b14/ab.txt 2>&1
python experiments/ab_bench.py --steps 2 --props 2,2,0 lzma_rs_amd/libmilzma.so >> gpurun_out/b14/ab.txt 2>&1
cat gpurun_out/b14/ab.txt
