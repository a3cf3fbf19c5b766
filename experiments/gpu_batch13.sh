#!/bin/bash
mkdir -p gpurun_out/b13
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > gpurun_out/b13/gputests.txt 2>&1
tail -5 gpurun_out/b13/gputests.txt
python experiments/ab_bench.py --steps 3 lzma_rs_amd/variants/libmilzma_l1.so lzma_rs_amd/libmilzma.so > gpurun_out/b13/ab.txt 2>&1
python experiments/ab_bench.py --steps 2 --props 3,0,4 lzma_rs_amd/libmilzma.so >> gpurun_out/b13/ab.txt 2>&1
python experiments/ab_bench.py --steps 2 --props 4,0,4 lzma_rs_amd/libmilzma.so >> gpurun_out/b13/ab.txt 2>&1
cat gpurun_out/b13/ab.txt
