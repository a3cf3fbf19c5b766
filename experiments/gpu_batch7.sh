#!/bin/bash
mkdir -p gpurun_out/b7
V=lzma_rs_amd/variants
python experiments/ab_bench.py --steps 3 $V/libmilzma_g1.so $V/libmilzma_g0.so $V/libmilzma_g1_ni.so > gpurun_out/b7/ab.txt 2>&1
cat gpurun_out/b7/ab.txt
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/b7/gputests.txt 2>&1
tail -5 gpurun_out/b7/gputests.txt
