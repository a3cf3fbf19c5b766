#!/bin/bash
mkdir -p gpurun_out/b9
V=lzma_rs_amd/variants
python experiments/ab_bench.py --steps 3 $V/libmilzma_pt20.so $V/libmilzma_pt21.so $V/libmilzma_pt22.so $V/libmilzma_pt24.so $V/libmilzma_pt26.so > gpurun_out/b9/ab.txt 2>&1
python experiments/ab_bench.py --steps 2 --wavetime $V/libmilzma_wt_pt20.so $V/libmilzma_wt_pt22.so > gpurun_out/b9/wavetime.txt 2>&1
cat gpurun_out/b9/ab.txt gpurun_out/b9/wavetime.txt
