#!/bin/bash
mkdir -p gpurun_out/b8
V=lzma_rs_amd/variants
python experiments/ab_bench.py --steps 3 $V/libmilzma_l1.so $V/libmilzma_p9.so $V/libmilzma_p11.so $V/libmilzma_p13.so $V/libmilzma_pt16.so $V/libmilzma_pt18.so $V/libmilzma_pt20.so > gpurun_out/b8/ab.txt 2>&1
python experiments/ab_bench.py --steps 2 --wavetime $V/libmilzma_wtd.so > gpurun_out/b8/wavetime.txt 2>&1
cat gpurun_out/b8/ab.txt gpurun_out/b8/wavetime.txt
