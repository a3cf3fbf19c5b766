#!/bin/bash
mkdir -p gpurun_out/b2
V=lzma_rs_amd/variants
python experiments/ab_bench.py --steps 3 $V/libmilzma_base.so $V/libmilzma_prio12.so $V/libmilzma_prio10.so $V/libmilzma_prio14.so $V/libmilzma_prio16.so $V/libmilzma_prio8.so $V/libmilzma_prios.so > gpurun_out/b2/ab.txt 2>&1
python experiments/ab_bench.py --steps 2 --wavetime $V/libmilzma_wt.so $V/libmilzma_wtprio12.so > gpurun_out/b2/wavetime.txt 2>&1
cat gpurun_out/b2/ab.txt gpurun_out/b2/wavetime.txt
