#!/bin/bash
mkdir -p gpurun_out/b6
V=lzma_rs_amd/variants
python experiments/ab_bench.py --steps 3 $V/libmilzma_t1.so $V/libmilzma_t1_fbns.so $V/libmilzma_t0_fbns.so $V/libmilzma_t1_fb_nt.so $V/libmilzma_t1_fb_ntd.so $V/libmilzma_t1_fbs_ns.so $V/libmilzma_t1_fb_nts.so > gpurun_out/b6/ab.txt 2>&1
cat gpurun_out/b6/ab.txt
