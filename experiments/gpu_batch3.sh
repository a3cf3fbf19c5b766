#!/bin/bash
mkdir -p gpurun_out/b3
V=lzma_rs_amd/variants
python experiments/ab_bench.py --steps 3 $V/libmilzma_base.so $V/libmilzma_fb_tree.so $V/libmilzma_fb_ts.so $V/libmilzma_fb_all.so $V/libmilzma_fb_tree_ns.so $V/libmilzma_fb_all_ns.so > gpurun_out/b3/ab.txt 2>&1
cat gpurun_out/b3/ab.txt
