#!/bin/bash
# round 3, call 8: how many deferred instructions per shadow (state table off)
O=gpurun_out/r3_08
mkdir -p $O
V=lzma_rs_amd/variants
python experiments/ab_bench.py --steps 4 $V/libmilzma_sh0.so $V/libmilzma_sh1.so $V/libmilzma_sh2.so $V/libmilzma_sh3.so lzma_rs_amd/libmilzma.so $V/libmilzma_sh0.so $V/libmilzma_sh1.so $V/libmilzma_sh2.so $V/libmilzma_sh3.so lzma_rs_amd/libmilzma.so > $O/ab_shadow.txt 2>&1
cat $O/ab_shadow.txt
