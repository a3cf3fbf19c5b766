#!/bin/bash
mkdir -p gpurun_out/b5
V=lzma_rs_amd/variants
for n in 4096 256; do MILZMA_LIB=$V/libmilzma_wp2.so python experiments/wait_prof.py $n 65536 >> gpurun_out/b5/wp2.txt 2>&1; done
cat gpurun_out/b5/wp2.txt
