#!/bin/bash
mkdir -p gpurun_out/b4
V=lzma_rs_amd/variants
python experiments/ab_bench.py --steps 3 $V/libmilzma_base.so $V/libmilzma_pads1.so $V/libmilzma_padv1.so $V/libmilzma_padb1.so $V/libmilzma_norms.so > gpurun_out/b4/ab.txt 2>&1
for d in 65536 4096; do for n in 4096 256; do MILZMA_LIB=$V/libmilzma_wp.so python experiments/wait_prof.py $n $d >> gpurun_out/b4/wp.txt 2>&1; done; done
python experiments/ab_bench.py --steps 2 --dict 4096 $V/libmilzma_base.so >> gpurun_out/b4/ab.txt 2>&1
cat gpurun_out/b4/ab.txt gpurun_out/b4/wp.txt
