#!/bin/bash
mkdir -p gpurun_out/b12
for pad in 0 8192 16384 65536; do
  MILZMA_LDS_PAD=$pad python experiments/ab_bench.py --steps 2 lzma_rs_amd/libmilzma.so | sed "s/^/lds_pad=$pad /" >> gpurun_out/b12/lds_pad.txt
done
cat gpurun_out/b12/lds_pad.txt
