#!/bin/bash
# round 3, call 3: form B for single decisions / literal levels on the shadow-scheduled loop
O=gpurun_out/r3_03
mkdir -p $O
python experiments/ab_bench.py --steps 4 lzma_rs_amd/libmilzma.so lzma_rs_amd/variants/libmilzma_fbs.so lzma_rs_amd/variants/libmilzma_fbl.so lzma_rs_amd/variants/libmilzma_fbsl.so lzma_rs_amd/variants/libmilzma_fbs_nd.so lzma_rs_amd/libmilzma.so > $O/ab.txt 2>&1
cat $O/ab.txt
