#!/bin/bash
O=gpurun_out/b17
mkdir -p $O
for k in random repeat zeros; do python experiments/ab_bench.py --steps 2 --kind $k --distinct 64 lzma_rs_amd/libmilzma.so | sed "s/^/kind=$k /" >> $O/classes.txt 2>&1; done
python experiments/ab_bench.py --steps 2 --streams 8192 lzma_rs_amd/libmilzma.so | sed "s/^/streams=8192 /" >> $O/classes.txt 2>&1
cat $O/classes.txt
python experiments/batch_api_bench.py 4096 64 lzma > $O/batch_lzma.txt 2>&1; tail -3 $O/batch_lzma.txt
python experiments/batch_api_bench.py 1024 32 xz > $O/batch_xz.txt 2>&1; tail -3 $O/batch_xz.txt
