#!/bin/bash
# round 3, call 2: GPU parity of the shadow-scheduled loop, A/B of the generator variants, micro-benchmark rerun
O=gpurun_out/r3_02
mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/gputests.txt 2>&1
tail -4 $O/gputests.txt
python experiments/ab_bench.py --steps 4 lzma_rs_amd/variants/libmilzma_r02.so lzma_rs_amd/variants/libmilzma_n64.so lzma_rs_amd/variants/libmilzma_ds.so lzma_rs_amd/variants/libmilzma_dt.so lzma_rs_amd/libmilzma.so lzma_rs_amd/variants/libmilzma_sh3.so lzma_rs_amd/variants/libmilzma_sh5.so lzma_rs_amd/variants/libmilzma_r02.so lzma_rs_amd/libmilzma.so > $O/ab.txt 2>&1
cat $O/ab.txt
timeout 300 ./build/shadow_slots > $O/shadow_slots.txt 2>&1; echo "micro rc=$?"
grep -E "6xB|spec6" $O/shadow_slots.txt
