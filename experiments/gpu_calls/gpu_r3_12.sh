#!/bin/bash
# round 3, call 12: biased off / lim (the increment's carry is the refill test), GPU suite on it
O=gpurun_out/r3_12
mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gputests.txt 2>&1
tail -4 $O/gputests.txt
V=lzma_rs_amd/variants
python experiments/ab_bench.py --steps 4 $V/libmilzma_nobias.so lzma_rs_amd/libmilzma.so $V/libmilzma_nobias.so lzma_rs_amd/libmilzma.so > $O/ab_bias.txt 2>&1
cat $O/ab_bias.txt
python experiments/ab_bench.py --steps 3 --kind random --distinct 64 $V/libmilzma_nobias.so lzma_rs_amd/libmilzma.so > $O/ab_random.txt 2>&1; cat $O/ab_random.txt
