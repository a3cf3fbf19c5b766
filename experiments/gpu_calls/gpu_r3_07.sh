#!/bin/bash
# round 3, call 7: state-after-literal by lane table (verdict 1a), GPU suite on it
O=gpurun_out/r3_07
mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/gputests.txt 2>&1
tail -4 $O/gputests.txt
python experiments/ab_bench.py --steps 4 lzma_rs_amd/variants/libmilzma_nostt.so lzma_rs_amd/libmilzma.so lzma_rs_amd/variants/libmilzma_sh2.so lzma_rs_amd/variants/libmilzma_nostt.so lzma_rs_amd/libmilzma.so > $O/ab_stt.txt 2>&1
cat $O/ab_stt.txt
python experiments/ab_bench.py --steps 3 --kind random --distinct 64 lzma_rs_amd/variants/libmilzma_nostt.so lzma_rs_amd/libmilzma.so > $O/ab_random.txt 2>&1; cat $O/ab_random.txt
