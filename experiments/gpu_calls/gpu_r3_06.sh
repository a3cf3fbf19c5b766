#!/bin/bash
# round 3, call 6: test-free direct-bit chains (DIRECT8), block order, LC4 at 12 waves/CU, pooled output buffers
O=gpurun_out/r3_06
mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/gputests.txt 2>&1
tail -4 $O/gputests.txt
python experiments/ab_bench.py --steps 4 lzma_rs_amd/variants/libmilzma_nod8.so lzma_rs_amd/libmilzma.so lzma_rs_amd/variants/libmilzma_nod8.so lzma_rs_amd/libmilzma.so > $O/ab_direct8.txt 2>&1
cat $O/ab_direct8.txt
for o in stride shuffle; do MILZMA_ORDER=$o python experiments/ab_bench.py --steps 4 lzma_rs_amd/libmilzma.so | sed "s/^/order=$o /" >> $O/ab_order.txt 2>&1; done
cat $O/ab_order.txt
python experiments/ab_bench.py --steps 3 --dict 8388608 lzma_rs_amd/variants/libmilzma_nod8.so lzma_rs_amd/libmilzma.so > $O/ab_dict8m.txt 2>&1; cat $O/ab_dict8m.txt
timeout 600 python experiments/lclp_bench.py 4,0,2 2,2,0 > $O/lclp_lc4.txt 2>&1; tail -2 $O/lclp_lc4.txt
python experiments/ab_bench.py --steps 3 --props 4,0,2 lzma_rs_amd/libmilzma.so > $O/ab_lc4.txt 2>&1; cat $O/ab_lc4.txt
timeout 900 python experiments/batch_api_bench.py 4096 64 lzma 6 > $O/batch_lzma.txt 2>&1; tail -4 $O/batch_lzma.txt
timeout 900 python experiments/batch_api_bench.py 1024 32 xz 6 > $O/batch_xz.txt 2>&1; tail -4 $O/batch_xz.txt
