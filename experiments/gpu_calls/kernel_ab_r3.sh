#!/bin/bash
# end-aligned last window (EOFWRAP: no off == lim test in the normalisation stubs), GPU suite on it
O=gpurun_out/r3_ab
mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gputests.txt 2>&1
tail -4 $O/gputests.txt
V=lzma_rs_amd/variants
python experiments/ab_bench.py --steps 4 $V/libmilzma_noeofw.so lzma_rs_amd/libmilzma.so $V/libmilzma_noeofw.so lzma_rs_amd/libmilzma.so > $O/ab_eofwrap.txt 2>&1
cat $O/ab_eofwrap.txt
python experiments/ab_bench.py --steps 3 --dict 8388608 $V/libmilzma_noeofw.so lzma_rs_amd/libmilzma.so > $O/ab_dict8m.txt 2>&1; cat $O/ab_dict8m.txt
