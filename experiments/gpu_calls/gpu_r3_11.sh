#!/bin/bash
# round 3, call 11: GPU suite + smoke on the final tree; one large whole-file call, grouped vs one launch
O=gpurun_out/r3_11
mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gputests.txt 2>&1
tail -4 $O/gputests.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 900 python experiments/batch_api_bench.py 4096 64 lzma 2 > $O/batch_lzma.txt 2>&1; tail -4 $O/batch_lzma.txt
timeout 900 python experiments/batch_api_bench.py 1024 32 xz 2 > $O/batch_xz.txt 2>&1; tail -4 $O/batch_xz.txt
