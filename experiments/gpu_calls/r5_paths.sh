#!/bin/bash
# Round 5: (1) the suite on the final library; (2) the time-sliced kernel's cost apart from the loop: 4096 streams through the
# ordinary kernel, through the sliced one (MILZMA_SLICE=1: persistent waves, queue; nothing parks), parked at every quantum (MILZMA_SLICE=2);
# (3) what a streamed launch pays per span turn and per byte: the whole-file call with 64 KiB / 256 KiB / 1 MiB spans and with streaming off;
# (4) the property classes side by side; (5) the default bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export MILZMA_BENCH_CACHE=/tmp/milzma_bench_cache
O=gpurun_out/r5_paths; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/suite.txt
L=lzma_rs_amd/libmilzma.so
( echo "## ordinary launch"; timeout 300 python experiments/ab_bench.py --steps 4 $L;
  echo "## MILZMA_SLICE=1: the time-sliced kernel, nothing parks"; MILZMA_SLICE=1 timeout 300 python experiments/ab_bench.py --steps 4 $L;
  echo "## MILZMA_SLICE=2: every unit parked and taken up again at every quantum (128 KiB: 8 times per stream)"; MILZMA_SLICE=2 timeout 300 python experiments/ab_bench.py --steps 4 $L;
  echo "## MILZMA_SLICE=2 MILZMA_QUANTUM=16384: 64 times per stream"; MILZMA_SLICE=2 MILZMA_QUANTUM=16384 timeout 300 python experiments/ab_bench.py --steps 4 $L ) 2>&1 | tee $O/sliced.txt
( for span in 65536 262144 1048576; do echo "## streamed, MILZMA_SPAN=$span"; MILZMA_SPAN=$span timeout 300 python experiments/batch_api_bench.py 4096 512 lzma 0 2>&1 | grep -E "run|kernel"; done;
  echo "## MILZMA_STREAM=0 (classic)"; MILZMA_STREAM=0 timeout 300 python experiments/batch_api_bench.py 4096 512 lzma 0 2>&1 | grep -E "run|kernel";
  echo "## xz, streamed (default)"; timeout 300 python experiments/batch_api_bench.py 1024 64 xz 0 2>&1 | grep -E "run|kernel" ) | tee $O/batch_api.txt
( echo "## asm loop: lc3 = LP0 variant; lc + lp >= 4 = HBM variant (4096 x 1 MiB, greedy-parse streams of tests/lzma_enc.py)";
  timeout 900 python experiments/lclp_bench.py --streams 4096 --size 1048576 --distinct 64 3,0,2 4,0,2 2,2,4 8,0,2 8,4,4 2>/dev/null ) | tee $O/lclp_classes.txt
timeout 900 python bench.py > $O/bench_default.json 2>$O/bench_default.err; tail -c 600 $O/bench_default.json; tail -3 $O/bench_default.err
