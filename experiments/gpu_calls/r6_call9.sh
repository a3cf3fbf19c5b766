#!/bin/bash
# Round 6, ninth call: the whole-file call with the pool taken under one lock and the leads going up in four pieces; the delivery's grain
# (MILZMA_SPAN 16 .. 64 KiB for the streamed launch, MILZMA_QUANTUM for push mode)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6_call9; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_production_paths.py -q -x 2>&1 | tail -3 | tee $O/prod_tests.txt
for span in 65536 32768 16384 8192; do
  echo "## MILZMA_SPAN=$span" | tee -a $O/spans.txt
  MILZMA_TRACE=1 MILZMA_SPAN=$span timeout 600 python experiments/batch_api_bench.py 4096 64 lzma 0 2>$O/trace_$span.txt | grep "one call" | head -3 | tee -a $O/spans.txt
done
grep milzma $O/trace_65536.txt | sed -n 8,14p | sed 's/^\[milzma [^ ]* group 0\]//'
for q in 131072 65536 32768; do
  echo "## push mode, MILZMA_QUANTUM=$q" | tee -a $O/quantum.txt
  MILZMA_QUANTUM=$q timeout 600 python experiments/streams_bench.py 2>/dev/null | tail -1 | cut -c330-520 | tee -a $O/quantum.txt
done
for span in 65536 32768; do
  echo "## xz, MILZMA_SPAN=$span" | tee -a $O/spans.txt
  MILZMA_SPAN=$span timeout 600 python experiments/batch_api_bench.py 1024 32 xz 0 2>/dev/null | grep "one call" | head -3 | tee -a $O/spans.txt
done
