#!/bin/bash
# Round 6, tenth call: streamed launches that announce only their first span when the result buffers are page-locked; push mode's quantum
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6_call10; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_production_paths.py tests/test_gpu_parity.py -q -x -k "production or streamed or stream or fuzz or xz" 2>&1 | tail -3 | tee $O/tests.txt
echo "## lzma" | tee -a $O/batch.txt
timeout 600 python experiments/batch_api_bench.py 4096 64 lzma 2 2>/dev/null | grep -v "^$" | tail -9 | tee -a $O/batch.txt
echo "## xz" | tee -a $O/batch.txt
timeout 600 python experiments/batch_api_bench.py 1024 32 xz 2 2>/dev/null | grep -v "^$" | tail -9 | tee -a $O/batch.txt
for q in 131072 65536 32768; do
  echo "## push mode, MILZMA_QUANTUM=$q" | tee -a $O/quantum.txt
  MILZMA_QUANTUM=$q timeout 600 python experiments/streams_bench.py 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print({k: d[k] for k in ('GBps', 'seconds', 'writes_s', 'finish_s', 'bad')})" | tee -a $O/quantum.txt
done
( MILZMA_STREAM_MIN=1,1,1 timeout 400 python experiments/parity_fuzz.py --seed 293 --rounds 2 2>&1 | tail -1 | sed "s/^/[streamed] /" ) | tee -a $O/fuzz.txt
