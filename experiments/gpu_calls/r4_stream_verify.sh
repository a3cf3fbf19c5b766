#!/bin/bash
# Round 4: the streamed whole-file calls at full scale with EVERY file of every run checked against its plaintext (CRC-32 on the host).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_stream_verify; mkdir -p $O
( echo "## 4096 x 1 MiB .lzma"; BATCH_VERIFY_ALL=1 timeout 200 python experiments/batch_api_bench.py 4096 512 lzma 0 2>&1 | grep -E "run|verified|Error|assert" ;
  echo "## 1024 x 4 MiB .xz"; BATCH_VERIFY_ALL=1 timeout 200 python experiments/batch_api_bench.py 1024 64 xz 0 2>&1 | grep -E "run|verified|Error|assert" ) | tee $O/verify.txt
