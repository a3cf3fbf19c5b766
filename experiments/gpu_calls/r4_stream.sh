#!/bin/bash
# Round 4: one whole-file call with the download (and most of the upload) under the kernel: streamed launches (the waves write their
# output to the host themselves; the input's tails go up while the kernel runs; xz: spans are put in place in the files' buffers)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_stream; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_xz_literals.py -m gpu -x -q -k "wrong_guesses or grows_output or grouped_over_lanes or async_two or generated_streams or xz or multi_device" 2>&1 | tail -6 | tee $O/tests.txt
( echo "## lzma streamed (default)"; MILZMA_TRACE=1 timeout 300 python experiments/batch_api_bench.py 4096 512 lzma 2 2>$O/trace_stream.txt | grep -E "run|calls" ;
  echo "## lzma classic (MILZMA_STREAM=0)"; MILZMA_STREAM=0 timeout 300 python experiments/batch_api_bench.py 4096 512 lzma 2 2>/dev/null | grep -E "run 1|run 2|calls" ;
  echo "## xz streamed (default)"; MILZMA_TRACE=1 timeout 300 python experiments/batch_api_bench.py 1024 64 xz 2 2>$O/trace_stream_xz.txt | grep -E "run|calls" ;
  echo "## xz classic (MILZMA_STREAM=0)"; MILZMA_STREAM=0 timeout 300 python experiments/batch_api_bench.py 1024 64 xz 2 2>/dev/null | grep -E "run 1|run 2|calls" ) | tee $O/batch_api.txt
grep -E "streamed|upload" $O/trace_stream.txt | tail -8; grep -E "streamed|upload" $O/trace_stream_xz.txt | tail -8
