#!/bin/bash
# Round 4: the whole GPU suite (+ the new streamed-batch test first)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_suite; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "streamed_whole_file" 2>&1 | tail -12 | tee $O/new.txt
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -40 | tee $O/suite.txt
