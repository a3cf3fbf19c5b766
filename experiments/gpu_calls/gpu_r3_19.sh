#!/bin/bash
# round 3, call 19: the default grouping (2 lanes, chip-sized groups) with the results download moved behind the kernels, and the opt-in
# 4-lane form with 16 hardware queues; the grouped / async GPU tests
cd $GRAFT_REPO_ROOT
G=gpurun_out/r3_19; mkdir -p $G
timeout 300 python experiments/batch_api_bench.py 4096 16 lzma 6 > $G/lzma_default.txt 2>/dev/null; echo "rc=$?"; grep -v "generated\|NO_GROUPS=1), run [01]\|default), run [01]" $G/lzma_default.txt
timeout 300 python experiments/batch_api_bench.py 1024 16 xz 6 > $G/xz_default.txt 2>/dev/null; echo "rc=$?"; grep -v "generated\|NO_GROUPS=1), run [01]\|default), run [01]" $G/xz_default.txt
MILZMA_LANES=4 GPU_MAX_HW_QUEUES=16 timeout 300 python experiments/batch_api_bench.py 4096 16 lzma 6 > $G/lzma_lanes4_q16.txt 2>/dev/null; echo "rc=$?"; grep -v "generated\|NO_GROUPS=1), run [01]\|default), run [01]" $G/lzma_lanes4_q16.txt
MILZMA_LANES=3 timeout 300 python experiments/batch_api_bench.py 4096 16 lzma 6 > $G/lzma_lanes3_q4.txt 2>/dev/null; echo "rc=$?"; grep -v "generated\|NO_GROUPS=1), run [01]\|default), run [01]" $G/lzma_lanes3_q4.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "grouped or async or batch" > $G/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $G/tests.txt
