#!/bin/bash
# What round 4 ran out of GPU minutes for; meant to be the first call of the next round (about 6 minutes):
#  1. the whole GPU suite on the final host library.  After the suite's last full run (107 passed, experiments/gpu_calls/r4_last.sh) host.cpp
#     still changed: units decoded again in another launch class are fetched from the device after a streamed launch, an on-demand .xz block
#     no longer copies over its successors' places, move lists have a page-locked buffer of their own, the wait half / milzma_crc_units /
#     milzma_move_units drain the device on every failure path, the second-part upload drains its copy stream whatever happened, a launch
#     that cannot be made a streamed one is not launched on incomplete input.  On the GPU these ran: the streamed fuzz (12 600 cases,
#     profiles/r04_parity_fuzz.txt) and 40 tests of the streamed / .xz / growable paths; the rest only on the CPU harness
#     (tests/test_host_pipeline_sanitized.py).
#  2. the differential fuzz with every batch through the streamed launch, a few more seeds;
#  3. the whole-file calls and the default bench line (the kernel sources are unchanged: profiles/r04_pmc_*.json stay valid as long as
#     bench.kernel_source_hash() says 01a4e6eafd106006).
cd $GRAFT_REPO_ROOT
O=gpurun_out/next_first; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $O/suite.txt
for seed in 81 82; do
  MILZMA_STREAM_MIN=1,1,1 timeout 120 python experiments/parity_fuzz.py --seed $seed --rounds 4 2>&1 | tail -1 | tee -a $O/fuzz_streamed.txt
done
( BATCH_VERIFY_ALL=1 timeout 200 python experiments/batch_api_bench.py 4096 512 lzma 0 2>&1 | grep -E "run|verified";
  BATCH_VERIFY_ALL=1 timeout 200 python experiments/batch_api_bench.py 1024 64 xz 0 2>&1 | grep -E "run|verified" ) | tee $O/batch_api.txt
timeout 600 python bench.py > $O/bench_default.json 2>$O/bench_default.err; tail -c 300 $O/bench_default.json
