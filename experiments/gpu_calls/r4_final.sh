#!/bin/bash
# Round 4, the evidence on the binary that ships: PMC passes for configs[1], [2], [3] + kernel trace stats (r4_evidence.sh), then the default bench line.
cd $GRAFT_REPO_ROOT
bash experiments/gpu_calls/r4_evidence.sh lzma64k dict8m xz
O=gpurun_out/r4_final; mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2>$O/bench_default.err; tail -c 600 $O/bench_default.json; tail -3 $O/bench_default.err
