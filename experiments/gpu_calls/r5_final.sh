#!/bin/bash
# Round 5, final kernel (alignment pinned): evidence passes (r5_evidence.sh), the whole GPU suite, the default bench line
cd $GRAFT_REPO_ROOT
bash experiments/gpu_calls/r5_evidence.sh
export TMPDIR=/tmp
export MILZMA_BENCH_CACHE=/tmp/milzma_bench_cache
O=gpurun_out/r5_final; rm -rf $O; mkdir -p $O
cp gpurun_out/r5_ev/r05_pmc_*.json profiles/ 2>/dev/null
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/suite.txt
timeout 900 python bench.py > $O/bench_default.json 2>$O/bench_default.err; tail -c 300 $O/bench_default.json; tail -2 $O/bench_default.err
timeout 600 python bench.py --unknown-size --steps 5 --warmup 2 > $O/bench_unknown.json 2>$O/bench_unknown.err; tail -c 600 $O/bench_unknown.json
timeout 600 python bench.py --distinct 0 --steps 5 --warmup 2 --other-configs none > $O/bench_distinct0.json 2>$O/bench_distinct0.err; tail -c 300 $O/bench_distinct0.json
timeout 600 python experiments/streams_bench.py > $O/streams_bench.json 2>$O/streams_bench.err; tail -c 600 $O/streams_bench.json; tail -2 $O/streams_bench.err
