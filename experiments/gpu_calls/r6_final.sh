#!/bin/bash
# Round 6, the evidence on the sources that ship: PMC passes + kernel trace (r6_evidence.sh), the whole GPU suite, the driver's bench command,
# the class matrix (text / random / repeat / zeros; the property classes), unknown sizes, push mode, the whole-file calls
cd $GRAFT_REPO_ROOT
bash experiments/gpu_calls/r6_evidence.sh
export TMPDIR=/tmp
export MILZMA_BENCH_CACHE=/tmp/milzma_bench_cache
O=gpurun_out/r6_final; rm -rf $O; mkdir -p $O
cp gpurun_out/r6_ev/r06_pmc_*.json profiles/ 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/suite.txt
timeout 900 python bench.py > $O/bench_default.json 2>$O/bench_default.err; tail -c 300 $O/bench_default.json; tail -2 $O/bench_default.err
for kind in random repeat zeros; do
  timeout 600 python bench.py --kind $kind --steps 5 --warmup 2 --other-configs none --no-cpu-baseline > $O/bench_$kind.json 2>$O/bench_$kind.err; tail -c 200 $O/bench_$kind.json | head -c 200; echo
done
timeout 900 python experiments/lclp_bench.py --size 1048576 --distinct 64 3,0,2 4,0,2 2,2,4 8,0,2 8,4,4 0,0,0 3,0,4 > $O/lclp_classes.txt 2>$O/lclp_classes.err; cat $O/lclp_classes.txt
timeout 600 python bench.py --unknown-size --steps 5 --warmup 2 > $O/bench_unknown.json 2>$O/bench_unknown.err; tail -c 300 $O/bench_unknown.json
timeout 600 python bench.py --distinct 0 --steps 5 --warmup 2 --other-configs none --no-cpu-baseline > $O/bench_distinct0.json 2>$O/bench_distinct0.err; tail -c 200 $O/bench_distinct0.json
timeout 600 python experiments/streams_bench.py > $O/streams_bench.json 2>$O/streams_bench.err; tail -c 400 $O/streams_bench.json; tail -4 $O/streams_bench.err | cut -c330-520
timeout 600 python experiments/batch_api_bench.py 4096 64 lzma 2 > $O/batch_api_lzma.txt 2>&1; tail -12 $O/batch_api_lzma.txt
timeout 600 python experiments/batch_api_bench.py 1024 32 xz 2 > $O/batch_api_xz.txt 2>&1; tail -12 $O/batch_api_xz.txt
