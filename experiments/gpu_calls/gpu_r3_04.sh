#!/bin/bash
# round 3, call 4: full GPU suite (multi entry points, literal XZ expectations, at-size lc+lp), the driver's bench line with
# other_configs, the in-process multi-device bench on one GPU, property-class throughput
O=gpurun_out/r3_04
mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/gputests.txt 2>&1
tail -6 $O/gputests.txt
( time timeout 900 python bench.py ) > $O/bench_default.txt 2> $O/bench_default.err; echo "bench rc=$?"
tail -c 1800 $O/bench_default.txt; tail -3 $O/bench_default.err
( time timeout 600 python bench.py --gpus 1 --inproc --no-cpu-baseline ) > $O/bench_inproc.txt 2> $O/bench_inproc.err; echo "inproc rc=$?"
tail -c 600 $O/bench_inproc.txt; tail -3 $O/bench_inproc.err
timeout 900 python experiments/lclp_bench.py 3,0,2 4,0,2 2,2,0 8,0,2 4,4,0 > $O/lclp.txt 2>&1; echo "lclp rc=$?"
cat $O/lclp.txt | tail -6
