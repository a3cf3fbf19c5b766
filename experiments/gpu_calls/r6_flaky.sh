#!/bin/bash
# Round 6: the GPU suite three times over on one box (is any test flaky?), then the driver's own bench command
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6_flaky; rm -rf $O; mkdir -p $O
for i in 1 2 3; do timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 | tee -a $O/suite.txt; done
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2>$O/bench_driver.err ) 2>&1 | tail -3 | tee $O/bench_time.txt
python -c "
import json
d = json.loads(open('$O/bench_driver.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_issue'].get('pmc'), d['cpu_baseline']['value'])" | tee -a $O/bench_time.txt
