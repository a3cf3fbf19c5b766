#!/bin/bash
mkdir -p gpurun_out/b10
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > gpurun_out/b10/gputests.txt 2>&1
tail -4 gpurun_out/b10/gputests.txt
python bench.py --steps 5 --warmup 1 --pcie > gpurun_out/b10/bench_default.json 2> gpurun_out/b10/bench_default.err; tail -c 3000 gpurun_out/b10/bench_default.json; tail -3 gpurun_out/b10/bench_default.err
python bench.py --steps 3 --warmup 1 --config dict8m --no-cpu-baseline > gpurun_out/b10/bench_dict8m.json 2> gpurun_out/b10/bench_dict8m.err; tail -c 1500 gpurun_out/b10/bench_dict8m.json; tail -3 gpurun_out/b10/bench_dict8m.err
python bench.py --steps 3 --warmup 1 --config xz --no-cpu-baseline > gpurun_out/b10/bench_xz.json 2> gpurun_out/b10/bench_xz.err; tail -c 1500 gpurun_out/b10/bench_xz.json; tail -3 gpurun_out/b10/bench_xz.err
python bench.py --gpus 2 > gpurun_out/b10/bench_gpus2.txt 2>&1; echo "gpus2 rc=$?"; tail -2 gpurun_out/b10/bench_gpus2.txt
