#!/bin/bash
# Round 4: growable output (park + resume).  The new tests first, then the whole GPU suite, then bench --unknown-size and the whole-file batch figures.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_grow; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "growable or wrong_guesses or grows_output" 2>&1 | tail -15 | tee $O/new_tests.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/suite.txt
timeout 600 python bench.py --unknown-size --steps 3 --warmup 1 > $O/unknown_size.json 2>$O/unknown_size.err; tail -c 2500 $O/unknown_size.json; tail -5 $O/unknown_size.err
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --other-configs none 2>/dev/null | cut -c1-300 | tee $O/bench_short.txt
