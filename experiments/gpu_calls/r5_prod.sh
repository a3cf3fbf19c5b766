#!/bin/bash
# Round 5: the production-size default-path tests (tests/test_gpu_production_paths.py), the streamed tests with the path asserted, then the suite
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5_prod; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "streamed" 2>&1 | tail -8 | tee $O/streamed.txt
timeout 900 python -m pytest tests -m gpu -q --durations=5 2>&1 | tail -14 | tee $O/suite.txt
