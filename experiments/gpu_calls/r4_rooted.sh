#!/bin/bash
# Round 4: the one-ingest-point entry (milzma_multi_decode_units_rooted: device-to-device scatter / gather behind the C ABI)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_rooted; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multi_device" 2>&1 | tail -5 | tee $O/tests.txt
timeout 300 python bench.py --gpus 1 --inproc --scatter --steps 3 --warmup 1 --no-cpu-baseline > $O/inproc_scatter_1.json 2>$O/err1.txt; cut -c1-900 $O/inproc_scatter_1.json; tail -3 $O/err1.txt
MILZMA_MULTI_REPLICAS=3 timeout 300 python bench.py --gpus 1 --inproc --scatter --steps 3 --warmup 1 --no-cpu-baseline > $O/inproc_scatter_3replicas.json 2>$O/err3.txt; cut -c1-900 $O/inproc_scatter_3replicas.json; tail -3 $O/err3.txt
