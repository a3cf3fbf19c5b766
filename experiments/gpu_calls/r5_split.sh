#!/bin/bash
# Round 5: the library after host.cpp's split into five translation units (no behaviour change): the whole GPU suite; the headline with 4096
# DIFFERENT streams (--distinct 0); the multi-rank bench logic with two ranks on the one GPU there is -- over RCCL it must REFUSE to print a
# scaling number (two ranks, one physical device), over gloo (labelled dry run) it prints the line with the `ranks` table
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export MILZMA_BENCH_CACHE=/tmp/milzma_bench_cache
O=gpurun_out/r5_split; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/suite.txt
timeout 900 python bench.py --distinct 0 --steps 5 --warmup 1 --no-cpu-baseline --other-configs none > $O/bench_distinct0.json 2>$O/bench_distinct0.err; tail -c 400 $O/bench_distinct0.json
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 2 --warmup 1 --streams 2048 > $O/gpus2_rccl.json 2> $O/gpus2_rccl.err; echo "two ranks on one GPU over RCCL: rc=$?"; grep -h "refusing" $O/gpus2_rccl.err $O/gpus2_rccl.json | head -2
MILZMA_DIST_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 2 --steps 2 --warmup 1 --streams 2048 --scatter > $O/gpus2_gloo.json 2> $O/gpus2_gloo.err; echo "gloo dry run rc=$?"
python - <<PY
import json
l=json.loads(open("$O/gpus2_gloo.json").read().strip().splitlines()[-1])
print(l["value"], l["n_gpus"], l.get("bit_exact"), l.get("ranks"), {k: v for k, v in l.get("scatter_gather", {}).items() if k != "note"})
PY
