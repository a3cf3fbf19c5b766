#!/bin/bash
# the multi-rank bench paths end to end on the one GPU there is: two ranks (gloo for the rendezvous / reductions / scatter / gather, staged
# through the host; both ranks decode on GPU 0): plain --gpus 2 and --scatter (rank 0 partitions a pool of 1024 different streams, ships
# the shares, every rank decodes ITS streams, rank 0 CRC-checks every gathered unit).  Everything but RCCL itself.
cd $GRAFT_REPO_ROOT
G=gpurun_out/r3_ranks; mkdir -p $G
export MILZMA_DIST_BACKEND=gloo
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 2 --warmup 1 > $G/gpus2.json 2> $G/gpus2.err; echo "gpus2 rc=$?"; tail -1 $G/gpus2.json | cut -c1-600
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 2 --steps 2 --warmup 1 --scatter > $G/scatter2.json 2> $G/scatter2.err; echo "scatter rc=$?"; tail -c 600 $G/scatter2.err; python - <<PY
import json
l=json.loads(open("$G/scatter2.json").read().strip().splitlines()[-1])
print(l["value"], l["n_gpus"], l.get("bit_exact"), l.get("scatter_gather"))
PY
