#!/bin/bash
# many small streams: 65536 x 64 KiB (the reference's own CPU-runnable stream size, configs[0]) -- device-resident and through the
# whole-file call
cd $GRAFT_REPO_ROOT
G=gpurun_out/r3_small; mkdir -p $G
timeout 600 python bench.py --streams 65536 --size 65536 --steps 3 --warmup 1 --no-cpu-baseline --other-configs none > $G/bench_64k.json 2> $G/bench_64k.err; echo "rc=$?"; tail -c 300 $G/bench_64k.err
python - <<PY
import json
l=json.loads(open("$G/bench_64k.json").read().strip().splitlines()[-1])
print(l["value"], l["ms_per_step"], l["roofline"]["kernel_ms"], l.get("bit_exact"))
PY
timeout 600 python experiments/batch_api_bench.py 65536 64 lzma 4 65536 > $G/batch_64k.txt 2>/dev/null; echo "rc=$?"; grep -v generated $G/batch_64k.txt | grep -v "run [01]"
