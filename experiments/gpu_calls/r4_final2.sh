#!/bin/bash
# Round 4, the evidence on the binary that ships after the round's second kernel batch: GPU suite, PMC passes for configs[1], [2], [3] + kernel trace
# stats (r4_evidence.sh), the default bench line, the one-ingest-point entry with 1 and 3 shares, the whole-file calls, the property classes.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_final2; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/suite.txt
bash experiments/gpu_calls/r4_evidence.sh lzma64k dict8m xz
timeout 900 python bench.py > $O/bench_default.json 2>$O/bench_default.err; tail -c 400 $O/bench_default.json; tail -2 $O/bench_default.err
( timeout 300 python bench.py --gpus 1 --inproc --scatter --steps 3 --warmup 1 --no-cpu-baseline 2>$O/err1.txt | cut -c1-1200;
  MILZMA_MULTI_REPLICAS=3 timeout 300 python bench.py --gpus 1 --inproc --scatter --steps 3 --warmup 1 --no-cpu-baseline 2>$O/err3.txt | cut -c1-1200 ) | tee $O/rooted.txt
tail -2 $O/err1.txt $O/err3.txt
( echo "## lzma streamed (default)"; timeout 300 python experiments/batch_api_bench.py 4096 512 lzma 2 2>/dev/null | grep -E "run|calls" ;
  echo "## xz streamed (default)"; timeout 300 python experiments/batch_api_bench.py 1024 64 xz 2 2>/dev/null | grep -E "run|calls" ) | tee $O/batch_api.txt
timeout 600 python experiments/lclp_bench.py --streams 4096 --size 1048576 --distinct 64 3,0,2 4,0,2 8,0,2 8,4,4 2>/dev/null | tee $O/lclp_classes.txt
