#!/bin/bash
# the GPU test suite + smoke on the final tree, then the round's kernel work in ONE A/B on one box: the generator with every round-3
# switch off (tools/build_variants.py "r02like:NORM64=0,DEFER=none,DIRECT8=0,OFFBIAS=0,EOFWRAP=0") against the shipped library,
# alternating (experiments/ab_bench.py: median kernel ms of the bench batch), dict 64 KiB and dict 8 MiB; then the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_ab
mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gputests.txt 2>&1
tail -4 $O/gputests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
V=lzma_rs_amd/variants
python experiments/ab_bench.py --steps 4 $V/libmilzma_r02like.so lzma_rs_amd/libmilzma.so $V/libmilzma_r02like.so lzma_rs_amd/libmilzma.so > $O/ab_round.txt 2>&1
cat $O/ab_round.txt
python experiments/ab_bench.py --steps 3 --dict 8388608 $V/libmilzma_r02like.so lzma_rs_amd/libmilzma.so > $O/ab_round_dict8m.txt 2>&1; cat $O/ab_round_dict8m.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python - <<PY
import json
l=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print(l["value"], l["ms_per_step"], l["roofline"]["kernel_ms"], l["roofline"].get("traffic"), l["roofline_issue"]["frac"], {k:(v["value"], v["roofline"].get("traffic")) for k,v in l["other_configs"].items()})
PY
