#!/bin/bash
# Round 6: the direct-bit chains on the vector ALU (MILZMA_GEN_VDIRECT; VDIRECT_S = the last so many bits of a chain stay scalar), alternating A/B
# on the bench batch; push mode with the views gathered and uploaded in four pipelined pieces
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6_call15; rm -rf $O; mkdir -p $O
L=lzma_rs_amd/libmilzma.so
V=""
for v in vd0 vd3 vd4 vd5 vd6; do V="$V lzma_rs_amd/variants/libmilzma_$v.so"; done
timeout 1500 python experiments/ab_bench.py --steps 4 $L $V $L $V | tee $O/ab_text.txt
timeout 600 python experiments/ab_bench.py --steps 3 --dict 8388608 $L $V | tee $O/ab_dict8m.txt
timeout 600 python experiments/streams_bench.py 2>$O/streams_bench.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print({k: d[k] for k in ('GBps', 'seconds', 'writes_s', 'finish_s', 'bad')})" | tee $O/streams_bench.txt
timeout 900 python -m pytest tests/test_gpu_streams.py tests/test_gpu_reader.py tests/test_gpu_feed.py -q -x 2>&1 | tail -3 | tee $O/streams_tests.txt
