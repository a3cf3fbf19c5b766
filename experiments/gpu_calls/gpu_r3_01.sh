#!/bin/bash
# round 3, call 1: shadow-slot / speculative-walk micro-benchmarks + the headline with 4096 distinct streams
O=gpurun_out/r3_01
mkdir -p $O
timeout 300 ./build/shadow_slots > $O/shadow_slots.txt 2>&1; echo "micro rc=$?"
cat $O/shadow_slots.txt
( time timeout 600 python bench.py --distinct 0 --steps 5 --warmup 1 --no-cpu-baseline ) > $O/bench_distinct0.txt 2>&1; echo "bench rc=$?"
tail -3 $O/bench_distinct0.txt
