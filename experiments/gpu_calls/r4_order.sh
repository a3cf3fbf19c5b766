#!/bin/bash
# Round 4: micro-benchmark of instruction orders inside a form-B decision (experiments/microbench/order_slots.hip, built into build/ before the call).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_order; mkdir -p $O
timeout 300 build/order_slots $1 > $O/order_slots$1.txt 2>&1; echo "rc=$?" >> $O/order_slots$1.txt
cat $O/order_slots$1.txt
