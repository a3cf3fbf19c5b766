#!/bin/bash
# Round 4: the one PMC pass r4_final2.sh did not get (configs[3], FETCH_SIZE: rocprofv3 hung three times there); merged with that call's other passes
# by tools/make_pmc_profile.py afterwards.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4_ev2; rm -rf $O; mkdir -p $O
for attempt in 1 2; do
  rm -rf $O/xz/pass_1
  timeout 100 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/xz/pass_1 -- python bench.py --config xz --steps 1 --warmup 0 --no-cpu-baseline --no-verify --other-configs none > $O/xz_pass_1.log 2>&1
  rc=$?
  n=$(find $O/xz/pass_1 -name "*counter_collection.csv" 2>/dev/null | wc -l)
  echo "xz pass 1 attempt $attempt rc=$rc csv=$n"
  [ "$n" -gt 0 ] && break
done
rm -rf $O/*/pass_*/*/*.db 2>/dev/null
find $O -name "*.csv" -size +2M -delete
