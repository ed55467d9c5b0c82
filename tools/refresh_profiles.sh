#!/bin/bash
# usage (GPU box, repo root): tools/refresh_profiles.sh <tag>   -> gpurun_out/<tag>/{bench_dag.json, dag_kernel_stats.csv, pmc_fetch.txt, pmc_write.txt}
TAG=${1:-r01d}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/bench_dag.json 2> $OUT/bench_dag.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks; rocprofv3 --kernel-trace --stats -d /tmp/ks -o k --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>&1
cp /tmp/ks/k_kernel_stats.csv $OUT/dag_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc $C -d /tmp/pm -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2>&1
  python - > $OUT/pmc_$C.txt <<PY
import csv,collections
rows=list(csv.DictReader(open("/tmp/pm/p_counter_collection.csv")))
agg=collections.defaultdict(float); n=collections.defaultdict(set)
for r in rows:
    if "dsp::" not in r["Kernel_Name"]: continue
    k=r["Kernel_Name"][:70]
    agg[k]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in agg: print(k, len(n[k]), "$C per dispatch (raw KB):", round(agg[k]/len(n[k]),1))
PY
done
