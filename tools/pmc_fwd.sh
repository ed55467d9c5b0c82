#!/bin/bash
# usage (GPU box): tools/pmc_fwd.sh "<counters>" B T L TR path   -> per-dispatch averages for the strip DP kernels
cd /tmp && export TMPDIR=/tmp
CTRS="$1"; shift
rm -rf /tmp/pmcx; rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/pmcx -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/${RUNNER:-run_fwd.py} "$@" > /tmp/o.log 2>&1
python - <<PY
import csv,collections
rows=list(csv.DictReader(open("/tmp/pmcx/p_counter_collection.csv")))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for r in rows:
    k=r["Kernel_Name"][:48]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in agg:
    if ("strip" in k or "grad_links" in k): print(k, len(n[k]), {c: round(v/len(n[k])) for c,v in sorted(agg[k].items())})
PY
