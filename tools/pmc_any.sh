#!/bin/bash
# usage (GPU box): tools/pmc_any.sh "<counters>" "<kernel substring>" <python script + args>
cd /tmp && export TMPDIR=/tmp
CTRS="$1"; PAT="$2"; shift; shift
rm -rf /tmp/pmcy; rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/pmcy -o p --output-format csv -- python "$@" > /tmp/o.log 2>&1
python - <<PY
import csv,collections
rows=list(csv.DictReader(open("/tmp/pmcy/p_counter_collection.csv")))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for r in rows:
    k=r["Kernel_Name"][:64]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in agg:
    if "$PAT" in k: print(k, len(n[k]), {c: round(v/len(n[k])) for c,v in sorted(agg[k].items())})
PY
