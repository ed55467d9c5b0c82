cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ac; rocprofv3 --kernel-trace --stats -d /tmp/ac -o k --output-format csv -- python $GRAFT_REPO_ROOT/tools/acoustic_stage_prof.py 20 2>&1 | grep acoustic
python - <<'PY'
import csv
rows = list(csv.DictReader(open("/tmp/ac/k_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
nb = 24
print(f"kernels per batch {calls / nb:.0f}; GPU ms per batch {tot / nb / 1e6:.2f} (24 batches incl. warm-up + 1 probe; model build excluded by name below is not possible: a few init kernels are included)")
import os
nrows, namew = int(os.environ.get("ROWS", "14")), int(os.environ.get("NAMEW", "100"))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:nrows]:
    print(f"  {r['Name'][:namew]:100s} calls/batch {int(r['Calls']) / nb:6.1f}  ms/batch {float(r['TotalDurationNs']) / nb / 1e6:6.3f}  {float(r['Percentage']):5.1f} %")
PY
