# per-kernel durations of the matrix-core extract_links kernels: bash tools/xl_mfma_prof.sh L TR
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/xp; rocprofv3 --kernel-trace -d /tmp/xp -o k --output-format csv -- python $GRAFT_REPO_ROOT/tools/xl_mfma_prof.py $1 $2 > /tmp/xp.log 2>&1
tail -3 /tmp/xp.log
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open("/tmp/xp/k_kernel_trace.csv")))
seq = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if "xl_mfma" not in n and "extract_links" not in n:
        continue
    seq.setdefault(n[:70], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in seq.items():
    v = sorted(v)
    print(f"{k:70s} calls {len(v):4d}  median {v[len(v) // 2]:9.1f} us  min {v[0]:9.1f}")
PY
