# per-kernel durations of tools/attention_bench.py (kernel time without the host-side call overhead)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/atp; rocprofv3 --kernel-trace -d /tmp/atp -o k --output-format csv -- python $GRAFT_REPO_ROOT/tools/attention_bench.py > /tmp/atp.log 2>&1
tail -8 /tmp/atp.log
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open("/tmp/atp/k_kernel_trace.csv")))
seq = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if "attention" not in n and "attn" not in n:
        continue
    key = (n[:60], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
    seq.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in seq.items():
    v = sorted(v)
    print(f"{k[0]:60s} grid {k[1]:>8s} wg {k[2]:>4s} calls {len(v):4d}  median {v[len(v) // 2]:8.1f} us  min {v[0]:8.1f}")
PY
