#!/bin/bash
# C5 per-GPU training step under rocprofv3 (GPU box): kernel stats per step + what the host was doing.  usage: tools/train_step_prof.sh [steps] [extra bench args]
cd /tmp && export TMPDIR=/tmp
STEPS=${1:-10}; shift
rm -rf /tmp/tr; rocprofv3 --kernel-trace --stats -d /tmp/tr -o k --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload train --steps $STEPS --warmup 4 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench line: ms_per_step', round(d['ms_per_step'],2), 'value', round(d['value'],1), d['unit'], '| peak GB', d.get('config',{}).get('peak_memory_GB', d.get('peak_memory_GB')))"
python - <<PY
import csv, collections
rows = list(csv.DictReader(open("/tmp/tr/k_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nst = $STEPS
# steady state = the last nst optimizer steps: delimit by the fused Adam kernel (one multi-tensor launch group per step)
ends = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r["Kernel_Name"] and "adam" in r["Kernel_Name"].lower()]
# group consecutive adam launches
groups = []
for i in ends:
    if groups and i - groups[-1][-1] < 40: groups[-1].append(i)
    else: groups.append([i])
cut = groups[-nst - 1][-1] + 1 if len(groups) > nst else 0
ss = rows[cut:groups[-1][-1] + 1]
t0, t1 = int(ss[0]["Start_Timestamp"]), int(ss[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ss)
print(f"steady state: {nst} steps, {len(ss) / nst:.0f} kernels per step, wall {(t1 - t0) / nst / 1e6:.2f} ms per step, GPU busy (sum of kernel durations, one stream) {busy / nst / 1e6:.2f} ms per step = {100 * busy / (t1 - t0):.0f} % of the wall")
gaps = [int(ss[i + 1]["Start_Timestamp"]) - int(ss[i]["End_Timestamp"]) for i in range(len(ss) - 1)]
big = sum(g for g in gaps if g > 20000)
print(f"idle gaps between kernels: total {sum(max(g, 0) for g in gaps) / nst / 1e6:.2f} ms per step, of which gaps > 20 us: {big / nst / 1e6:.2f} ms ({sum(1 for g in gaps if g > 20000) / nst:.0f} per step)")
agg = collections.defaultdict(lambda: [0, 0])
for r in ss:
    n = r["Kernel_Name"]
    n = n[:110]
    agg[n][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); agg[n][1] += 1
for n, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:28]:
    print(f"  {d / nst / 1e6:7.3f} ms  {c / nst:7.1f} x  {100 * d / busy:5.1f} %  {n}")
PY
