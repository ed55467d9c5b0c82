"""Aggregate a rocprofv3 kernel trace CSV by (kernel name, grid): python tools/agg_trace.py trace.csv [skip_first_n_per_key]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void dsp::", "")[:60]
    key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("LDS_Block_Size", ""))
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg.setdefault(key, []).append(d)
tot = sum(sum(v) for v in agg.values())
print(f"total kernel time {tot/1e6:.3f} ms over {len(rows)} dispatches")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{sum(v)/1e6:9.3f} ms  {len(v):5d} x {sum(v)/len(v)/1e3:9.1f} us  {100*sum(v)/tot:5.1f}%  {k}")
