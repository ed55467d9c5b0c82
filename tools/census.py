import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.dp_microbench import inputs
from daspeech_amd import custom_ops as ops, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m, k, ol, tl = inputs(B, 512, 4096, 32)
_lib.set_option("dp_path", 4)
mg = m.clone().requires_grad_()
ops.dag_loss(mg, k, ol, tl); torch.cuda.synchronize()
rows = [l.split() for l in open("/tmp/census.txt")]
cus = collections.defaultdict(list)
for r in rows:
    hw = int(r[1], 16); xcc = hw >> 32; hwid = hw & 0xffffffff
    cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
    cus[(xcc, se, sh, cu)].append((int(r[3]), int(r[4]), int(r[0])))
print("WGs", len(rows), "distinct CUs", len(cus))
t0 = min(v[0] for vs in cus.values() for v in vs)
ov = 0
for key, vs in list(cus.items())[:6]:
    print(key, [(a - t0, e - t0, tk) for a, e, tk in sorted(vs)])
for vs in cus.values():
    vs = sorted(vs)
    for i in range(len(vs) - 1):
        if vs[i + 1][0] < vs[i][1]: ov += 1
print("overlapping consecutive pairs on a CU:", ov)
print("max start offset", max(v[0] for vs in cus.values() for v in vs) - t0, "max end", max(v[1] for vs in cus.values() for v in vs) - t0)
