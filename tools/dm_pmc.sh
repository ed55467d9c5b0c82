#!/bin/bash
# PMC passes over the dense MFMA DP kernel at a throughput-bound shape (no wavefront waiting): where do the wave cycles go?
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tools/dm_dbg.py 16 128 4096"
pmc() {
  rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc $1 -d /tmp/pm -o p --output-format csv -- $CMD > /dev/null 2>&1
  python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open("/tmp/pm/p_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"][:60]
    if "dense_mfma" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in sorted(agg): print(k, "| dispatches", len(n[k]), "| per dispatch:", {c: round(v / len(n[k]), 1) for c, v in sorted(agg[k].items())})
PY
}
pmc "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM"
pmc "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
pmc "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT"
pmc "SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM"
