#!/bin/bash
# usage (GPU box, repo root): tools/pmc_dense.sh <outfile> — counter passes (separate --pmc runs, --kernel-trace only) over the dense-window DAG
# workload (C2 with TR = L-1): what is contended while the block products run (verdict r05 item 5: L2 weight-fragment stream vs LDS vs readiness spins)
OUT=$GRAFT_REPO_ROOT/${1:-gpurun_out/pmc_dense.txt}
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --workload dag --tr 4095 --no-cpu-baseline --no-c1 --no-peaked --no-live-traffic --steps 3 --warmup 1"
pmc() {
  local C="$1" PAT="$2"; shift; shift
  rm -rf /tmp/pm; timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/pm -o p --output-format csv -- "$@" > /tmp/pm.log 2>&1
  python - "$PAT" <<'PY'
import csv, collections, sys, os
pat = sys.argv[1].split("|")
f = "/tmp/pm/p_counter_collection.csv"
if not os.path.exists(f):
    print("  (no counter file: " + open("/tmp/pm.log").read()[-300:].replace("\n", " | ") + ")"); sys.exit(0)
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"][:70]
    if not any(p in k for p in pat): continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in sorted(agg): print(" ", k, "| dispatches", len(n[k]), "| per dispatch:", {c: round(v / len(n[k]), 1) for c, v in sorted(agg[k].items())})
PY
}
{
echo "# rocprofv3 --pmc passes over: $CMD"
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z_]*\(HIT\|MISS\|REQ\|READ\|EA_RDREQ\)[A-Za-z_\[\]0-9]*" | sort -u | head -30 | tr '\n' ' '; echo
for C in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_INSTS_SMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  echo "## $C"; pmc "$C" "dense" $CMD
done
} > $OUT 2>&1
