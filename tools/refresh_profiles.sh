#!/bin/bash
# usage (GPU box, repo root): tools/refresh_profiles.sh <tag>  -> gpurun_out/<tag>/...: the driver-shaped bench line, rocprofv3 kernel
# stats of the DAG workload, FETCH / WRITE / SQ counter passes (separate --pmc runs, --kernel-trace only) for the DP and HiFi-GAN kernels
TAG=${1:-r03}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > $OUT/bench_headline.json 2> $OUT/bench_headline.err
cd /tmp && export TMPDIR=/tmp
DAGCMD="python $GRAFT_REPO_ROOT/bench.py --workload dag --no-cpu-baseline --no-c1 --no-peaked --steps 10 --warmup 2"
rm -rf /tmp/ks; rocprofv3 --kernel-trace --stats -d /tmp/ks -o k --output-format csv -- $DAGCMD > /dev/null 2>&1
cp /tmp/ks/k_kernel_stats.csv $OUT/dag_kernel_stats.csv
pmc() {   # $1 = counters, $2 = kernel-name substring filter, $3.. = command
  local C="$1" PAT="$2"; shift; shift
  rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc $C -d /tmp/pm -o p --output-format csv -- "$@" > /dev/null 2>&1
  python - "$PAT" <<'PY'
import csv, collections, sys
pat = sys.argv[1].split("|")
rows = list(csv.DictReader(open("/tmp/pm/p_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"][:90]
    if not any(p in k for p in pat): continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in sorted(agg): print(k, "| dispatches", len(n[k]), "| per dispatch:", {c: round(v / len(n[k]), 1) for c, v in sorted(agg[k].items())})
PY
}
{
echo "# rocprofv3 --pmc passes (separate runs, --kernel-trace only) over: $DAGCMD"
echo "## FETCH_SIZE (raw KB; double it for 16-byte-per-lane streams, MI355X_MICROARCH.md HBM section)"; pmc FETCH_SIZE "dsp::" $DAGCMD
echo "## WRITE_SIZE (KB)"; pmc WRITE_SIZE "dsp::" $DAGCMD
echo "## SQ issue / wait counters of the DP kernels (quad-cycle units)"
pmc "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "strip|maxstrip|grad_links|backtrace" $DAGCMD
pmc "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "strip|maxstrip|grad_links|backtrace" $DAGCMD
pmc "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "strip|maxstrip|grad_links|backtrace" $DAGCMD
} > $OUT/pmc_dag.txt 2>&1
# the bench line's roofline.traffic comes from THIS pass: per-dispatch FETCH_SIZE (doubled: 16-byte-per-lane streams on gfx950) + WRITE_SIZE of the DP forward
python - "$OUT/pmc_dag.txt" "$TAG" <<'PY'
import json, re, sys, os
txt = open(sys.argv[1]).read(); tag = sys.argv[2]
def grab(counter):
    m = re.search(r"dag_strip4g_kernel[^\n]*per dispatch: \{[^}]*'%s': ([0-9.]+)" % counter, txt)
    return float(m.group(1)) if m else None
f, w = grab("FETCH_SIZE"), grab("WRITE_SIZE")
path = os.path.join(os.environ["GRAFT_REPO_ROOT"], "profiles", "pmc_dag_fwd.json")
rec = json.load(open(path))
if f and w:
    rec["tr32"] = {"FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB": w, "hbm_bytes_per_launch": int(round((2 * f + w) * 1024)), "shape": [32, 512, 4096, 32],
                   "source": "profiles/%s_pmc_dag.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, this round)" % tag}
    rec["_source"] = "tr32: refreshed by tools/refresh_profiles.sh %s; FETCH_SIZE doubled per MI355X_MICROARCH.md (HBM section); older rounds' records kept below" % tag
    json.dump(rec, open(os.path.join(os.path.dirname(sys.argv[1]), "pmc_dag_fwd.json"), "w"), indent=1)
    print("pmc_dag_fwd.json: FETCH", f, "WRITE", w)
PY
HGCMD="python $GRAFT_REPO_ROOT/tools/hifigan_f32_prof.py 32 329 3"      # the bench's vocoder call: fp32-accurate, fused ResBlock units, whole batch
rm -rf /tmp/kh; rocprofv3 --kernel-trace --stats -d /tmp/kh -o k --output-format csv -- $HGCMD > $OUT/hifigan_bench.txt 2>&1
cp /tmp/kh/k_kernel_stats.csv $OUT/hifigan_kernel_stats.csv
{
echo "# rocprofv3 --pmc passes over: $HGCMD"
echo "## MFMA busy / LDS (hifigan_resunit_f32_kernel = fused ResBlock units, hifigan_conv_f32_kernel = conv_pre and the upsamplers)"; pmc "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "hifigan" $HGCMD
pmc "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "hifigan" $HGCMD
echo "## FETCH_SIZE (raw KB)"; pmc FETCH_SIZE "hifigan" $HGCMD
echo "## WRITE_SIZE (KB)"; pmc WRITE_SIZE "hifigan" $HGCMD
} > $OUT/pmc_hifigan.txt 2>&1

# the other workloads of the bench (one line each), the training-step and acoustic-stage kernel breakdowns
cd $GRAFT_REPO_ROOT
{ python bench.py --workload train --steps 10 --warmup 4 2>/dev/null | tail -1; python bench.py --workload s2tt --steps 10 --warmup 4 2>/dev/null | tail -1; } > $OUT/bench_train_s2tt.txt
bash tools/train_step_prof.sh 10 > $OUT/train_step_kernels.txt 2>&1
bash tools/acoustic_stage_prof.sh > $OUT/acoustic_stage_kernels.txt 2>&1
ROWS=40 NAMEW=110 bash tools/acoustic_stage_prof.sh > $OUT/acoustic_stage_kernels.txt 2>&1
ROWS=40 bash tools/acoustic_trace.sh > $OUT/acoustic_stage_by_grid.txt 2>&1
bash tools/attention_prof.sh > $OUT/attention_kernels.txt 2>&1
cd /tmp
ACCMD="python $GRAFT_REPO_ROOT/tools/acoustic_stage_prof.py 5"
{
echo "# rocprofv3 --pmc over: $ACCMD  (matrix-core kernels of the acoustic stage; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES))"
pmc "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "conv1d_split_kernel|ffn_split|attention_split" $ACCMD
pmc "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "conv1d_split_kernel|ffn_split|attention_split" $ACCMD
} > $OUT/pmc_acoustic.txt 2>&1
