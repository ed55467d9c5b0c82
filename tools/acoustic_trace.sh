# per (kernel, grid) aggregation of the acoustic stage's dispatches (rocprofv3 --kernel-trace; tools/agg_trace.py)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/act; rocprofv3 --kernel-trace -d /tmp/act -o k --output-format csv -- python $GRAFT_REPO_ROOT/tools/acoustic_stage_prof.py 10 > /tmp/act.log 2>&1
grep -E "acoustic stage|graph lengths" /tmp/act.log
python $GRAFT_REPO_ROOT/tools/agg_trace.py /tmp/act/k_kernel_trace.csv | head -${ROWS:-45}
