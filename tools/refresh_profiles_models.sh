#!/bin/bash
# usage (GPU box, repo root): tools/refresh_profiles_models.sh <tag> -> gpurun_out/<tag>/bench_{s2tt,s2tt_jointviterbi,s2st,s2st_bf16,train,train_bf16}.json, hifigan_kernel_stats.csv
TAG=${1:-r01f}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py --workload s2tt > $OUT/bench_s2tt.json 2> $OUT/err.txt
python bench.py --workload s2tt --decode-strategy jointviterbi > $OUT/bench_s2tt_jointviterbi.json 2>> $OUT/err.txt
python bench.py --workload s2st > $OUT/bench_s2st.json 2>> $OUT/err.txt
python bench.py --workload s2st --amp bf16 > $OUT/bench_s2st_bf16.json 2>> $OUT/err.txt
python bench.py --workload train > $OUT/bench_train.json 2>> $OUT/err.txt
python bench.py --workload train --amp bf16 > $OUT/bench_train_bf16.json 2>> $OUT/err.txt
python tools/hifigan_bench.py 32 330 > $OUT/hifigan_bench.txt 2>> $OUT/err.txt
python tools/hifigan_bench.py 8 330 >> $OUT/hifigan_bench.txt 2>> $OUT/err.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/hk; rocprofv3 --kernel-trace --stats -d /tmp/hk -o k --output-format csv -- python $GRAFT_REPO_ROOT/tools/hifigan_bench.py 32 330 > /dev/null 2>&1
cp /tmp/hk/k_kernel_stats.csv $OUT/hifigan_kernel_stats.csv
