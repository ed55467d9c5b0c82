set -x
mkdir -p gpurun_out/bw
timeout 1500 python -m pytest tests/test_gpu_dag_ops.py -x -q 2>&1 | tail -15 > gpurun_out/bw/tests.txt
DSP_PITCH_FILL=nan timeout 900 python -m pytest tests/test_gpu_dag_ops.py -x -q -k "pitched or planes or blocks_of_32 or windows_33" 2>&1 | tail -15 > gpurun_out/bw/tests_nanfill.txt
rm -f gpurun_out/bw/bench.txt
for sh in "32 512 4096 64" "32 512 4094 64" "32 64 4096 128" "32 128 4095 128" "16 300 2046 48"; do
  echo "== $sh" >> gpurun_out/bw/bench.txt
  timeout 300 python tools/bwd_wide_bench.py $sh --oracle >> gpurun_out/bw/bench.txt 2>&1
done
