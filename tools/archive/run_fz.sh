mkdir -p gpurun_out/fz; rm -f gpurun_out/fz/replay.txt
for k5 in 0 1; do
  echo "== k5_path $k5" >> gpurun_out/fz/replay.txt
  DSP_FUZZ_ONLY=216 DSP_FUZZ_K5=$k5 timeout 900 python tools/fuzz_dag.py 500 22 narrow weak 2>&1 | grep -v amdgpu | cut -c1-600 >> gpurun_out/fz/replay.txt
done
