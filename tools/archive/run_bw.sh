mkdir -p gpurun_out/bw
rm -f gpurun_out/bw/bench2.txt
for sh in "32 512 4096 32" "32 64 4096 32" "32 16 4096 32"; do
  echo "== bwd_bench $sh" >> gpurun_out/bw/bench2.txt
  timeout 300 python tools/bwd_bench.py $sh >> gpurun_out/bw/bench2.txt 2>&1
done
for sh in "32 512 4096 64" "32 64 4096 128" "32 128 4096 128" "16 300 2048 48"; do
  echo "== bwd_wide_bench $sh" >> gpurun_out/bw/bench2.txt
  timeout 300 python tools/bwd_wide_bench.py $sh >> gpurun_out/bw/bench2.txt 2>&1
done
timeout 900 python -m pytest tests/test_gpu_dag_ops.py -x -q -k "backward or fused or grad or pitched or planes or blocks_of_32 or weak" 2>&1 | tail -3 >> gpurun_out/bw/bench2.txt
