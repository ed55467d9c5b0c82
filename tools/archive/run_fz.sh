mkdir -p gpurun_out/fz
for args in "600 11 mid" "600 12 mid weak" "600 13 mid2" "600 14 mid2 weak"; do
  echo "== DSP_FUZZ_K5=3 fuzz_dag.py $args" >> gpurun_out/fz/fuzz.txt
  DSP_FUZZ_K5=3 timeout 900 python tools/fuzz_dag.py $args 2>&1 | tail -4 >> gpurun_out/fz/fuzz.txt
done
echo "== fuzz_dag.py 400 15 (narrow windows, auto)" >> gpurun_out/fz/fuzz.txt
timeout 900 python tools/fuzz_dag.py 400 15 2>&1 | tail -3 >> gpurun_out/fz/fuzz.txt
