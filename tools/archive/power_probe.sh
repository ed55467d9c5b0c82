# sclk / power while a workload loops: tools/power_probe.sh "<python command>"  (GPU box)
( for i in 1 2 3 4 5 6; do $1 > /dev/null 2>&1; done ) &
BG=$!
sleep 25
for i in 1 2 3 4 5; do /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|power" | tr '\n' ' '; echo; sleep 1.5; done
wait $BG
echo "idle:"; sleep 3; /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|power" | tr '\n' ' '; echo
/opt/rocm/bin/rocm-smi --showmaxpower 2>/dev/null | grep -i -E "max|power" | head -3
