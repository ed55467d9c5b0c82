#!/bin/bash
# vocoder group size sweep of the S2ST pipeline (utterances per vocoder call)
for g in 8 16 32; do echo -n "group $g: "; timeout 300 python bench.py --workload s2st --steps 20 --warmup 5 --vocoder-group $g 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d['value'],1), 'utt/s', round(d['ms_per_step'],2), 'ms')"; done
