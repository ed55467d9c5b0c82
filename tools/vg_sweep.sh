for g in 8 16 32; do python bench.py --workload s2st --steps 10 --warmup 4 --vocoder-group $g 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('group', d['config']['workload'].split('groups of ')[1][:3], d['value'], d['ms_per_step'], d['roofline']['frac'] if d['roofline'] else None)"; done
