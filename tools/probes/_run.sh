export TMPDIR=/tmp PYTHONPATH=.
for a in "--no-profile" "" "--warmup 20" "--no-profile --warmup 20"; do python bench.py --cpu-sample 0 --no-fused $a 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'], d['timing_spread']['ms_per_step_min'], d['timing_spread']['ms_per_step_median'])"; done
