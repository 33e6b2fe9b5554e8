export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/${1:-r6r}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_ddp.py -m gpu -q -x 2>&1 | tail -3
timeout 600 python bench.py --solver ddp --cpu-sample 0 --profile-all --no-fused 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['metric'], d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['config'].get('step_lengths'))"
timeout 300 python tools/closed_loop_soak.py 300 2>&1 | tail -1
