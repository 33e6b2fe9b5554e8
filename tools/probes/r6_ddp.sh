# round 6: the DDP solver on the fast kernels - tests, bench line, kernel table
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/${1:-r6d}; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ddp.py -m gpu -q -x 2>&1 | tail -15 | tee $O/pytest_ddp.txt
timeout 600 python bench.py --solver ddp --cpu-sample 2 --profile-all --no-fused > $O/bench_ddp.json 2> $O/bench_ddp.err; tail -3 $O/bench_ddp.err
tail -1 $O/bench_ddp.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['metric'], d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['config'].get('step_lengths'), d['config'].get('roll_out_points_mean'), d.get('cpu_baseline'), d['roofline'].get('write_roof'))"
timeout 600 python bench.py --solver ddp --cpu-sample 0 > $O/bench_ddp_default.json 2>> $O/bench_ddp.err
tail -1 $O/bench_ddp_default.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['metric'], d['value'], d['ms_per_step'], (d.get('fused') or {}).get('value'), d['kernel_ms_per_step'])"
timeout 300 python bench.py --cpu-sample 0 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['metric'], d['value'], d['ms_per_step'], (d.get('fused') or {}).get('value'), d['kernel_ms_per_step'], d['roofline']['frac'], d['roofline'].get('write_roof'))"
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest.txt
