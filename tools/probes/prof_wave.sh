cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for w in 0 1; do echo "wave=$w"; BPMPC_RICCATI_WAVE=$w timeout 300 python bench.py --batch 4096 --profile-all --steps 10 --warmup 2 --cpu-sample 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"ms_per_step\"], d[\"kernel_ms_per_step\"], (d.get(\"fused\") or {}).get(\"ms_per_step\"))"; done
BPMPC_RICCATI_WAVE=1 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_wave -o w -- python bench.py --batch 4096 --steps 5 --warmup 1 --cpu-sample 0 --no-fused > /dev/null 2>&1
python - <<'P'
import csv,glob
f=glob.glob("gpurun_out/prof_wave/**/*kernel_stats.csv",recursive=True)
for r in list(csv.DictReader(open(f[0])))[:8]: print(r["Name"][:60], r["Calls"], r["AverageNs"])
P
