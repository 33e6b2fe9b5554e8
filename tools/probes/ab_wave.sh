# A/B of the sweep variants inside one gpurun call: edit the list of shapes / BPMPC_RICCATI_WAVE values, then
#   gpurun --timeout 1500 -- 'bash tools/probes/ab_wave.sh'
# (0 workgroup kernels only, 1 default choice by batch, 2 riccati_wave.h, 4 riccati_wave2.h, 5 riccati_pair.h at every batch size)
for args in "--batch 512" "--batch 1024" "--robot g1 --batch 1024" "--batch 2048" "--batch 4096"; do
  for w in 1 5; do echo -n "[$args] wave=$w "; BPMPC_RICCATI_WAVE=$w timeout 400 python bench.py $args --profile-all --steps 10 --warmup 2 --cpu-sample 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"ms_per_step\"], d[\"kernel_ms_per_step\"][\"riccati\"], (d.get(\"fused\") or {}).get(\"value\"))"; done; done
