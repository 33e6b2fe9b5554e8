bash tools/probes/prof_ric2.sh "1" 2>&1 | grep ricp
bash tools/probes/ab_step.sh "ricm3 ricm5" abr8 "--batch 256" 2>&1 | tail -4
bash tools/probes/test_ab.sh "ricm5" "tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_hard_cone.py -m gpu" 2>&1 | tail -3
