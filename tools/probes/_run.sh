export TMPDIR=/tmp PYTHONPATH=.
bash tools/probes/ab_step.sh "ricm3 ricm9" abr12 "--batch 256|--robot g1 --batch 256" 2>&1 | tail -8
cp bipedal_control_amd/libbpmpc.so /tmp/keep2.so
cp tools/probes/lib_ricm9.bin bipedal_control_amd/libbpmpc.so
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
cp tools/probes/lib_wavep.bin bipedal_control_amd/libbpmpc.so
python tools/riccati_wave_phase_profile.py 4096 h1 4 2>&1 | tail -2
python tools/riccati_wave_phase_profile.py 1024 g1 2 2>&1 | tail -2
cp /tmp/keep2.so bipedal_control_amd/libbpmpc.so
