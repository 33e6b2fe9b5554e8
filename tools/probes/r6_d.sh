export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r6e; mkdir -p $O
cp bipedal_control_amd/libbpmpc.so /tmp/keep2.so; cp tools/probes/lib_img_timeline.bin bipedal_control_amd/libbpmpc.so
python tools/lin_timeline.py 2>&1 | tail -25 | tee $O/timeline.txt
cp /tmp/keep2.so bipedal_control_amd/libbpmpc.so
bash tools/probes/prof_lin.sh "img_linprof" 2>&1 | tee $O/prof_lin.txt
bash tools/probes/ab_lin.sh "aux img aux img" r6e "--batch 256|--batch 4096|--robot g1 --batch 1024"
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
