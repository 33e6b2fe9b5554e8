# round 6: the small transfers of an MPC tick batched (lib_new.bin) against one hipMemcpyAsync per array (lib_old.bin)
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r6tick; mkdir -p $O
cp bipedal_control_amd/libbpmpc.so /tmp/keep.so
for rep in 1 2; do for v in old new; do cp tools/probes/lib_$v.bin bipedal_control_amd/libbpmpc.so; echo "== $v"; python tools/probes/tick_probe.py 2>&1 | grep -v amdgpu; python tools/latency_probe.py 2>&1 | tail -2; done; done 2>&1 | tee $O/ab.txt
cp /tmp/keep.so bipedal_control_amd/libbpmpc.so
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee $O/tests.txt
