export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/${1:-r6x}; mkdir -p $O
V=${2:-base acl12}
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee $O/tests.txt
{ bash tools/probes/ab_ric.sh "$V"
  bash tools/probes/ab_ric.sh "$V" --batch 128
  bash tools/probes/ab_ric.sh "$V" --robot g1 --batch 256; } 2>&1 | tee $O/ab.txt
