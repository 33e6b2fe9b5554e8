# round 6: ring roll-out of the eight-wave sweep against the head of the round (tools/probes/lib_head.bin = the library of commit 1ed5d28; others: variants of the working tree)
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/${1:-r6ring}; mkdir -p $O
V=${2:-head ring}
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee $O/tests.txt
bash tools/probes/prof_ric2.sh "ringp" 2>&1 | grep ricp | tee $O/prof.txt
{ bash tools/probes/ab_ric.sh "$V"
  bash tools/probes/ab_ric.sh "$V" --batch 64
  bash tools/probes/ab_ric.sh "$V" --batch 128
  bash tools/probes/ab_ric.sh "$V" --robot g1 --batch 256; } 2>&1 | tee $O/ab.txt
