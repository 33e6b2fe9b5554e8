# round 6: the last wide round of the line search judged by k_ls_tail (lib_lsfold.bin) against the tree before it (lib_base.bin)
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/${1:-r6ls}; mkdir -p $O
V=${2:-base lsfold}
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee $O/tests.txt
{ bash tools/probes/ab_ric.sh "$V"
  bash tools/probes/ab_ric.sh "$V" --gait-start -1.225
  bash tools/probes/ab_ric.sh "$V" --batch 4096
  bash tools/probes/ab_ric.sh "$V" --robot g1 --batch 1024; } 2>&1 | tee $O/ab.txt
