export TMPDIR=/tmp PYTHONPATH=.
bash tools/probes/prof_lin.sh "linprof linprof128" 2>&1 | grep -v amdgpu
