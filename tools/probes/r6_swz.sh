# round 6: swizzled LDS layout of the eight-wave sweep against the row-major one - bit identity, parity, timing, LDS counters (one box)
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/${1:-r6w}; mkdir -p $O
bash tools/probes/ab_hash.sh "base swz" 2>&1 | tee $O/hash.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_g1.py -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest.txt
bash tools/probes/ab_ric.sh "base swz" 2>&1 | tee $O/ab.txt
bash tools/probes/ab_ric.sh "base swz" --robot g1 --batch 256 2>&1 | tee -a $O/ab.txt
B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM"
rocprofv3 --kernel-trace --pmc $B --output-format csv -d $O/pmc_sqB -o run -- python bench.py --steps 3 --warmup 1 --settle 0 --cpu-sample 0 > /dev/null 2> $O/pmc.log || tail -5 $O/pmc.log
python - <<PY 2>&1 | tee $O/lds.txt
import csv, glob, collections
f = glob.glob("$O/pmc_sqB/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0]
    if "riccati" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); seen.add((k, r["Dispatch_Id"]))
for k in acc:
    nl = len([1 for kk, d in seen if kk == k])
    c = acc[k]
    print(k, "launches", nl, "conflict", c["SQ_LDS_BANK_CONFLICT"] / nl, "idx_active", c["SQ_LDS_IDX_ACTIVE"] / nl, "share", c["SQ_LDS_BANK_CONFLICT"] / max(1.0, c["SQ_LDS_IDX_ACTIVE"]), "lds insts", c["SQ_INSTS_LDS"] / nl)
PY
