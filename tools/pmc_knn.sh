cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVES"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $set -d /root/repo/gpurun_out/pmc_knn$i -o p --output-format csv -- python /root/repo/tools/knn_only.py > /root/repo/gpurun_out/pmc_knn$i.log 2>&1
  python - $i <<'PY'
import csv, glob, collections, sys
i = sys.argv[1]
fs = glob.glob(f"/root/repo/gpurun_out/pmc_knn{i}/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(float)
for r in csv.DictReader(open(fs[0])):
    if "knn_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]] += float(r["Counter_Value"]) / 2
print({a: "%.4g" % b for a, b in acc.items()})
PY
done
