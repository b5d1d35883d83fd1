cd /tmp; export TMPDIR=/tmp
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_STALL_MULTI_MISS_sum" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $set -d /root/repo/gpurun_out/pmc_dg$i -o p --output-format csv -- python /root/repo/tools/dgemm_only.py > /root/repo/gpurun_out/pmc_dg$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for i in (1,2,3):
    fs = glob.glob(f"/root/repo/gpurun_out/pmc_dg{i}/**/*counter_collection.csv", recursive=True)
    if not fs: print("no file", i); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "dgemm_kernel<192" not in k or ", true, 0" not in k.replace("true, true","X, true") and False: pass
        if "dgemm_kernel" not in k: continue
        key = k[k.index("dgemm_kernel"):k.index(">")+1]
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); 
    for key, v in acc.items():
        print(i, key, {a: "%.4g" % b for a, b in v.items()})
PY
