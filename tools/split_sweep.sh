cd /tmp; export TMPDIR=/tmp
for sp in 1 2 3 6 12; do
  export HSSK_DGEMM_SPLIT=$sp
  timeout 120 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_DRAM_32B_sum -d /root/repo/gpurun_out/sweep_$sp -o p --output-format csv -- python /root/repo/tools/dgemm_only.py > /root/repo/gpurun_out/sweep_$sp.log 2>&1
  python - $sp <<'PY'
import csv, glob, collections, sys
sp = sys.argv[1]
fs = glob.glob(f"/root/repo/gpurun_out/sweep_{sp}/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"]
    if "dgemm_kernel" not in k: continue
    key = k[k.index("dgemm_kernel"):k.index(">")+1]
    acc[key].append(float(r["Counter_Value"]) * 32 / 1e9)
ts = collections.defaultdict(list)
fs = glob.glob(f"/root/repo/gpurun_out/sweep_{sp}/**/*kernel_trace.csv", recursive=True)
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"]
    if "dgemm" not in k: continue
    key = k[k.index("dgemm"):k.index(">")+1] if ">" in k else "dgemm_reduce"
    ts[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for key in sorted(ts):
    print("split", sp, key, "ms", ["%.2f" % t for t in ts[key]], "GB", ["%.1f" % g for g in acc.get(key, [])])
PY
done
