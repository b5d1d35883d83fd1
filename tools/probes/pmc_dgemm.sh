# PMC passes over the two sketch GEMMs on their own (tools/dgemm_only.py); prints per-kernel sums over 2 launches each
cd /tmp; export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $set -d /root/repo/gpurun_out/pmc_dg$i -o p --output-format csv -- python /root/repo/tools/dgemm_only.py > /root/repo/gpurun_out/pmc_dg$i.log 2>&1 || tail -3 /root/repo/gpurun_out/pmc_dg$i.log
done
python - $i <<'PY'
import csv, glob, collections, sys
for i in range(1, int(sys.argv[1]) + 1):
    fs = glob.glob(f"/root/repo/gpurun_out/pmc_dg{i}/**/*counter_collection.csv", recursive=True)
    if not fs: print("no file", i); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "dgemm_kernel" not in k: continue
        key = k[k.index("dgemm_kernel"):k.index(">")+1]
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
    for key, v in acc.items():
        print(i, key, {a: "%.5g" % b for a, b in v.items()})
PY
