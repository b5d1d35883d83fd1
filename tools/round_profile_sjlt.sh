#!/bin/bash
# SJLT-sketch workload (bench.py --sketch sjlt): bench line, rocprofv3 kernel summary, HBM read traffic of the two sketch
# kernels (TCC_EA0_RDREQ_DRAM_32B counts 32-byte DRAM reads exactly; FETCH_SIZE for comparison), separate PMC passes.
# usage (on the GPU box, from the repo root): bash tools/round_profile_sjlt.sh <tag>
tag=${1:-rXX}
out=/root/repo/gpurun_out/$tag
mkdir -p $out
cd /root/repo
python bench.py --sketch sjlt --no-cpu-baseline > $out/bench_sjlt_n1.json 2> $out/bench_sjlt_n1.err
cd /tmp; export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats -d $out/kt_sjlt -o kt --output-format csv -- python /root/repo/bench.py --sketch sjlt --no-cpu-baseline > $out/bench_sjlt_under_rocprof.json 2> $out/kt_sjlt.err
for c in TCC_EA0_RDREQ_DRAM_32B_sum FETCH_SIZE; do
  timeout 120 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_sjlt_$c -o p --output-format csv -- python /root/repo/bench.py --sketch sjlt --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_sjlt_$c.log 2>&1
done
python - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {}
for c in ("TCC_EA0_RDREQ_DRAM_32B_sum", "FETCH_SIZE"):
    fs = glob.glob(out + "/pmc_sjlt_%s/**/*counter_collection.csv" % c, recursive=True)
    if not fs:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        for name in ("sjlt_n_kernel", "sjlt_t_kernel"):
            if name in k:
                acc[name].append(float(r["Counter_Value"]))
    res[c + "_per_launch"] = {k: sum(v) / len(v) for k, v in acc.items()}
d = res.get("TCC_EA0_RDREQ_DRAM_32B_sum_per_launch", {})
if d:
    res["read_bytes_per_launch_RDREQ_DRAM_32B_x32"] = {k: v * 32 for k, v in d.items()}
    res["hbm_read_bytes_per_launch"] = sum(v * 32 for v in d.values()) / len(d)
f = res.get("FETCH_SIZE_per_launch", {})
if f:
    res["read_bytes_per_launch_FETCH_SIZE_KB_x1024"] = {k: v * 1024 for k, v in f.items()}
res["algorithmic_bytes_per_launch"] = 8.0 * 100000 * 100000
json.dump(res, open(out + "/pmc_sjlt_traffic.json", "w"), indent=1)
print(json.dumps(res))
PY
