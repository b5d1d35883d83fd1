#!/bin/bash
# One GPU call that refreshes everything under profiles/ for a round: the default bench line, the rocprofv3
# kernel summary of the same command, and the HBM-traffic PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs).
# usage (on the GPU box, from the repo root): bash tools/round_profile.sh <tag>
tag=${1:-rXX}
out=/root/repo/gpurun_out/$tag
mkdir -p $out
cd /root/repo
python bench.py > $out/bench_n1.json 2> $out/bench_n1.err
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python /root/repo/bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/kt.err
for c in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_DRAM_32B_sum; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_$c -o p --output-format csv -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_$c.log 2>&1
done
python - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_DRAM_32B_sum"):
    fs = glob.glob(out + "/pmc_%s/**/*counter_collection.csv" % c, recursive=True)
    if not fs:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "dgemm_kernel" in k or "fill_toeplitz" in k:
            name = k[k.index("dgemm_kernel"):k.index(">") + 1] if "dgemm_kernel" in k else "fill_toeplitz_kernel"
            acc[name].append(float(r["Counter_Value"]))
    res[c + "_per_launch"] = {k: sum(v) / len(v) for k, v in acc.items()}
main = [k for k in res.get("FETCH_SIZE_per_launch", {}) if k.endswith("true, 0>")]
if main:
    # MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE (KB) tallies the 128-byte requests of a wide (16 B/lane)
    # streaming read at 64 B -> double it.  Cross-check: TCC_EA0_RDREQ_DRAM_32B counts 32-byte units exactly.
    f = {k: res["FETCH_SIZE_per_launch"][k] * 1024 * 2 for k in main}
    x = {k: res.get("TCC_EA0_RDREQ_DRAM_32B_sum_per_launch", {}).get(k, 0) * 32 for k in main}
    w = {k: res.get("WRITE_SIZE_per_launch", {}).get(k, 0) * 1024 for k in main}
    res["main_launch_read_bytes_FETCH_SIZE_x2"] = f
    res["main_launch_read_bytes_RDREQ_DRAM_32B_x32"] = x
    res["main_launch_write_bytes_WRITE_SIZE"] = w
    res["hbm_bytes_per_launch"] = sum(f.values()) / len(f) + sum(w.values()) / len(w)
    res["algorithmic_read_bytes_per_launch"] = 8.0 * 100000 * 98304 + 8.0 * 192 * 100000   # A columns of the main grid + R once
json.dump(res, open(out + "/pmc_traffic_raw.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
head -12 $out/kt/kt_kernel_stats.csv | cut -c1-160
tail -c 1500 $out/bench_n1.json
# the other two workloads: SJLT-sketch variant (bench, kernel summary, HBM traffic) and the kernel-matrix front end
bash /root/repo/tools/round_profile_sjlt.sh $tag > $out/sjlt_profile.log 2>&1
cd /root/repo
python bench.py --workload kernel --no-cpu-baseline > $out/bench_kernel_n1.json 2> $out/bench_kernel.err
(cd /tmp; timeout 120 rocprofv3 --kernel-trace --stats -d $out/kt_kernel -o kt --output-format csv -- python /root/repo/bench.py --workload kernel --no-cpu-baseline > $out/bench_kernel_rocprof.json 2> $out/kt_kernel.err)
# copy into profiles/ afterwards:  bench_n1.json -> rNN_bench_n1.json, kt/kt_kernel_stats.csv -> rNN_kernel_stats_bench_n100k.csv,
# pmc_traffic_raw.json -> rNN_pmc_traffic.json, bench_sjlt_n1.json, kt_sjlt/kt_kernel_stats.csv, pmc_sjlt_traffic.json,
# bench_kernel_n1.json, kt_kernel/kt_kernel_stats.csv
