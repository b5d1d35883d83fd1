#!/bin/bash
# Round 5, first GPU call: the single-launch tree pass on the MI355X -- its GPU test, an A/B of the default bench line with the
# pass on / off, and the launch-by-launch timeline of the last step.  usage (GPU box, repo root): bash tools/round5_tree_ab.sh <tag>
tag=${1:-r05_tree}; out=/root/repo/gpurun_out/$tag; mkdir -p $out; cd /root/repo
timeout 600 python -m pytest tests/test_hss_gpu.py -x -q -m gpu -k "one_launch or factor_ahead" > $out/pytest_tree.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_tree.log
export STRUMPACK_AMD_BENCH_NO_PMC=1
for mode in 1 0; do
  STRUMPACK_AMD_TREE_LAUNCH=$mode timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $out/bench_tree$mode.json 2> $out/bench_tree$mode.err; echo "bench tree=$mode rc=$?"
done
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python /root/repo/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $out/bench_under_rocprof.json 2> $out/kt.err
python /root/repo/tools/trace_tail.py $out/kt > /dev/null 2>&1
cp $out/kt/kt_kernel_stats.csv $out/kernel_stats.csv 2>/dev/null
[ -f $out/kt/trace_tail.txt ] && cp $out/kt/trace_tail.txt $out/trace_tail.txt
rm -rf $out/kt
python - $out <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_tree*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "ms %.3f"%d["ms_per_step"], {k:round(v*1e3,3) for k,v in d.get("phases_s",{}).items()}, d["hss"], d["checks"])
    except Exception as e: print(f, "failed", e)
PY
