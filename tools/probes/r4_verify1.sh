# round 4, second GPU call: whole GPU tier + smoke with the new sketch kernel, the default bench line (with its live traffic
# passes), the kernel summary of one bench run, the MFMA counter passes
O=/root/repo/gpurun_out/r04b; mkdir -p $O; cd /root/repo
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 400 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; cat $O/bench_n1.json
cd /tmp; export TMPDIR=/tmp
STRUMPACK_AMD_BENCH_NO_PMC=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python /root/repo/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.err
python /root/repo/tools/trace_tail.py $O/kt > /dev/null 2>&1
cp $O/kt/kt_kernel_stats.csv $O/kernel_stats_bench_n100k.csv 2>/dev/null; [ -f $O/kt/trace_tail.txt ] && cp $O/kt/trace_tail.txt $O/trace_tail.txt
rm -rf $O/kt
cd /root/repo; bash tools/pmc_mfma.sh r04b > $O/pmc.log 2>&1; tail -5 $O/pmc.log
for i in 1 2 3 4; do rm -f $O/pass$i/*kernel_trace.csv.bak; done
du -sh $O
