# round 4: BLR -- row-panel kernel for the trailing updates, LU of the next diagonal tile issued behind its own update
O=/root/repo/gpurun_out/r04h; mkdir -p $O; cd /root/repo
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" > $O/pytest_k.log 2>&1; echo "kernel tests rc=$?"; tail -2 $O/pytest_k.log
timeout 900 python -m pytest tests/test_blr_front_gpu.py tests/test_hss_gpu.py -x -q -k "front or blr" > $O/pytest_blr.log 2>&1; echo "blr tests rc=$?"; tail -3 $O/pytest_blr.log
run() {  # tag, env, args
  env $2 timeout 600 python bench.py --workload blr_front --no-cpu-baseline $3 > $O/$1.json 2> $O/$1.err
}
run blr64 "X=1" "--steps 3 --warmup 1"
run blr64_norp "HSSK_GEMM_NO_ROWPANEL=1" "--steps 3 --warmup 1"
run blr64_nolu "STRUMPACK_AMD_BLR_NO_LU_AHEAD=1" "--steps 3 --warmup 1"
run blr96 "X=1" "--front-n 96 --steps 3 --warmup 1"
run blr96_norp "HSSK_GEMM_NO_ROWPANEL=1" "--front-n 96 --steps 3 --warmup 1"
run blr_200x100 "X=1" "--front-n 200 --front-ny 100 --steps 2 --warmup 1"
run blr_root "X=1" "--front-n 200 --front-upd none --steps 2 --warmup 1"
run blr_root_nolu "STRUMPACK_AMD_BLR_NO_LU_AHEAD=1" "--front-n 200 --front-upd none --steps 2 --warmup 1"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r04h/b*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d["roofline"]
        print(f.split("/")[-1], "ms %.2f"%d["ms_per_step"], d["phases_ms"], "schur %.1f TF/s %.0f GB/s"%(r.get("tflops",0), r.get("bytes_per_step",0)/max(r.get("phase_ms",1),1e-9)*1e-6), d["checks"])
    except Exception as e: print(f, "failed", e)
PY
