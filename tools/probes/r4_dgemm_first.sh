# round 4, first GPU call: the LDS-DMA form of the sketch GEMM -- parity tests, then the in-process A/B against the four-wave form
O=gpurun_out/r04a; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k dgemm > $O/pytest_dgemm.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_dgemm.log
timeout 200 python tools/dgemm_ab.py 100000 > $O/dgemm_ab.log 2>&1; echo "ab rc=$?"; cat $O/dgemm_ab.log
