python tools/probes/gemm_tall_probe.py 2>&1 | grep -v amdgpu.ids; HSSK_GEMM_NO_TALL=1 python tools/probes/gemm_tall_probe.py 2>&1 | grep -v amdgpu.ids
