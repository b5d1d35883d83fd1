"""Development aid (CPU only): the kernel-level and C-interface cases added in round 2 (typed operand streaming, image
expansion, wide split-K reduce, concurrent calls, in-place reduced blocks of the factorization, Schur complement) on the
sanitizer build of the emulator library -- run by tools/asan_emu.sh with libasan / libubsan preloaded."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import emu_lib
emu_lib.PATH = sys.argv[1]
emu_lib.build = lambda: emu_lib.PATH
from strumpack_amd import capi, hssk as K
import hss_cases as HC, kernel_cases as KC
L = capi.load(emu_lib.PATH)
hk = K.Hssk(emu_lib.PATH)
KC.case_expand_image(hk); print("expand ok", flush=True)
KC.case_contract_codes(hk); print("contract ok", flush=True)
KC.case_dgemm(hk, 64, 72, 12800, 1, alpha=-1.5, beta=0.5, lda_pad=0, ldb_pad=0); print("dgemm wide ok", flush=True)
KC.case_upload_two_threads(hk); print("two threads ok", flush=True)
HC.check_host_stream_blocks(L, n=200); print("stream blocks ok", flush=True)
HC.check_concurrent_ops(L); print("concurrent ok", flush=True)
HC.check_api_semantics(L); print("api ok", flush=True)
c = HC.golden_cases()
for name in ("HSS_seq_5", "HSS_seq_12", "HSS_seq_22"):
    HC.check_against_golden(L, c[name]); print(name, "ok", flush=True)
for name in ("HSS_seq_2", "HSS_seq_11"):
    HC.check_schur(L, c[name]); print("schur", name, "ok", flush=True)
