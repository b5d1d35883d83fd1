"""Stage clocks of id_group_kernel (library built with -DIDG_TIMING as strumpack_amd/lib/libstrumpack_amd_timing.so): per
workgroup of panel 0, 10 ns ticks summed over the steps -- local arg max, candidate exchange, reflector, update."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from strumpack_amd import hssk as K
hk = K.Hssk(os.path.join(root, "strumpack_amd", "lib", "libstrumpack_amd_timing.so"))
rng = np.random.default_rng(3)
for (d, m, count, mr) in ((256, 256, 1, 129), (256, 256, 94, 129), (192, 391, 8, 1000)):
    descs, keep = [], []
    for i in range(count):
        A = rng.standard_normal((d, 140)) @ rng.standard_normal((140, m))
        dW = hk.array(A); dperm = hk.empty((m,), np.int32); drank = hk.empty((1,), np.int32); dwork = hk.empty((3 * m,))
        keep.append((dW, dperm, drank, dwork))
        descs.append(K.IdDesc(dW.ptr, d, d, m, 1e-14, 1e-14, mr, dperm.ptr, drank.ptr, dwork.ptr))
    hk.batch("hssk_id_vbatched", descs); hk.sync()
    w = keep[0][3].get(); r = int(keep[0][2].get()[0])
    H = 2 if m <= 256 else 4
    for h in range(H):
        t = w[8 * h: 8 * h + 5]
        print("d %d m %d count %d rank %d wg %d: per step us  argmax %.2f  exchange %.2f  reflector %.2f  update %.2f  (owner of %d steps)" %
              (d, m, count, r, h, *(t[:4] / 100. / max(r, 1)), int(t[4])), flush=True)
