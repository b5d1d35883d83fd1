"""Host-side simulation of hssk_knn's candidate scan on the bench data (N = 1e5 uniform points in R^8, cobble order, k = 64,
256-point tiles visited outwards from the query's own tile): how many 4-candidate trips of a wave take the heap-insertion
branch.  Uses the clustering of the CPU emulator build (tests/emu) -- a development aid, no GPU needed.
Result (12 sampled waves): 9.0 % of the trips insert (2240 of 25000 per wave), 298 insertions per query; half of them
happen in the first 65 of 391 tiles, the first 3 tiles insert on nearly every trip."""
import numpy as np, sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import emu_lib
from strumpack_amd import kernel as KM
lib = KM.load(emu_lib.build())
n, d, k, C = 100000, 8, 64, 256
rng = np.random.default_rng(2025)
X = rng.random((n, d))
t=time.time()
Xp, perm, leaves = KM.clustering(lib, X, "cobble", 256)
print("clustered", time.time()-t, Xp.shape)
Xp = np.ascontiguousarray(Xp if Xp.shape[0]==n else Xp.T)
ntile = (n + C - 1)//C
waves = rng.choice(n//64, 12, replace=False)
tot_trips = 0; ins_trips = 0; ins_lane = 0; per_tile_ins = np.zeros(ntile)
for w in waves:
    q = np.arange(w*64, w*64+64)
    own = (w*64 // 256 * 256)//C
    best = np.full((64, k), np.inf)
    thresh = np.full(64, np.inf)
    xq = Xp[q]
    for t in range(ntile):
        off = (t+1)>>1
        c0 = (((own+off) if (t&1) else (own-off+ntile)) % ntile)*C
        cand = Xp[c0:c0+C]
        D = ((xq[:,None,:]-cand[None,:,:])**2).sum(-1)   # 64 x C
        ids = np.arange(c0, c0+cand.shape[0])
        D[q[:,None]==ids[None,:]] = np.inf
        # trips of 4 candidates
        for c in range(0, cand.shape[0], 4):
            blk = D[:, c:c+4]
            p = blk < thresh[:,None]
            tot_trips += 1
            if p.any():
                ins_trips += 1
                lanes = np.where(p.any(1))[0]
                ins_lane += int(p.sum())
                per_tile_ins[t] += 1
                for l in lanes:
                    m = np.concatenate([best[l], blk[l][p[l]]])
                    m.sort()
                    best[l] = m[:k]
                    thresh[l] = best[l][-1]
print("trips", tot_trips, "with insertion", ins_trips, "fraction %.3f"%(ins_trips/tot_trips), "lane insertions per query %.1f"%(ins_lane/(64*len(waves))))
cs = np.cumsum(per_tile_ins)/per_tile_ins.sum()
for frac in (0.5,0.8,0.9,0.99): print("tiles to reach", frac, int(np.searchsorted(cs, frac)), "of", ntile)
print("insertion trips in first 8 tiles", per_tile_ins[:8]/len(waves), " of 64 trips per tile")
print("avg insertion trips per tile in tiles 50..ntile: %.2f of 64"%(per_tile_ins[50:].mean()/len(waves)))
