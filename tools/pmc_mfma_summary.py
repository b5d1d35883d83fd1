"""Per-kernel sums of the counter passes of tools/pmc_mfma.sh -> JSON (profiles/r04_pmc_mfma.json).
mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES per SE-normalised); see MI355X_MICROARCH.md for the units."""
import collections
import csv
import glob
import json
import sys

out, npass = sys.argv[1], int(sys.argv[2])
want = ("sketch_kernel", "dgemm_kernel", "leaf_update_kernel", "gemm_vbatched_kernel", "gemm_tall_kernel", "gemm_panel_kernel",
        "sweep_mma_kernel")


def kname(k):
    k = k.replace("void ", "").replace("(anonymous namespace)::", "")
    return k.split("(")[0]


acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
dur = collections.defaultdict(float)
for i in range(1, npass + 1):
    fs = glob.glob(f"{out}/pass{i}/**/*counter_collection.csv", recursive=True)
    if not fs:
        continue
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if not any(w in k for w in want):
            continue
        key = kname(k)
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[key][r["Counter_Name"]] += 1
    ts = glob.glob(f"{out}/pass{i}/**/*kernel_trace.csv", recursive=True)
    if ts and i == 1:
        for r in csv.DictReader(open(ts[0])):
            k = r["Kernel_Name"]
            if any(w in k for w in want):
                key = kname(k)
                dur[key] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
res = {}
for key, v in sorted(acc.items()):
    d = {a: b for a, b in v.items()}
    d["launches"] = max(calls[key].values())
    d["total_ms_under_counters"] = round(dur.get(key, 0.), 4)
    if v.get("SQ_BUSY_CYCLES") and v.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        # SQ_VALU_MFMA_BUSY_CYCLES counts per-SIMD busy cycles summed over the chip's 1024 SIMDs; SQ_BUSY_CYCLES is per
        # shader engine (32 on the chip): busy fraction = mfma / (4 SIMD x 256 CU) over busy / 32 SE
        d["mfma_busy_frac_of_sq_busy"] = round((v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (v["SQ_BUSY_CYCLES"] / 32.0), 4)
    if v.get("GRBM_GUI_ACTIVE") and v.get("SQ_INSTS_VALU_MFMA_MOPS_F64"):
        # MOPS_F64 counts 512 flop units? reported raw; flops per GUI-active cycle for cross-checking against 32 flop/clk/SIMD
        d["mops_f64_per_gui_cycle"] = round(v["SQ_INSTS_VALU_MFMA_MOPS_F64"] / v["GRBM_GUI_ACTIVE"], 3)
    res[key] = d
print(json.dumps(res, indent=1))
