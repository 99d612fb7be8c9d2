#!/usr/bin/env python
"""Per-kernel MFMA utilisation from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass (with
--kernel-trace): util = MFMA busy cycles (summed over the SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)."""
import collections, csv, glob, json, sys
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"how": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE of `python bench.py "
              "--steps 2 --warmup 1 --no-cpu-baseline --no-e2e`; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * "
              "1024 SIMDs) — GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (checked against the analytic MFMA count "
              "of k_gemm_nt_bf16: 32 cycles x 32 MFMAs per wave tile); the GEMMs of this path are HBM-bound (K = 128..768), so a low MFMA utilisation is the "
              "expected reading", "kernels": {}}
for k, c in acc.items():
    if "gemm" not in k.lower() and "Cijk" not in k and "k_encoder_" not in k:
        continue
    mf, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0.0]), c.get("GRBM_GUI_ACTIVE", [0.0])
    name = k.split("(")[0].replace("void ", "")[:90]
    tot_mf, tot_gui = sum(mf), sum(gui)
    out["kernels"][name] = {"launches": len(mf), "SQ_VALU_MFMA_BUSY_CYCLES_sum": tot_mf, "GRBM_GUI_ACTIVE_sum": tot_gui,
                            "mfma_util": (tot_mf / (tot_gui / 8 * 1024)) if tot_gui else None}
json.dump(out, sys.stdout, indent=1)
