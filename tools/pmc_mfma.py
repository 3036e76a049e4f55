"""Dev tool: MFMA utilisation per kernel from rocprofv3 PMC passes -> profiles/<tag>_mfma_util.json.

usage: python tools/pmc_mfma.py <dir with counter_collection csv files (searched recursively)> <out.json>
Definition (the gfx94x derived-metric form; ROCm 7.2 ships no gfx950 section, MI355X_MICROARCH.md "rocprofv3 PMC slots"):
  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs  x  256 CUs x 4 SIMDs)
SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy cycles summed over all SIMDs; GRBM_GUI_ACTIVE is reported summed over
the 8 XCDs.  SQ_INSTS_MFMA (wave-instructions) and SQ_BUSY_CYCLES are kept beside it."""
import csv, glob, json, os, sys
from collections import defaultdict
root, out = sys.argv[1], sys.argv[2]
tot, disp = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(set))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "hicom::" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("hicom::")[1].split("(")[0]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k][r["Counter_Name"]].add((f, r["Dispatch_Id"]))
res = {"definition": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs), per launch, rocprofv3 --pmc (own pass)",
       "kernels": {}}
for k in sorted(tot):
    mean = {c: tot[k][c] / max(1, len(disp[k][c])) for c in tot[k]}
    e = dict(mean)
    e["launches"] = max(len(v) for v in disp[k].values())
    if mean.get("GRBM_GUI_ACTIVE"):
        e["mfma_busy_frac"] = mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (mean["GRBM_GUI_ACTIVE"] / 8 * 256 * 4)
    res["kernels"][k] = e
    print("%-52s launches %4d  MFMA busy %6.3f  insts %12.0f" % (k[:52], e["launches"], e.get("mfma_busy_frac", float("nan")),
                                                              mean.get("SQ_INSTS_MFMA", 0)))
json.dump(res, open(out, "w"), indent=1)
