"""Dev tool: mean per-dispatch counter values of one kernel from rocprofv3 counter_collection csv files."""
import csv, glob, os, sys
from collections import defaultdict
root, pat = sys.argv[1], sys.argv[2]
tot = defaultdict(float); disp = defaultdict(set)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            disp[r["Counter_Name"]].add((f, r["Dispatch_Id"]))
for k in sorted(tot):
    print("%-28s %16.0f per launch (%d launches)" % (k, tot[k] / max(1, len(disp[k])), len(disp[k])))
