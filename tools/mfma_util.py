"""Matrix-core utilisation per kernel from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` counter-collection CSV:
mfma_util = busy cycles / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs), averaged over a kernel's dispatches.
usage: python tools/mfma_util.py <counter_collection.csv> [out.json]"""
import collections
import csv
import json
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for row in csv.DictReader(open(sys.argv[1])):
    name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*$", "", name)
    agg[name][row["Counter_Name"]] += float(row["Counter_Value"])
    cnt[name][row["Counter_Name"]] += 1
out = {}
for k, v in agg.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in v or "GRBM_GUI_ACTIVE" not in v:
        continue
    busy = v["SQ_VALU_MFMA_BUSY_CYCLES"] / cnt[k]["SQ_VALU_MFMA_BUSY_CYCLES"]
    gui = v["GRBM_GUI_ACTIVE"] / cnt[k]["GRBM_GUI_ACTIVE"]
    out[k] = {"dispatches": cnt[k]["GRBM_GUI_ACTIVE"], "mfma_busy_cycles": busy, "grbm_gui_active": gui,
              "mfma_util": busy / (gui / 8 * 256 * 4) if gui else None}
out = dict(sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_cycles"] * kv[1]["dispatches"])[:12])
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
