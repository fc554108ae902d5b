"""Condense rocprofv3 CSV output (kernel trace stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    name = name.split("(")[0]
    return name.replace("void dflo::", "").replace("dflo::", "")[:60]


print("== kernel stats (rocprofv3 --kernel-trace --stats)")
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-62s calls %6s  avg %10.1f us  total %10.1f us  %5s%%" % (
            short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3, r["Percentage"]))

vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        vals[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
traffic = {}
print("== PMC (mean per dispatch)")
for k in sorted(vals):
    print(k)
    c = {n: sum(v) / len(v) for n, v in vals[k].items()}
    for n in sorted(c):
        print("   %-24s %16.1f  (n=%d)" % (n, c[n], len(vals[k][n])))
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # guide: FETCH_SIZE on gfx950 counts 64 B per 128-B request for wide coalesced reads -> x2;
        # FETCH_SIZE / WRITE_SIZE are in KiB
        fetch, write = c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024
        print("   hbm bytes/launch: fetch(raw) %.4e  fetch(x2 gfx950 correction) %.4e  write %.4e  total(corrected) %.4e" % (
            fetch, 2 * fetch, write, 2 * fetch + write))
    if "SQ_WAVE_CYCLES" in c and "SQ_WAIT_ANY" in c:
        wc = c["SQ_WAVE_CYCLES"]
        print("   wave-cycle split: wait_any %.1f%%  wait_inst_any %.1f%%  active_inst_any %.1f%%  active_valu %.1f%%" % (
            100 * c["SQ_WAIT_ANY"] / wc, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
            100 * c.get("SQ_ACTIVE_INST_VALU", 0) / wc))
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        traffic.setdefault(k, {})["hbm_bytes_per_launch"] = 2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024
        traffic[k]["fetch_kib_raw"] = c["FETCH_SIZE"]
        traffic[k]["write_kib"] = c["WRITE_SIZE"]
    if "GRBM_GUI_ACTIVE" in c and "SQ_ACTIVE_INST_VALU" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0   # summed over the 8 XCDs
        print("   pipes: kernel cycles %.0f  valu_busy = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x cycles) = %.3f  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles) = %.4f" % (
            cyc, 4 * c["SQ_ACTIVE_INST_VALU"] / 1024 / cyc, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024 / cyc))
    if "TCC_HIT_sum" in c:
        print("   L2 hit rate %.1f%%" % (100 * c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])))

import json
json.dump(traffic, open(os.path.join(root, "traffic.json"), "w"), indent=1)
