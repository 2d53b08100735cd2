"""Summarise the rocprofv3 passes of tools/profile_c2.sh into profiles/.

  python tools/summarize_profile.py TAG OUTPREFIX
writes OUTPREFIX_kernel_stats.csv (copy of rocprofv3's --stats table) and OUTPREFIX_pmc.json
(per kernel: launches, FETCH_SIZE and WRITE_SIZE summed over its dispatches of ONE bench step, in bytes,
raw as rocprofv3 reports them (KB * 1024) -- MI355X_MICROARCH.md: on gfx950 FETCH_SIZE under-counts wide
coalesced reads by 2x, other access widths are uncalibrated)."""
import csv, glob, json, re, shutil, sys
from collections import defaultdict

tag, outprefix = sys.argv[1], sys.argv[2]


def short(name):
    m = re.search(r"(k_\w+|bloom_stamp_fill)", name)
    return m.group(1) if m else name[:40]


stats = glob.glob("gpurun_out/prof_%s_trace/**/*kernel_stats.csv" % tag, recursive=True)
if stats:
    shutil.copy(stats[0], outprefix + "_kernel_stats.csv")
pmc = {}
for counter, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    files = glob.glob("gpurun_out/prof_%s_%s/**/*counter_collection.csv" % (tag, sub), recursive=True)
    agg = defaultdict(lambda: [0.0, 0])
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                a = agg[short(row["Kernel_Name"])]
                a[0] += float(row["Counter_Value"]) * 1024.0
                a[1] += 1
    for k, (b, n) in agg.items():
        pmc.setdefault(k, {})[counter + "_bytes"] = b
        pmc[k]["launches"] = n
json.dump(pmc, open(outprefix + "_pmc.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(pmc.items(), key=lambda kv: -kv[1].get("FETCH_SIZE_bytes", 0)):
    print("%-20s launches %4d  fetch %8.3f GB  write %8.3f GB" % (k, v.get("launches", 0), v.get("FETCH_SIZE_bytes", 0) / 1e9, v.get("WRITE_SIZE_bytes", 0) / 1e9))
