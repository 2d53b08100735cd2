"""Summarise the SQ counter pass of tools/profile_sq.sh into profiles/.

  python tools/summarize_sq.py TAG OUT.json
per kernel of the library: every counter summed over its dispatches of one bench step, plus the shares of wave cycles
spent issuing VALU instructions and waiting."""
import csv, glob, json, re, sys
from collections import defaultdict

tag, out = sys.argv[1], sys.argv[2]
agg = defaultdict(lambda: defaultdict(float))
for f in glob.glob("gpurun_out/prof_%s_sq/**/*counter_collection.csv" % tag, recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            m = re.search(r"(k_\w+)", row["Kernel_Name"])
            if not m or "at::native" in row["Kernel_Name"]:
                continue
            agg[m.group(1)][row["Counter_Name"]] += float(row["Counter_Value"])
res = {}
for k, c in agg.items():
    d = dict(c)
    wc = d.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        d["valu_share_of_wave_cycles"] = round(d.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 4)
        d["wait_share_of_wave_cycles"] = round(d.get("SQ_WAIT_ANY", 0.0) / wc, 4)
    res[k] = d
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
for k, d in sorted(res.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    print("%-24s valu %.3f wait %.3f  VALU insts %.3g" % (k, d.get("valu_share_of_wave_cycles", 0), d.get("wait_share_of_wave_cycles", 0), d.get("SQ_INSTS_VALU", 0)))
