"""Summarise the rocprofv3 passes of tools/profile_c2.sh into profiles/.

  python tools/summarize_profile.py TAG OUTPREFIX
writes OUTPREFIX_kernel_stats.csv (copy of rocprofv3's --stats table) and OUTPREFIX_pmc.json
(per kernel: launches, FETCH_SIZE and WRITE_SIZE summed over its dispatches of ONE bench step, in bytes,
raw as rocprofv3 reports them (KB * 1024) -- MI355X_MICROARCH.md: on gfx950 FETCH_SIZE under-counts wide
coalesced reads by 2x, other access widths are uncalibrated; from the l2 / atomic passes, when present, the raw
counts TCC_HIT / TCC_MISS / TCC_REQ / TCC_ATOMIC / TCC_EA0_ATOMIC / TCC_READ / TCC_WRITE / TCP_TOTAL_ATOMIC_* and
L2_hit_rate = TCC_HIT / (TCC_HIT + TCC_MISS); "_steps" = bench steps per pass)."""
import csv, glob, json, os, re, shutil, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from csrc_hash import csrc_sha256  # noqa: E402

tag, outprefix = sys.argv[1], sys.argv[2]


def short(name):
    m = re.search(r"(k_\w+|bloom_stamp_fill)", name)
    return m.group(1) if m else name[:40]


stats = glob.glob("gpurun_out/prof_%s_trace/**/*kernel_stats.csv" % tag, recursive=True)
if stats:
    shutil.copy(stats[0], outprefix + "_kernel_stats.csv")
pmc = {}
PASSES = [("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")]
PASSES += [(c, "l2") for c in ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_ATOMIC_sum")]
PASSES += [(c, "atomic") for c in ("TCC_EA0_ATOMIC_sum", "TCC_READ_sum", "TCC_WRITE_sum", "TCP_TOTAL_ATOMIC_WITH_RET_sum", "TCP_TOTAL_ATOMIC_WITHOUT_RET_sum")]
for counter, sub in PASSES:
    files = glob.glob("gpurun_out/prof_%s_%s/**/*counter_collection.csv" % (tag, sub), recursive=True)
    agg = defaultdict(lambda: [0.0, 0])
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                a = agg[short(row["Kernel_Name"])]
                a[0] += float(row["Counter_Value"]) * (1024.0 if counter.endswith("_SIZE") else 1.0)
                a[1] += 1
    for k, (b, n) in agg.items():
        pmc.setdefault(k, {})[(counter + "_bytes") if counter.endswith("_SIZE") else counter[:-4]] = b
        pmc[k]["launches"] = n
# the bench makes its synthetic reads with torch inside the profiled process: keep the library's kernels only
pmc = {k: v for k, v in pmc.items() if k.startswith("k_") or k == "bloom_stamp_fill"}
for k, v in pmc.items():
    if v.get("TCC_HIT", 0) + v.get("TCC_MISS", 0) > 0:
        v["L2_hit_rate"] = v["TCC_HIT"] / (v["TCC_HIT"] + v["TCC_MISS"])
pmc["_steps"] = 1
pmc["_csrc_sha256"] = csrc_sha256()   # the kernels these counters belong to: bench.py quotes the traffic only for the same sources
pmc["_bench_args"] = os.environ.get("BENCH_ARGS", "")
json.dump(pmc, open(outprefix + "_pmc.json", "w"), indent=1, sort_keys=True)
if not pmc["_bench_args"]:   # the default workload's summary is also the one bench.py reads
    json.dump(pmc, open(os.path.join(os.path.dirname(outprefix) or ".", "c2_pmc_current.json"), "w"), indent=1, sort_keys=True)
tot = 0.0
for k, v in sorted(((k, v) for k, v in pmc.items() if isinstance(v, dict)), key=lambda kv: -kv[1].get("FETCH_SIZE_bytes", 0)):
    tot += 2 * v.get("FETCH_SIZE_bytes", 0) + v.get("WRITE_SIZE_bytes", 0)
    print("%-22s launches %4d  fetch %8.3f GB  write %8.3f GB  L2 hit %5.3f  atomics at L2 %.3e  to memory %.3e  TCP ret/noret %.3e/%.3e"
          % (k, v.get("launches", 0), v.get("FETCH_SIZE_bytes", 0) / 1e9, v.get("WRITE_SIZE_bytes", 0) / 1e9, v.get("L2_hit_rate", float("nan")),
             v.get("TCC_ATOMIC", 0), v.get("TCC_EA0_ATOMIC", 0), v.get("TCP_TOTAL_ATOMIC_WITH_RET", 0), v.get("TCP_TOTAL_ATOMIC_WITHOUT_RET", 0)))
print("2 x FETCH + WRITE over the step: %.1f GB" % (tot / 1e9))
