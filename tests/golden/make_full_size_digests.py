"""Full-size oracle results for a BASELINE configuration, as digests (tests/golden/full_size_<name>.json).

The reads come from the counter-based generator (oracle/ro_synth.c on the host = tools/rg_synth.hip on the device),
so the GPU test regenerates exactly the same reads in HBM and compares what the HIP path makes of them with what the
oracle -- the sequential restatement of the reference's handlers -- made of them here, at the configuration's full size:
occurrences, GridTable size, Bloom statistics, the whole histogram, minimum / peak / repeat coverage, the bytes of
CoverageDistribution.txt, and the sha256 of the sorted Vertex::dump records after pass 1, pass 2 and the purge.

  python tests/golden/make_full_size_digests.py C1     (about 5 minutes on one core)
  python tests/golden/make_full_size_digests.py C2 [out.json]    (hours on one core, about 12 GB of memory)
"""
import hashlib
import json
import os
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ray_oracle as ro  # noqa: E402
from ray_amd.synth import CONFIGS, n_reads_for  # noqa: E402  (pure Python: the configuration table)

NAMES = {"C1": "C1_ecoli_50x_100bp_k31", "C2": "C2_100Mbp_60x_150bp_k31"}


def digest_records(rec):
    return hashlib.sha256(rec.tobytes()).hexdigest()


def main(which, dst=None):
    name = NAMES[which]
    cfg = CONFIGS[name]
    n = n_reads_for(cfg["genome"], cfg["read_len"], cfg["coverage"])
    t0 = time.time()
    packed, offsets, lengths = ro.synth_generate(n, cfg["genome"], cfg["read_len"], cfg["seed"], cfg["error"])
    print("%s: %d reads generated in %.0f s" % (name, n, time.time() - t0), flush=True)
    g = ro.Graph(cfg["k"], nranks=1)
    g.set_reads(packed, offsets, lengths)
    t0 = time.time()
    g.count_kmers()
    print("count_kmers %.0f s" % (time.time() - t0), flush=True)
    out = {"_made_by": "tests/golden/make_full_size_digests.py " + which, "workload": name, "generator": "counter v1 (tools/rg_synth.hip, oracle/ro_synth.c)",
           "config": cfg, "reads": n, "occurrences": int(g.occurrences()), "size": int(g.size()),
           "bloom": [int(g.bloom_bits(0)), int(g.bloom_set_bits(0)), int(g.bloom_insertions(0))]}
    out["sha256_sorted_records_after_pass1"] = digest_records(g.export())
    cov, cnt = g.histogram()
    out["histogram"] = [cov.tolist(), cnt.tolist()]
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "CoverageDistribution.txt")
        out["minimum_peak_repeat"] = list(ro.coverage_distribution(cov, cnt, path))
        out["sha256_CoverageDistribution_txt"] = hashlib.sha256(open(path, "rb").read()).hexdigest()
    dst = dst or os.path.join(HERE, "full_size_%s.json" % which)
    with open(dst, "w") as f:   # pass 1 is the headline metric: kept even if the run is cut short before the edges are done
        json.dump(dict(out, _partial="pass 1 only: the run was still going"), f, indent=1)
    t0 = time.time()
    g.add_edges()
    print("add_edges %.0f s" % (time.time() - t0), flush=True)
    out["sha256_sorted_records_after_pass2"] = digest_records(g.export())
    t0 = time.time()
    g.purge_edges()
    print("purge_edges %.0f s" % (time.time() - t0), flush=True)
    out["sha256_sorted_records_after_purge"] = digest_records(g.export())
    out["degree_distribution"] = g.degree_distribution().reshape(-1).tolist()
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", dst, flush=True)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "C1", sys.argv[2] if len(sys.argv) > 2 else None)
