#!/usr/bin/env python3
"""Writes tests/golden/oracle_end_to_end.json: end-to-end vectors of the graph-construction path on tiny seeded inputs.

These vectors come from the CPU restatement (oracle/ray_oracle.c), NOT from a run of the reference: the reference cannot
be built here (RayPlatform is absent), and it holds no end-to-end output of its own.  What pins the
restatement to the reference is tests/golden/reference_known_answers.json.  These vectors freeze the restatement's
results, so that the oracle and the HIP path cannot drift together unnoticed: tests/test_golden_vectors.py checks the
oracle against them on the CPU and the HIP path against them on the GPU (without calling the oracle).

    python tests/golden/make_oracle_vectors.py        # rewrites the JSON next to this script
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

CASES = [
    # name, reads (ray_amd.synth.make_reads_numpy arguments), k, ranks, -bloom-filter-bits (None = Ray's default sizing)
    dict(name="k31_default_bloom", reads=dict(genome=5000, read_len=100, n_reads=1000, seed=101, error=0.01), k=31, nranks=1, bloom=None),
    dict(name="k31_squeezed_bloom_false_positives", reads=dict(genome=5000, read_len=100, n_reads=1000, seed=102, error=0.01), k=31, nranks=1, bloom=20000),
    dict(name="k31_bloom_off", reads=dict(genome=5000, read_len=100, n_reads=1000, seed=103, error=0.01), k=31, nranks=1, bloom=0),
    dict(name="k21_ragged_reads", reads=dict(genome=4000, read_len=80, n_reads=900, seed=104, error=0.02, ragged=True), k=21, nranks=1, bloom=15000),
    dict(name="k63_two_words", reads=dict(genome=6000, read_len=150, n_reads=800, seed=105, error=0.01), k=63, nranks=1, bloom=30000),
    dict(name="k31_two_ranks", reads=dict(genome=5000, read_len=100, n_reads=1001, seed=106, error=0.01), k=31, nranks=2, bloom=20000),
    dict(name="k31_eight_ranks", reads=dict(genome=5000, read_len=100, n_reads=1003, seed=107, error=0.01), k=31, nranks=8, bloom=6000),
    dict(name="k15_tiny_full_records", reads=dict(genome=60, read_len=30, n_reads=40, seed=108, error=0.02), k=15, nranks=1, bloom=600),
]


def digest(rec):
    """sha256 over the sorted export records: keys, coverages, edge bytes"""
    h = hashlib.sha256()
    h.update(rec["key"].tobytes())
    h.update(rec["coverage"].tobytes())
    h.update(rec["edges"].tobytes())
    return h.hexdigest()


def expected(ro, case):
    from helpers import oracle_run
    from ray_amd.synth import make_reads_numpy
    packed, offsets, lengths = make_reads_numpy(**case["reads"])
    want = oracle_run(ro, packed, offsets, lengths, case["k"], case["nranks"], case["bloom"], purge=(case["nranks"] == 1))
    g = want["graph"]
    out = {"occurrences": int(want["occ"]), "size": int(want["size"]),
           "bloom": [[int(x) for x in t] for t in want["bloom"]],
           "histogram": [[int(c) for c in want["hist"][0]], [int(n) for n in want["hist"][1]]],
           "input_sha256": hashlib.sha256(packed.tobytes() + offsets.tobytes() + lengths.tobytes()).hexdigest()}
    if case["nranks"] == 1:
        out["records"] = {"pass1": [len(want["pass1"]), digest(want["pass1"])], "pass2": [len(want["pass2"]), digest(want["pass2"])],
                          "purged": [len(want["purged"]), digest(want["purged"])]}
        out["degrees"] = [int(x) for x in want["degrees"].ravel()]
        if "full" in case["name"]:
            r = want["purged"]
            out["purged_records"] = [[[int(w) for w in (key if hasattr(key, "__len__") else [key])], int(c), int(e)]
                                     for key, c, e in zip(r["key"].tolist(), r["coverage"].tolist(), r["edges"].tolist())]
    else:
        out["records"] = {"pass2_per_rank": [[len(g.export(r)), digest(g.export(r))] for r in range(case["nranks"])]}
    g.close()
    return out


def main():
    from oracle import ray_oracle as ro
    ro.build()
    doc = {"source": "oracle/ray_oracle.c (CPU restatement; not a run of the reference, see make_oracle_vectors.py)", "cases": []}
    for case in CASES:
        doc["cases"].append(dict(case, expected=expected(ro, case)))
    with open(os.path.join(HERE, "oracle_end_to_end.json"), "w") as fh:
        json.dump(doc, fh, indent=1)
    print("wrote %d cases" % len(doc["cases"]))


if __name__ == "__main__":
    main()
