"""End-to-end vectors under tests/golden/oracle_end_to_end.json (made by tests/golden/make_oracle_vectors.py from the CPU
restatement; they freeze its results, they are not outputs of the reference -- see that script's header).
CPU: the oracle still reproduces them.  GPU: the HIP path reproduces them through the C ABI, without the oracle."""
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_oracle_vectors as gold  # noqa: E402

DOC = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_end_to_end.json")))
IDS = [c["name"] for c in DOC["cases"]]


@pytest.mark.parametrize("case", DOC["cases"], ids=IDS)
def test_oracle_reproduces_the_vectors(oracle, case):
    spec = {key: case[key] for key in ("name", "reads", "k", "nranks", "bloom")}
    assert gold.expected(oracle, spec) == case["expected"]


def test_vector_list_matches_the_generator():
    assert [{key: c[key] for key in ("name", "reads", "k", "nranks", "bloom")} for c in DOC["cases"]] == gold.CASES


def _reads(case):
    import hashlib
    from ray_amd.synth import make_reads_numpy
    packed, offsets, lengths = make_reads_numpy(**case["reads"])
    assert hashlib.sha256(packed.tobytes() + offsets.tobytes() + lengths.tobytes()).hexdigest() == case["expected"]["input_sha256"]
    return packed, offsets, lengths


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in DOC["cases"] if c["nranks"] == 1], ids=[c["name"] for c in DOC["cases"] if c["nranks"] == 1])
@pytest.mark.parametrize("sweep", [0, 1 | 16 | (9 << 8), 2 | 32])
def test_hip_path_reproduces_the_vectors(case, sweep):
    import ray_amd
    from ray_amd.graph import RG_BLOOM_DEFAULT
    want = case["expected"]
    packed, offsets, lengths = _reads(case)
    b = ray_amd.GraphBuilder(case["k"], bloom_bits=RG_BLOOM_DEFAULT if case["bloom"] is None else case["bloom"], sweep=sweep)
    b.add_reads(packed, offsets, lengths)
    b.count_kmers()
    st = b.stats()
    assert st["occurrences"] == want["occurrences"] and b.size_() == want["size"]
    assert [[st["bloom_bits"], st["bloom_set_bits"], st["bloom_insertions"]]] == want["bloom"]
    rec = b.export()
    assert [len(rec), gold.digest(rec)] == want["records"]["pass1"]
    cov, cnt = b.histogram()
    assert [cov.tolist(), cnt.tolist()] == want["histogram"]
    b.add_edges()
    rec = b.export()
    assert [len(rec), gold.digest(rec)] == want["records"]["pass2"]
    b.purge_edges()
    rec = b.export()
    assert [len(rec), gold.digest(rec)] == want["records"]["purged"]
    assert [int(x) for x in np.asarray(b.degree_distribution()).ravel()] == want["degrees"]
    if "purged_records" in want:
        got = [[[int(w) for w in (key if hasattr(key, "__len__") else [key])], int(c), int(e)]
               for key, c, e in zip(rec["key"].tolist(), rec["coverage"].tolist(), rec["edges"].tolist())]
        assert got == want["purged_records"]
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in DOC["cases"] if c["nranks"] > 1], ids=[c["name"] for c in DOC["cases"] if c["nranks"] > 1])
@pytest.mark.parametrize("route", ["records", "entries"])
def test_hip_ranks_reproduce_the_vectors(case, route):
    from test_multirank_gpu import run_ranks
    want = case["expected"]
    packed, offsets, lengths = _reads(case)
    builders = run_ranks(packed, offsets, lengths, case["k"], case["nranks"], case["bloom"], sweep=1 | 16 | (9 << 8), route=route,
                         one_merge=(route == "records"))
    assert sum(b.stats()["occurrences"] for b in builders) == want["occurrences"]
    assert [[b.stats()[f] for f in ("bloom_bits", "bloom_set_bits", "bloom_insertions")] for b in builders] == want["bloom"]
    assert [[len(b.export()), gold.digest(b.export())] for b in builders] == want["records"]["pass2_per_rank"]
    merged = {}
    for b in builders:
        for c, n in zip(*b.histogram()):
            merged[int(c)] = merged.get(int(c), 0) + int(n)
    assert [sorted(merged), [merged[c] for c in sorted(merged)]] == want["histogram"]
    for b in builders:
        b.close()
