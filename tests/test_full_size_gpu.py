"""BASELINE configurations at FULL size, bit-exact against the oracle.

The oracle (the sequential restatement of the reference's handlers) needs minutes (C1) to hours (C2) for these inputs, so
its results were computed once, on the CPU, by tests/golden/make_full_size_digests.py and are committed as digests
(tests/golden/full_size_C1.json, full_size_C2.json: occurrences, GridTable size, Bloom statistics, the whole histogram,
minimum / peak / repeat coverage, the sha256 of CoverageDistribution.txt and of the sorted Vertex::dump records after pass 1,
after pass 2 and after the purge).  The reads come from the counter-based generator, whose device half (tools/rg_synth.hip)
must give the reads its host half (oracle/ro_synth.c) gave the oracle -- checked first.  RG_LIVE_ORACLE=1 runs the oracle
on C1 here as well (about seven minutes on one core) instead of trusting the committed digests."""
import hashlib
import json
import os

import numpy as np
import pytest

from ray_amd.synth import CONFIGS, make_reads_device, n_reads_for

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_device_generator_gives_the_host_generators_reads(oracle):
    for genome, L, n, seed, error, first in ((100000, 100, 5000, 1, 0.0, 0), (50000, 150, 3000, 2, 0.01, 123456789), (3000, 37, 2000, 9, 0.2, 7)):
        hp, ho, hl = oracle.synth_generate(n, genome, L, seed, error, first)
        dp, do, dl = make_reads_device(genome, L, n, seed, error, first)
        assert np.array_equal(dp.cpu().numpy()[: len(hp)], hp), "packed bases differ (genome %d, error %g)" % (genome, error)
        assert np.array_equal(do.cpu().numpy().astype(np.uint64), ho) and np.array_equal(dl.cpu().numpy().astype(np.uint32), hl)


def _order_of(keys):
    if len(keys) > 20_000_000:   # k <= 31: keys are below 2^62, signed order = unsigned order
        import torch
        return torch.from_numpy(keys.view(np.int64)).cuda().argsort().cpu().numpy()
    return np.argsort(keys, kind="stable")


def _run_and_compare(name, want, tmp_path):
    import torch
    import ray_amd
    cfg = CONFIGS[name]
    assert want["workload"] == name and want["config"] == cfg
    n = n_reads_for(cfg["genome"], cfg["read_len"], cfg["coverage"])
    assert n == want["reads"]
    packed, offsets, lengths = make_reads_device(cfg["genome"], cfg["read_len"], n, cfg["seed"], cfg["error"])
    b = ray_amd.GraphBuilder(cfg["k"], stream=torch.cuda.current_stream().cuda_stream)
    b.add_reads_device(packed.data_ptr(), n * ((cfg["read_len"] + 3) // 4), offsets.data_ptr(), lengths.data_ptr(), n, cfg["read_len"],
                       keep=(packed, offsets, lengths))
    b.count_kmers()
    st = b.stats()
    assert st["occurrences"] == want["occurrences"]
    assert b.size_() == want["size"]
    assert [st["bloom_bits"], st["bloom_set_bits"], st["bloom_insertions"]] == want["bloom"]
    cov, cnt = b.histogram()
    assert cov.tolist() == want["histogram"][0] and cnt.tolist() == want["histogram"][1], "coverage histogram"
    path = str(tmp_path / "CoverageDistribution.txt")
    assert list(ray_amd.coverage_distribution(cov, cnt, path)) == want["minimum_peak_repeat"]
    assert hashlib.sha256(open(path, "rb").read()).hexdigest() == want["sha256_CoverageDistribution_txt"]
    rec = b.export(sort=False)
    order = _order_of(np.ascontiguousarray(rec["key"][:, 0]))
    assert hashlib.sha256(rec[order].tobytes()).hexdigest() == want["sha256_sorted_records_after_pass1"], "records after pass 1"
    if "sha256_sorted_records_after_pass2" in want:
        b.add_edges()
        rec = b.export(sort=False)
        assert hashlib.sha256(rec[order].tobytes()).hexdigest() == want["sha256_sorted_records_after_pass2"], "records after pass 2"
        b.purge_edges()
        rec = b.export(sort=False)
        assert hashlib.sha256(rec[order].tobytes()).hexdigest() == want["sha256_sorted_records_after_purge"], "records after the purge"
        assert b.degree_distribution().reshape(-1).tolist() == want["degree_distribution"]
    b.close()
    return packed, offsets, lengths


def test_C1_whole_is_bit_exact(oracle, tmp_path):
    """E. coli-size genome, 50x, 100 bp, k = 31: 2.3 M reads, 161 M k-mer occurrences -- the designated CPU-reference config"""
    want = json.load(open(os.path.join(HERE, "golden", "full_size_C1.json")))
    packed, offsets, lengths = _run_and_compare("C1_ecoli_50x_100bp_k31", want, tmp_path)
    if os.environ.get("RG_LIVE_ORACLE"):
        g = oracle.Graph(31, nranks=1)
        g.set_reads(packed.cpu().numpy()[: len(lengths) * 25], offsets.cpu().numpy().astype(np.uint64), lengths.cpu().numpy().astype(np.uint32))
        g.count_kmers()
        assert hashlib.sha256(g.export().tobytes()).hexdigest() == want["sha256_sorted_records_after_pass1"]
        g.add_edges()
        g.purge_edges()
        assert hashlib.sha256(g.export().tobytes()).hexdigest() == want["sha256_sorted_records_after_purge"]


def test_C2_whole_is_bit_exact(tmp_path):
    """the configuration the metric is quoted on: 100 Mbp, 60x, 150 bp, k = 31 -- 40 M reads, 4.8e9 occurrences, 1e8 vertices"""
    path = os.path.join(HERE, "golden", "full_size_C2.json")
    if not os.path.exists(path):
        pytest.skip("tests/golden/full_size_C2.json has not been made (hours of oracle time)")
    _run_and_compare("C2_100Mbp_60x_150bp_k31", json.load(open(path)), tmp_path)
