"""BASELINE.json's five configurations as parity cases: each at a scale the oracle finishes in seconds, bit-exact
against it (single rank for the one-GPU configurations, the R-rank exchange for the 8-GPU ones), and the bench
configuration (C2) once at its full size through properties that need no oracle: totals of the coverage histogram,
reciprocity of the edge bitmaps after the purge, and agreement between two independent routes through the library
(one rank; two ranks exchanging records, then entries)."""
import numpy as np
import pytest

from helpers import gpu_vs_oracle, oracle_run
from test_multirank_gpu import check_ranks, purge_ranks, run_ranks
from ray_amd.exchange import partition_reads
from ray_amd.synth import CONFIGS, make_reads_numpy, n_reads_for

pytestmark = pytest.mark.gpu


def scaled(name, genome):
    c = dict(CONFIGS[name])
    c["n_reads"] = n_reads_for(genome, c["read_len"], c["coverage"])
    c["genome"] = genome
    return c


def test_c1_ecoli_shape_single_rank(oracle, tmp_path):
    """configs[0]: -k 31, 100 bp reads at 50x (E. coli scaled 1:50); default filter of 16*k*reads bits"""
    c = scaled("C1_ecoli_50x_100bp_k31", 92_000)
    reads = make_reads_numpy(c["genome"], c["read_len"], c["n_reads"], c["seed"], c["error"])
    want = gpu_vs_oracle(oracle, *reads, c["k"], tmpdir=str(tmp_path), sweeps=(1 | 16 | (14 << 8), 2 | 32))
    cov, cnt = want["hist"]
    assert 30 <= int(cov[np.argmax(cnt)]) <= 40  # the peak sits at 50 * (100 - 31 + 1) / 100 = 35


def test_c2_shape_single_rank_with_errors(oracle, tmp_path):
    """configs[1] scaled 1:1000, once as the bench runs it (error-free) and once with 0.5 % substitutions"""
    c = scaled("C2_100Mbp_60x_150bp_k31", 100_000)
    for error in (0.0, 0.005):
        reads = make_reads_numpy(c["genome"], c["read_len"], c["n_reads"], c["seed"], error)
        gpu_vs_oracle(oracle, *reads, c["k"], tmpdir=str(tmp_path), sweeps=(1 | 16 | (15 << 8),))


@pytest.mark.parametrize("name,genome,nranks", [("C3_500Mbp_40x_150bp_k63", 60_000, 8), ("C5_3.1Gbp_30x_150bp_k31_1pct", 80_000, 8)])
def test_eight_rank_configs(oracle, name, genome, nranks):
    """configs[2] (two-word k-mers, k = 63) and configs[4] (1 % errors): eight ranks on one device, both exchange routes,
    then the purge across ranks"""
    c = scaled(name, genome)
    packed, offsets, lengths = make_reads_numpy(c["genome"], c["read_len"], c["n_reads"], c["seed"], c["error"])
    want = oracle_run(oracle, packed, offsets, lengths, c["k"], nranks, None)
    for sweep, route in ((1 | 16 | (12 << 8), "records"), (1 | 16 | (12 << 8), "entries")):
        builders = run_ranks(packed, offsets, lengths, c["k"], nranks, None, sweep=sweep, route=route)
        check_ranks(builders, want, route)
    want["graph"].purge_edges()
    purge_ranks(builders)
    for r, b in enumerate(builders):
        got, ref = b.export(), want["graph"].export(r)
        assert np.array_equal(got["key"], ref["key"]) and np.array_equal(got["coverage"], ref["coverage"])
        assert np.array_equal(got["edges"], ref["edges"]), "rank %d after purge" % r


def test_c4_metagenome(oracle, tmp_path):
    """configs[3]: 200 species, log-normal abundance (synth.CONFIGS' C4, genomes scaled 1:5000 to 400-1200 bases): coverage from
    below 1x to hundreds, so singletons, false positives and deep vertices all occur in one run"""
    from ray_amd.synth import metagenome_numpy
    c = CONFIGS["C4_meta_200sp_lognormal_30x_150bp_k31"]
    genomes, n_reads = metagenome_numpy(c, 1 / 5000)
    assert len(genomes) == 200 and 25_000 <= n_reads <= 40_000
    reads = make_reads_numpy(0, c["read_len"], n_reads, seed=40, error=0.005, genomes=genomes)
    want = gpu_vs_oracle(oracle, *reads, c["k"], tmpdir=str(tmp_path), sweeps=(1 | 16 | (13 << 8), 2 | 32))
    cov = want["pass1"]["coverage"]
    assert cov.min() <= 2 and cov.max() >= 50


def test_c4_reads_made_on_the_device(oracle):
    """the bench's generator for C4 (genomes back to back in HBM, species drawn by coverage x length): parity on what it makes"""
    import torch
    from ray_amd.synth import make_reads_torch, metagenome_plan
    c = CONFIGS["C4_meta_200sp_lognormal_30x_150bp_k31"]
    lengths, coverage, n_reads = metagenome_plan(c, 1 / 5000)
    packed, offsets, rl = make_reads_torch(0, c["read_len"], n_reads, c["seed"], 0.003, species=(lengths, coverage))
    torch.cuda.synchronize()
    reads = (packed.cpu().numpy(), offsets.cpu().numpy().astype(np.uint64), rl.cpu().numpy().astype(np.uint32))
    want = gpu_vs_oracle(oracle, *reads, c["k"], sweeps=(1 | 16 | (12 << 8),))
    cov = want["pass1"]["coverage"]
    assert cov.min() <= 2 and cov.max() >= 50


def _device_reads(c, n_reads, read_seed=None):
    from ray_amd.synth import make_reads_torch
    return make_reads_torch(c["genome"], c["read_len"], n_reads, c["seed"], c["error"], read_seed=read_seed)


def _neighbours(keys, edges, k):
    """numpy restatement of Kmer::_getOutgoingEdges for one-word keys (Kmer.cpp:208-241): for every set bit
    4+c the child key, and the bit the child holds for the parent on its ingoing side"""
    out = []
    mask = np.uint64((1 << (2 * k)) - 1)
    for c in range(4):
        sel = (edges >> np.uint8(4 + c)) & np.uint8(1) == 1
        parent = keys[sel]
        child = ((parent >> np.uint64(2)) | (np.uint64(c) << np.uint64(2 * (k - 1)))) & mask
        back = (parent & np.uint64(3)).astype(np.uint8)  # the base the child sees on its ingoing side
        out.append((child, back))
    return out


def test_c2_full_size_properties():
    """configs[1] at full size (100 Mbp, 60x, 150 bp, k = 31: 40 M reads, 4.8 G k-mer occurrences)"""
    import torch
    import ray_amd
    c = dict(CONFIGS["C2_100Mbp_60x_150bp_k31"])
    n_reads = n_reads_for(c["genome"], c["read_len"], c["coverage"])
    k = c["k"]
    packed, offsets, lengths = _device_reads(c, n_reads)
    torch.cuda.synchronize()

    def build(bloom):
        b = ray_amd.GraphBuilder(k, bloom_bits=bloom)
        b.add_reads_device(packed.data_ptr(), packed.numel(), offsets.data_ptr(), lengths.data_ptr(), n_reads, c["read_len"])
        b.count_kmers()
        return b

    # without the filter every occurrence is counted: the histogram's first moment is the number of occurrences
    b0 = build(0)
    st0 = b0.stats()
    occ = n_reads * (c["read_len"] - k + 1)
    assert st0["occurrences"] == occ and st0["sweep_partitioned"] == 1
    cov0, cnt0 = b0.histogram()
    assert int((cov0.astype(np.uint64) * cnt0).sum()) == 2 * occ  # both strands of every pair are visited
    assert int(cnt0.sum()) == b0.size_() == 2 * st0["stored_pairs"]
    assert st0["stored_pairs"] <= c["genome"] - k + 1  # error-free reads hold no k-mer the genome lacks
    assert 40 <= int(cov0[np.argmax(cnt0)]) <= 56      # 60 * (150 - 31 + 1) / 150 = 48

    # with the default filter: a k-mer seen N >= 2 times is stored with N, or N + 1 if its first sight was a false
    # positive; one seen once is stored (with 2) only as a false positive
    from ray_amd.graph import RG_BLOOM_DEFAULT
    b1 = build(RG_BLOOM_DEFAULT)
    st1 = b1.stats()
    assert st1["bloom_partitioned"] == 2   # the default filter of C2 is the reference's cap of 2^33 bits: resolved by key partitions
    rec1 = b1.export(sort=False)
    rng = np.random.default_rng(0)
    pick = rec1[rng.integers(0, len(rec1), 2_000_000)]
    cov_plain, _ = b0.find(pick["key"])
    d = pick["coverage"].astype(np.int64) - cov_plain.astype(np.int64)
    assert ((d == 0) | (d == 1)).all() and (pick["coverage"] >= 2).all()
    cov1, cnt1 = b1.histogram()
    singles = int(cnt0[cov0 == 1].sum()) // 2
    assert st0["stored_pairs"] - singles <= st1["stored_pairs"] <= st0["stored_pairs"]
    # first moment with the filter = occurrences - dropped singletons + false positives: that gives the number of
    # false positives, which the sample must agree with
    dropped = st0["stored_pairs"] - st1["stored_pairs"]
    fp = int((cov1.astype(np.uint64) * cnt1).sum()) // 2 - occ + dropped
    assert 0 <= fp <= st0["stored_pairs"] // 50  # a small share of the candidates at this filter fill
    assert st1["bloom_insertions"] == st0["stored_pairs"] - fp  # a first sight enters the filter unless it "was there"
    p_fp = fp / st1["stored_pairs"]
    assert abs((d == 1).mean() - p_fp) < 6 * np.sqrt(max(p_fp, 1e-9) / len(d)) + 1e-6
    del b0

    # pass 2 and the purge: every outgoing edge that survives points at a stored vertex that holds the ingoing edge back
    b1.add_edges()
    b1.purge_edges()
    rec = b1.export(sort=False)
    sample = rec[rng.integers(0, len(rec), 1_000_000)]
    keys = sample["key"][:, 0]
    for child, back in _neighbours(keys, sample["edges"], k):
        ccov, cedges = b1.find(child.reshape(-1, 1))
        assert (ccov > 0).all()
        assert (((cedges >> back) & 1) == 1).all()
    deg = b1.degree_distribution()
    assert int(deg.sum()) == b1.size_()

    # a second, independent route to the same graph: two ranks exchanging records, then entries, no filter;
    # together they must hold what one rank held (same pairs, same first moment)
    del b1, rec, rec1, sample, pick
    builders = []
    for r in range(2):
        first, last = partition_reads(n_reads, r, 2)
        bpr = (c["read_len"] + 3) // 4
        b = ray_amd.GraphBuilder(k, rank=r, size=2, bloom_bits=0, first_read_id=first)
        b.add_reads_device(packed.data_ptr() + first * bpr, (last - first) * bpr, offsets.data_ptr(), lengths.data_ptr(),
                           last - first, c["read_len"])
        builders.append(b)
    sent = []
    for b in builders:
        b.records_local()
        counts, unit = b.record_counts()
        buf = torch.empty(max(1, sum(counts)) * unit, dtype=torch.uint8, device="cuda")
        b.export_records(buf.data_ptr(), max(1, sum(counts)))
        sent.append((counts, buf, unit))
    for dst, b in enumerate(builders):
        recv = torch.cat([buf[sum(cn[:dst]) * u:(sum(cn[:dst]) + cn[dst]) * u] for cn, buf, u in sent])
        torch.cuda.synchronize()
        b.count_records(recv.data_ptr(), recv.numel() // sent[0][2])
        del recv
    del sent
    from ray_amd.graph import RG_ENTRY_BYTES
    sends = []
    for b in builders:
        counts = b.owner_counts()
        buf = torch.empty(max(1, sum(counts)) * RG_ENTRY_BYTES, dtype=torch.uint8, device="cuda")
        b.export_owner_buckets(buf.data_ptr(), max(1, sum(counts)))
        sends.append((counts, buf))
    torch.cuda.synchronize()
    for dst, b in enumerate(builders):
        for counts, buf in sends:
            start = sum(counts[:dst])
            b.merge_entries(buf[start * RG_ENTRY_BYTES:].data_ptr(), counts[dst])
    del sends
    pairs, moment = 0, 0
    for b in builders:
        b.resolve_admission()
        cv, cn = b.histogram()
        pairs += b.stats()["stored_pairs"]
        moment += int((cv.astype(np.uint64) * cn).sum())
    assert pairs == st0["stored_pairs"] and moment == 2 * occ
