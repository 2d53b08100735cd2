"""shared by the parity tests: run the oracle and the HIP path on the same reads and compare everything"""
import os

import numpy as np


def oracle_run(ro, packed, offsets, lengths, k, nranks=1, bloom_bits=None, W=None, purge=False):
    g = ro.Graph(k, nranks=nranks, bloom_bits=ro.BLOOM_DEFAULT if bloom_bits is None else bloom_bits, W=W)
    g.set_reads(packed, offsets, lengths)
    g.count_kmers()
    out = {"pass1": g.export(), "hist": g.histogram(), "size": g.size(), "occ": g.occurrences(),
           "bloom": [(g.bloom_bits(r), g.bloom_set_bits(r), g.bloom_insertions(r)) for r in range(nranks)]}
    g.add_edges()
    out["pass2"] = g.export()
    if purge:
        g.purge_edges()
        out["purged"] = g.export()
        out["degrees"] = g.degree_distribution()
        out["kmers_txt"] = g.kmers_txt()
    out["graph"] = g
    return out


def assert_records_equal(a, b, what=""):
    assert len(a) == len(b), "%s: %d vs %d records" % (what, len(a), len(b))
    assert np.array_equal(a["key"], b["key"]), what + ": retained k-mer sets differ"
    assert np.array_equal(a["coverage"], b["coverage"]), what + ": coverages differ"
    assert np.array_equal(a["edges"], b["edges"]), what + ": edge bitmaps differ"


def gpu_vs_oracle(ro, packed, offsets, lengths, k, bloom_bits=None, W=None, purge=True, batches=1, tmpdir=None, first_read_id=0, table_slots=0,
                  sweeps=(1 | 16 | (8 << 8) | 0x20000, 1 | 32 | (12 << 8), 1 | 16 | (19 << 8), 1 | 16 | (12 << 8) | 0x10000, 1 | 16 | (19 << 8) | 0x200000, 1 | 32 | (10 << 8) | 0x200000, 1 | 16 | (12 << 8) | 0x400000, 1 | 32 | (8 << 8) | 0x800000, 1 | 16 | 0x2000000, 1 | 32 | (9 << 8) | 0x4000000, 1 | 16 | (17 << 8) | 0x8000000, 1 | 32 | (12 << 8) | 0x8000000 | 0x10000, 1 | 16 | (12 << 8) | 0x40000000, 1 | 32 | 0x2000000 | 0x80000000, 1 | 16 | 8 | (13 << 8), 1 | 32 | 8 | (17 << 8) | 0x100000, 1 | 16 | 8 | 0x2000000, 1 | 32 | 0x80 | (12 << 8), 2 | 32, 2 | 16 | 0x20000)):
    """single-rank parity through the C ABI, once per sweep implementation: the partitioned sweep
    (super-k-mer records + LDS aggregation) with 2^8, 2^12 and 2^19 buckets = one, two and three partition
    levels (records cut by k_superkmers_lanes where it applies; flag 0x10000 forces k_superkmers; aggregated by a wavefront
    per bucket, k_aggregate, or with flag 0x200000 by a workgroup per bucket, k_aggregate_wg), and the direct sweep;
    flag 0x400000 builds the vertex table range by range from partitioned entries (k_build_ranges) however small it is, 0x800000 never
    (one table upsert per occurrence); flag 0x2000000 with no bucket count given sizes the partition from a sample of the records
    (the route big inputs take by themselves), flag 0x4000000 aggregates with the 512-slot tables of diverse inputs; flag 0x80000000 counts the sample's distinct k-mers with the nested sweep instead of the one-launch set; flag 8 folds the first partition level into the record kernel (k = 31: runs per workgroup instead of chunks, what big inputs get by themselves) and flag 0x80 forbids it; flag 0x40000000 cuts the records of k = 31 with the generic lane-per-read kernel (every other sweep here takes the kernel unrolled for its window of 17 m-mers); flag 0x8000000 keeps the 24-byte records where k <= 31 would take the 20-byte form (every other sweep here takes that one when k <= 31); flag 16 asks for the Bloom-region
    partitioning (it falls back by itself when the filter is too full), flag 32 for the dense first-touch
    stamps; flag 0x20000 makes the Bloom items straight into their first partition level even
    when the key array is sparse (k_bloom_scatter_items instead of k_bloom_items + one more k_scatter64 level);
    returns the oracle outputs"""
    want = oracle_run(ro, packed, offsets, lengths, k, 1, bloom_bits, W, purge)
    for sweep in sweeps:
        want["builder"] = _gpu_check(ro, want, packed, offsets, lengths, k, bloom_bits, W, purge, batches, tmpdir, sweep, first_read_id, table_slots)
    return want


def _gpu_check(ro, want, packed, offsets, lengths, k, bloom_bits, W, purge, batches, tmpdir, sweep, first_read_id=0, table_slots=0):
    """first_read_id: the global id of the first read (a constant added to every arrival stamp changes no result);
    table_slots: a minimum capacity of the vertex table (2^20 slots or more make a range-by-range build of two levels, the
    smallest the one-pass admission applies to)"""
    import ray_amd
    from ray_amd.graph import RG_BLOOM_DEFAULT
    b = ray_amd.GraphBuilder(k, bloom_bits=RG_BLOOM_DEFAULT if bloom_bits is None else bloom_bits,
                             kmer_words=W if W else 0, sweep=sweep, first_read_id=first_read_id, table_slots=table_slots)
    n = len(lengths)
    cuts = [n * i // batches for i in range(batches + 1)]
    for i in range(batches):
        lo, hi = cuts[i], cuts[i + 1]
        if hi > lo:
            b.add_reads(packed, offsets[lo:hi], lengths[lo:hi])
    b.count_kmers()
    st = b.stats()
    tag = "sweep 0x%x: " % sweep
    assert st["sweep_partitioned"] == (1 if (sweep & 3) == 1 else 0), tag + "wrong sweep ran"
    assert st["occurrences"] == want["occ"], tag + "occurrences"
    assert b.size_() == want["size"], tag + "size"
    assert_records_equal(b.export(), want["pass1"], tag + "after pass 1")
    cov, cnt = b.histogram()
    assert np.array_equal(cov, want["hist"][0]) and np.array_equal(cnt, want["hist"][1]), tag + "coverage histogram differs"
    assert (st["bloom_bits"], st["bloom_set_bits"], st["bloom_insertions"]) == want["bloom"][0], tag + "Bloom statistics"
    b.add_edges()
    assert_records_equal(b.export(), want["pass2"], tag + "after pass 2")
    if purge:
        b.purge_edges()
        assert_records_equal(b.export(), want["purged"], tag + "after purge")
        assert np.array_equal(b.degree_distribution(), want["degrees"])
    if tmpdir is not None and len(cov):
        p1, p2 = os.path.join(tmpdir, "gpu.txt"), os.path.join(tmpdir, "ora.txt")
        v1 = ray_amd.coverage_distribution(cov, cnt, p1)
        v2 = ro.coverage_distribution(want["hist"][0], want["hist"][1], p2)
        assert v1 == v2
        assert open(p1, "rb").read() == open(p2, "rb").read(), "CoverageDistribution.txt differs"
    return b
