"""Parity tests proper: the HIP path, called through the C ABI, against the oracle on the same seeded
reads.  Bit-exact bar: retained k-mer set, per-vertex coverage, edge bitmaps (before and after the
purge), coverage histogram, CoverageDistribution.txt bytes, Bloom statistics, degree table."""
import numpy as np
import pytest

from helpers import assert_records_equal, gpu_vs_oracle, oracle_run
from ray_amd.synth import make_reads_numpy

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k,read_len,bloom,error", [
    (31, 100, None, 0.01),    # the default filter: 16*k*reads bits
    (31, 100, 0, 0.01),       # -bloom-filter-bits 0: order-free mode, singletons kept
    (31, 100, 200000, 0.02),  # squeezed filter: false positives are common
    (31, 100, 4096, 0.02),    # saturated filter: nearly every first sight is a false positive
    (21, 60, None, 0.005),
    (63, 150, None, 0.01),    # two-word k-mers (MAXKMERLENGTH=64)
    (63, 150, 300000, 0.02),
    (33, 100, 0, 0.0),
    (47, 100, 100000, 0.01),
    (5, 30, 64, 0.0),         # tiny k: dense graph, every edge bit exercised
    (15, 40, None, 0.0),
])
def test_graph_matches_oracle(oracle, tmp_path, k, read_len, bloom, error):
    packed, offsets, lengths = make_reads_numpy(genome=20000, read_len=read_len, n_reads=6000, seed=k + int(bloom or 0),
                                                error=error)
    gpu_vs_oracle(oracle, packed, offsets, lengths, k, bloom_bits=bloom, tmpdir=str(tmp_path))


def test_long_records_fill_the_bucket_tables(oracle):
    """k = 63: records of up to 49 k-mers, expanded 64 k-mers per round, fill the 256-slot LDS table of a bucket within a
    few rounds; buckets that do not fit are dropped unwritten and redone in sub-passes -- nothing of a dropped attempt
    may stay in the log (a k-mer written on its own before the drop was once counted twice)"""
    packed, offsets, lengths = make_reads_numpy(genome=15000, read_len=150, n_reads=2500, seed=65, error=0.015)
    gpu_vs_oracle(oracle, packed, offsets, lengths, 63, bloom_bits=0, sweeps=(1 | 16 | (16 << 8), 1 | 16 | (8 << 8)))
    gpu_vs_oracle(oracle, packed, offsets, lengths, 63, bloom_bits=90000, sweeps=(1 | 16 | (16 << 8),))


def test_two_word_layout_with_short_k(oracle):
    """a MAXKMERLENGTH=64 build running -k 31 hashes with the two-word formula (Kmer.cpp:287-293)"""
    packed, offsets, lengths = make_reads_numpy(genome=8000, read_len=80, n_reads=2500, seed=5, error=0.01)
    gpu_vs_oracle(oracle, packed, offsets, lengths, 31, bloom_bits=60000, W=2)


def test_ragged_short_and_empty_reads(oracle, tmp_path):
    """reads shorter than k are skipped (KmerAcademyBuilder.cpp:130-134); lengths 1..90; odd byte offsets"""
    packed, offsets, lengths = make_reads_numpy(genome=6000, read_len=90, n_reads=5000, seed=9, error=0.01, ragged=True)
    assert (lengths < 31).any() and (lengths % 4 != 0).any()
    gpu_vs_oracle(oracle, packed, offsets, lengths, 31, bloom_bits=50000, tmpdir=str(tmp_path))
    gpu_vs_oracle(oracle, packed, offsets, lengths, 21, bloom_bits=None)


def test_reads_added_in_several_batches_and_table_growth(oracle):
    """rg_add_reads may be called repeatedly; read ids (hence arrival stamps) continue, the table grows"""
    packed, offsets, lengths = make_reads_numpy(genome=30000, read_len=100, n_reads=8000, seed=21, error=0.02)
    want = gpu_vs_oracle(oracle, packed, offsets, lengths, 31, bloom_bits=150000, batches=7)
    st = want["builder"].stats()
    assert st["table_slots"] * 0.7 >= st["stored_pairs"]   # load factor <= 0.7, whichever route built the table


def test_bloom_region_partitioning_is_used_and_exact(oracle):
    """a sparse filter (the regime the reference sizes for) must go through the LDS region path, not fall back"""
    import ray_amd
    packed, offsets, lengths = make_reads_numpy(genome=40000, read_len=100, n_reads=9000, seed=71, error=0.02)
    # 2^25 bits: 256 regions, one scatter level; 50 M and 2^28: two levels; 3 M is so dense that a region's
    # contested bits exceed the LDS rank table, so the library must notice and redo it with dense stamps; 12 M: regions
    # of more items than a workgroup holds in registers (several rounds, stamps kept in the spare buffer)
    for bits, partitioned in ((1 << 25, 1), (50_000_000, 1), (1 << 28, 1), (12_000_000, 1), (3_000_000, 0)):
        want = oracle_run(oracle, packed, offsets, lengths, 31, 1, bits)
        b = ray_amd.GraphBuilder(31, bloom_bits=bits, sweep=1 | 16)
        b.add_reads(packed, offsets, lengths)
        b.count_kmers()
        st = b.stats()
        assert st["bloom_partitioned"] == partitioned, bits
        assert (st["bloom_bits"], st["bloom_set_bits"], st["bloom_insertions"]) == want["bloom"][0]
        b.add_edges()
        assert_records_equal(b.export(), want["pass2"], "bloom bits %d" % bits)


def test_no_reads_and_no_kmers():
    import ray_amd
    b = ray_amd.GraphBuilder(31)
    b.count_kmers()
    assert b.size_() == 0 and len(b.export()) == 0
    cov, cnt = b.histogram()
    assert len(cov) == 0
    b.add_edges()
    b.purge_edges()
    with pytest.raises(ray_amd.RayGraphError):  # "Assembler panic: no k-mers found in reads"
        ray_amd.coverage_distribution(cov, cnt)
    packed, offsets, lengths = ray_amd.pack_reads(["ACGTACGT", "NNNN", "", "ACG"])
    b = ray_amd.GraphBuilder(31)
    b.add_reads(packed, offsets, lengths)
    b.count_kmers()
    assert b.size_() == 0 and b.stats()["occurrences"] == 0


def test_reference_edge_bitmap_example():
    """Vertex.h:84-115 through the GPU: ATGAC <- GATGA, -> TGACT / TGACA stores 0x94; GTCAT reads 0x29"""
    import ray_amd
    packed, offsets, lengths = ray_amd.pack_reads(["GATGACT", "GATGACA"])
    b = ray_amd.GraphBuilder(5, bloom_bits=0)
    b.add_reads(packed, offsets, lengths)
    b.count_kmers()
    b.add_edges()
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    word = lambda s: sum(code[c] << (2 * i) for i, c in enumerate(s))
    cov, edges = b.find(np.array([[word("ATGAC")], [word("GTCAT")], [word("AAAAA")]], dtype=np.uint64))
    assert cov.tolist() == [2, 2, 0]
    assert edges.tolist() == [0x94, 0x29, 0]


def test_phase_order_is_enforced():
    import ray_amd
    b = ray_amd.GraphBuilder(21)
    for call in (b.histogram, b.add_edges, b.purge_edges, b.size_, b.export):
        with pytest.raises(ray_amd.RayGraphError) as e:
            call()
        assert e.value.code == -4
    b.count_kmers()
    with pytest.raises(ray_amd.RayGraphError):
        b.count_kmers()
    with pytest.raises(ray_amd.RayGraphError):
        b.purge_edges()  # needs rg_add_edges first


def test_find_both_strands_and_kmers_txt(oracle, tmp_path):
    packed, offsets, lengths = make_reads_numpy(genome=3000, read_len=70, n_reads=1500, seed=33, error=0.01)
    want = gpu_vs_oracle(oracle, packed, offsets, lengths, 21, bloom_bits=30000)
    b, g = want["builder"], want["graph"]
    rec = want["purged"]
    rng = np.random.default_rng(1)
    pick = rec["key"][rng.integers(0, len(rec), 300)]
    queries = []
    for key in pick:
        words = (int(key[0]), 0)
        queries.append(words if rng.random() < 0.5 else oracle.complement_vertex(words, 21))
    queries += [(int(x), 0) for x in rng.integers(0, 1 << 42, 200)]  # mostly absent
    cov, edges = b.find(np.array([[q[0]] for q in queries], dtype=np.uint64))
    for q, c, e in zip(queries, cov, edges):
        found, wc, we = g.find(q)
        assert (int(c), int(e)) == ((wc, we) if found else (0, 0))
    from kmers_txt import body_lines
    got = sorted(body_lines(b.export(), 21))
    assert got == sorted(want["kmers_txt"].strip().split("\n"))


def test_plugin_mirror_runs_the_master_mode_sequence(oracle, tmp_path):
    import plugin_mirror
    packed, offsets, lengths = make_reads_numpy(genome=5000, read_len=80, n_reads=3000, seed=44, error=0.01)
    subgraph, (cov, cnt), values = plugin_mirror.build_graph(packed, offsets, lengths, k=22, prefix=str(tmp_path), write_kmers=True)
    want = oracle_run(oracle, packed, offsets, lengths, 21, purge=True)  # even k becomes k-1 (Parameters.cpp:892-907)
    assert subgraph.size() == want["size"]
    assert_records_equal(subgraph.export(), want["purged"], "plugin mirror")
    ref = str(tmp_path / "ref.txt")
    assert values == oracle.coverage_distribution(want["hist"][0], want["hist"][1], ref)
    assert open(str(tmp_path / "CoverageDistribution.txt"), "rb").read() == open(ref, "rb").read()
    lines = open(str(tmp_path / "kmers.txt")).read().strip().split("\n")
    assert lines[0] == "# Generated by Ray 2.3.2-devel" and len([l for l in lines if l.startswith("#")]) == 13
    assert sorted(l for l in lines if not l.startswith("#")) == sorted(want["kmers_txt"].strip().split("\n"))


def test_two_runs_give_identical_exports(oracle):
    import ray_amd
    packed, offsets, lengths = make_reads_numpy(genome=20000, read_len=100, n_reads=6000, seed=2, error=0.01)
    outs = []
    for _ in range(2):
        b = ray_amd.GraphBuilder(31)
        b.add_reads(packed, offsets, lengths)
        b.count_kmers()
        b.add_edges()
        outs.append(b.export(sort=True).tobytes())
    assert outs[0] == outs[1]


def test_long_reads_many_steps(oracle):
    """reads of thousands of bases: many 128-position steps per read in k_superkmers, records cut at step
    boundaries, 16-bit positions; plus reads of exactly k and k+1 bases"""
    packed, offsets, lengths = make_reads_numpy(genome=20000, read_len=6000, n_reads=120, seed=4, error=0.01, ragged=True)
    gpu_vs_oracle(oracle, packed, offsets, lengths, 31, bloom_bits=400000, sweeps=(1 | 16 | (9 << 8), 2 | 32))
    import ray_amd
    seqs = ["ACGTTGCATGCCGTAAGCTTGACCATGCAAT", "ACGTTGCATGCCGTAAGCTTGACCATGCAATG", "ACGTTGCATGCCGTAAGCTTGACCATGCAAT"[::-1],
            "A" * 300, "AC" * 200, "T" * 31, "ACGTTGCATGCCGTAAGCTTGACCATGCAATG"]
    p, o, l = ray_amd.pack_reads(seqs)
    gpu_vs_oracle(oracle, p, o, l, 31, bloom_bits=0, sweeps=(1 | (8 << 8), 2))
    gpu_vs_oracle(oracle, p, o, l, 31, bloom_bits=5000, sweeps=(1 | 16 | (8 << 8), 2 | 32))


def test_reads_up_to_the_lane_kernel_limit(oracle):
    """ragged reads of up to 1024 bases, the longest k_superkmers_lanes takes: 64 reads of very different lengths walk
    in lock step (lanes idle behind the longest), dozens of 16-base refills per read, record positions near 1000;
    the wavefront-per-read kernel cuts the same input for comparison (flag 0x10000)"""
    packed, offsets, lengths = make_reads_numpy(genome=30000, read_len=1024, n_reads=700, seed=12, error=0.01, ragged=True)
    assert lengths.max() > 1000 and (lengths < 31).any()
    gpu_vs_oracle(oracle, packed, offsets, lengths, 31, bloom_bits=300000, sweeps=(1 | 16 | (10 << 8), 1 | 16 | (10 << 8) | 0x10000))
    gpu_vs_oracle(oracle, packed, offsets, lengths, 21, bloom_bits=0, sweeps=(1 | (9 << 8),))
    gpu_vs_oracle(oracle, packed, offsets, lengths, 37, bloom_bits=None, sweeps=(1 | 16 | (9 << 8),))   # two-word keys, w = 23


def test_log_overflow_is_redone_with_a_full_log(oracle):
    """inputs of little redundancy overflow the log of distinct k-mers (sized for a quarter of the occurrences): the
    aggregation then runs again with one slot per occurrence; flag 64 makes the first log one chunk long"""
    packed, offsets, lengths = make_reads_numpy(genome=200000, read_len=100, n_reads=3000, seed=8, error=0.02)
    want = gpu_vs_oracle(oracle, packed, offsets, lengths, 31, bloom_bits=None, sweeps=(1 | 16 | 64 | (10 << 8), 1 | 16 | (10 << 8)))
    assert want["builder"].stats()["log_entries"] > 512


def test_log_sized_from_a_sample_of_the_buckets(oracle):
    """before the first aggregation a sample of the buckets is aggregated alone and the log sized from it (flag
    0x40000 runs that on small inputs and always resizes): same result with an input of little redundancy, where the
    projection is what the log holds, with a redundant one, and with buckets that overflow the LDS table (few
    buckets, k = 63), whose second launch is part of the sample"""
    packed, offsets, lengths = make_reads_numpy(genome=200000, read_len=100, n_reads=3000, seed=8, error=0.02)
    gpu_vs_oracle(oracle, packed, offsets, lengths, 31, bloom_bits=None, sweeps=(1 | 16 | 0x40000 | (10 << 8), 1 | 32 | 0x40000 | (12 << 8)))
    packed, offsets, lengths = make_reads_numpy(genome=4000, read_len=100, n_reads=4000, seed=9, error=0.0)
    gpu_vs_oracle(oracle, packed, offsets, lengths, 31, bloom_bits=None, sweeps=(1 | 16 | 0x40000 | (8 << 8),))
    packed, offsets, lengths = make_reads_numpy(genome=60000, read_len=150, n_reads=4000, seed=10, error=0.01)
    gpu_vs_oracle(oracle, packed, offsets, lengths, 63, bloom_bits=None, sweeps=(1 | 16 | 0x40000 | (4 << 8),))


def test_level_histograms_from_digit_bytes_and_from_records(oracle):
    """a partition level's scatter leaves the next level's digit of every record in a byte array, which that level's
    histogram reads instead of the records; flag 0x100000 keeps the histograms on the records: same buckets, same
    result, with two and three levels, one- and two-word keys, ragged reads (segments that end inside a tile)"""
    packed, offsets, lengths = make_reads_numpy(genome=30000, read_len=120, n_reads=6000, seed=14, error=0.01, ragged=True)
    gpu_vs_oracle(oracle, packed, offsets, lengths, 31, bloom_bits=None,
                  sweeps=(1 | 16 | (19 << 8), 1 | 16 | (19 << 8) | 0x100000, 1 | 16 | (13 << 8), 1 | 16 | (13 << 8) | 0x100000))
    gpu_vs_oracle(oracle, packed, offsets, lengths, 63, bloom_bits=200000, sweeps=(1 | 16 | (18 << 8), 1 | 16 | (18 << 8) | 0x100000))


@pytest.mark.parametrize("k,log2_bits", [(31, 12), (31, 16), (31, 19), (21, 22), (31, 27), (63, 18), (63, 32), (31, 33)])
def test_power_of_two_filters_by_key_partitions(oracle, k, log2_bits):
    """rg_bloomx.h: one item per key partitioned by the key's origin, eight streams per region -- from a filter far
    smaller than the key set (every late key a false positive) to the reference's cap of 2^33 bits; zero, one and two
    partition levels below the one that makes the items; log and vertex table as the key array"""
    packed, offsets, lengths = make_reads_numpy(genome=20000, read_len=150 if k < 63 else 200, n_reads=6000, seed=90 + log2_bits, error=0.02)
    want = gpu_vs_oracle(oracle, packed, offsets, lengths, k, bloom_bits=1 << log2_bits, purge=False,
                         sweeps=(1 | (10 << 8) | 0x1000000, 1 | (9 << 8) | 0x1000000 | 0x400000, 2 | 0x1000000))
    assert want["builder"].stats()["bloom_partitioned"] == 2


@pytest.mark.parametrize("k,log2_bits", [(31, 28), (31, 33), (21, 30), (63, 32)])
def test_admission_items_made_by_the_aggregation(oracle, k, log2_bits):
    """one rank, a filter of 2^28 bits or more: k_aggregate makes the admission's items while it writes the log (runs per
    workgroup by the top 5 bits of the key's origin) and the key partition starts from them -- with the flag that forbids
    it the same result comes from k_xor_keys; one- and two-word keys, with and without the folded record level"""
    # (k = 63 makes fewer records: more reads, so that the record sample -- 1 / 4096 of the buckets -- is not empty)
    packed, offsets, lengths = make_reads_numpy(genome=25000 if k < 63 else 120000, read_len=150 if k < 63 else 200, n_reads=7000 if k < 63 else 40000, seed=190 + log2_bits, error=0.02)
    base = 1 | 0x2000000 | 0x1000000 | 0x400000
    want = gpu_vs_oracle(oracle, packed, offsets, lengths, k, bloom_bits=1 << log2_bits, sweeps=(base, base | 8))
    st = want["builder"].stats()
    assert st["bloom_partitioned"] == 2 and st["items_from_sweep"] == 1
    # one-word keys: the log beside the items is written in 16-byte entries; a table this small is not built by the one-pass
    # admission (one partition level), so the log is expanded once the admission is decided
    assert st["compact_log"] == (2 if k <= 32 else 0)
    # ... a table of 2^21 slots is: the 16-byte entries are read as they are
    want = gpu_vs_oracle(oracle, packed, offsets, lengths, k, bloom_bits=1 << log2_bits, sweeps=(base, base | 8), table_slots=1 << 21)
    st = want["builder"].stats()
    assert st["items_from_sweep"] == 1 and st["compact_log"] == (1 if k <= 32 else 0) and st["table_slots"] == 1 << 21
    want = gpu_vs_oracle(oracle, packed, offsets, lengths, k, bloom_bits=1 << log2_bits, sweeps=(base | 4,), table_slots=1 << 21)
    st = want["builder"].stats()
    assert st["items_from_sweep"] == 0 and st["compact_log"] == 0
    want = gpu_vs_oracle(oracle, packed, offsets, lengths, k, bloom_bits=1 << log2_bits, sweeps=(base | 0x8000000,), table_slots=1 << 21)
    st = want["builder"].stats()
    assert st["items_from_sweep"] == 1 and st["compact_log"] == 0


@pytest.mark.parametrize("k", [21, 31])
@pytest.mark.parametrize("route,flags,expanded", [
    ("key partition from the items, admission in place, classic insert", 0, True),
    ("admission decided, the table too small for the one-pass admission: counting pass and exact ranges", 0x400000, True),
    ("never range by range", 0x800000, True),
    ("Bloom regions of (bit, key) items", 16, True),
    ("dense first-touch stamps", 32, True),
    ("key partitions from a histogram pass", 0x10000000, True),
])
def test_a_compact_log_is_expanded_for_the_routes_that_read_full_entries(oracle, k, route, flags, expanded):
    """one-word keys with items made by the aggregation: the log is written in 16-byte entries; every route but the key
    partition that starts from the items + the one-pass admission gets it expanded first -- with the arrival stamps taken
    back from the items where the admission is still to be decided"""
    packed, offsets, lengths = make_reads_numpy(genome=25000, read_len=150, n_reads=7000, seed=77 + k, error=0.02)
    base = 1 | 0x2000000 | 0x1000000
    want = gpu_vs_oracle(oracle, packed, offsets, lengths, k, bloom_bits=1 << 28, sweeps=(base | flags,))
    st = want["builder"].stats()
    assert st["compact_log"] == (2 if expanded else 1), route


def test_a_compact_log_without_the_key_partition(oracle):
    """items and a 16-byte log were made, but the admission does not take the key partition (too few keys and nothing forces
    it): the log is expanded with its stamps before the Bloom regions or the dense stamps read it"""
    packed, offsets, lengths = make_reads_numpy(genome=25000, read_len=150, n_reads=7000, seed=91, error=0.02)
    for flags in (0, 16, 32):
        want = gpu_vs_oracle(oracle, packed, offsets, lengths, 31, bloom_bits=1 << 28, sweeps=(1 | 0x2000000 | flags,))
        st = want["builder"].stats()
        assert st["compact_log"] == 2 and st["bloom_partitioned"] != 2, flags


@pytest.mark.parametrize("k", [21, 31, 47, 63])
def test_hashes_are_checked_against_the_callers(oracle, k):
    """rg_check_hashes / rg_hash_kmers: the device's hash_function_1 / _2 against the caller's own (here the oracle's, the
    stand-in for Ray's Kmer::hash_function_1/2); a caller whose hash differs in one bit gets RG_ERR_HASH"""
    import ray_amd
    from ray_amd._lib import RG_ERR_HASH
    b = ray_amd.GraphBuilder(k)
    W = b.kmer_words
    b.check_hashes(lambda w: oracle.hash1(w, W), lambda w: oracle.hash2(w, W), n_samples=512)
    with pytest.raises(ray_amd.RayGraphError) as err:
        b.check_hashes(lambda w: oracle.hash1(w, W) ^ (1 << 40), lambda w: oracle.hash2(w, W), n_samples=64)
    assert err.value.code == RG_ERR_HASH and "hash_function_1" in str(err.value)
    with pytest.raises(ray_amd.RayGraphError) as err:
        b.check_hashes(lambda w: oracle.hash1(w, W), lambda w: oracle.hash2(w, W) + (1 if w[0] == 1 else 0), n_samples=64)
    assert err.value.code == RG_ERR_HASH and "hash_function_2" in str(err.value)
    rng = np.random.default_rng(k)
    kmers = rng.integers(0, 1 << 62, size=(100, W), dtype=np.uint64)
    if W == 2:
        kmers[:, 1] &= np.uint64((1 << (2 * k - 64)) - 1)
    else:
        kmers &= np.uint64((1 << (2 * k)) - 1)
    h1, h2 = b.hash_kmers(kmers)
    assert [int(x) for x in h1] == [oracle.hash1(tuple(int(v) for v in row), W) for row in kmers]
    assert [int(x) for x in h2] == [oracle.hash2(tuple(int(v) for v in row), W) for row in kmers]
    # the survey's known answers through the device code
    if k == 31:
        h1, h2 = b.hash_kmers(np.array([[0x24e4e4e4e4e4e4e4]], dtype=np.uint64))
        assert (int(h1[0]), int(h2[0])) == (0x14605c73bf1f7ba9, 0x2a0e90af81ed09a8)
    b.close()


def test_export_stream_hands_over_the_same_records_chunk_by_chunk(oracle):
    """rg_export_stream against rg_export: same bytes in the same order, whatever the chunk size; the consumer can stop it"""
    import ray_amd
    packed, offsets, lengths = make_reads_numpy(genome=30000, read_len=100, n_reads=12000, seed=8, error=0.01)
    for k in (31, 63):
        b = ray_amd.GraphBuilder(k)
        b.add_reads(packed, offsets, lengths)
        b.count_kmers()
        b.add_edges()
        whole = b.export(sort=False)
        assert len(whole) > 20000
        for chunk in (0, 1000, 4097, len(whole), 10 * len(whole)):
            got = []
            b.export_stream(lambda rec: got.append(rec.copy()), chunk_records=chunk)   # the staging memory is only valid during the call
            assert len(got) == (1 if chunk == 0 or chunk >= len(whole) else -(-len(whole) // chunk))
            assert np.concatenate(got).tobytes() == whole.tobytes(), "k=%d chunk=%d" % (k, chunk)
        seen = []
        with pytest.raises(ray_amd.RayGraphError) as e:
            b.export_stream(lambda rec: seen.append(len(rec)) or (len(seen) == 3), chunk_records=500)
        assert e.value.code == -4 and len(seen) == 3
        b.close()
