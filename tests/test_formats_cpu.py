"""On-disk formats of the path: the product's host-side writers against the oracle's
restatement of the reference writers, byte for byte, and against a reader that follows the reference's
own reader.  No GPU: the writers take rg_export-format records, here produced by the oracle."""
import struct

import numpy as np
import pytest

from ray_amd.synth import make_reads_numpy


def read_genome_graph(path, W):
    """MessageProcessor.cpp:596-655: LargeCount n, then per entry key, int coverage, int parents + keys,
    int children + keys"""
    data = open(path, "rb").read()
    pos = 0
    (n,) = struct.unpack_from("<Q", data, pos)
    pos += 8
    out = []
    for _ in range(n):
        key = struct.unpack_from("<%dQ" % W, data, pos)
        pos += 8 * W
        (cov, parents) = struct.unpack_from("<ii", data, pos)
        pos += 8
        pk = [struct.unpack_from("<%dQ" % W, data, pos + 8 * W * j) for j in range(parents)]
        pos += 8 * W * parents
        (children,) = struct.unpack_from("<i", data, pos)
        pos += 4
        ck = [struct.unpack_from("<%dQ" % W, data, pos + 8 * W * j) for j in range(children)]
        pos += 8 * W * children
        out.append((key, cov, pk, ck))
    assert pos == len(data)
    return out


@pytest.mark.parametrize("k,W", [(21, 1), (31, 1), (31, 2), (45, 2), (63, 2)])
def test_genome_graph_checkpoint(oracle, tmp_path, k, W):
    import ray_amd
    packed, offsets, lengths = make_reads_numpy(genome=3000, read_len=90, n_reads=600, seed=5 + k, error=0.02)
    g = oracle.Graph(k, nranks=2, bloom_bits=30000, W=W)
    g.set_reads(packed, offsets, lengths)
    g.count_kmers()
    g.add_edges()
    g.purge_edges()
    for r in range(2):
        want, got = str(tmp_path / ("want%d" % r)), str(tmp_path / ("got%d" % r))
        g.write_genome_graph(r, want)
        rec = g.export(r)
        assert len(rec) > 50
        ray_amd.write_genome_graph(rec, k, got)
        assert open(got, "rb").read() == open(want, "rb").read()
        # what the reference's reader rebuilds from it: both strands of every vertex, edges as neighbour k-mers
        entries = read_genome_graph(got, W)
        assert len(entries) == g.size(r) == 2 * len(rec)
        for i, v in enumerate(rec):
            key, cov, parents, children = entries[2 * i]
            assert tuple(int(x) for x in v["key"]) == key and cov == int(v["coverage"])
            e = int(v["edges"])
            assert len(parents) == bin(e & 15).count("1") and len(children) == bin(e >> 4).count("1")
            for nb in parents + children:   # after the purge every neighbour named is a vertex of the graph
                assert g.find(list(nb) + [0] * (2 - W))[0]
            rkey = entries[2 * i + 1][0]
            assert list(rkey) == [int(x) for x in oracle.complement_vertex(list(key) + [0] * (2 - W), k)][:W]


def test_coverage_checkpoint_and_partition_file(oracle, tmp_path):
    import ray_amd
    packed, offsets, lengths = make_reads_numpy(genome=5000, read_len=100, n_reads=1500, seed=3, error=0.01)
    g = oracle.Graph(31, nranks=3, bloom_bits=50000)
    g.set_reads(packed, offsets, lengths)
    g.count_kmers()
    cov, cnt = g.histogram()
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    oracle.write_coverage_checkpoint(cov, cnt, a)
    ray_amd.write_coverage_checkpoint(cov, cnt, b)
    data = open(b, "rb").read()
    assert data == open(a, "rb").read()
    # MachineHelper.cpp:226-241 reads int n, then (int coverage, LargeCount count) * n
    assert struct.unpack_from("<i", data)[0] == len(cov) and len(data) == 4 + 12 * len(cov)
    assert struct.unpack_from("<iQ", data, 4) == (int(cov[0]), int(cnt[0]))
    g.write_partition(a)
    ray_amd.write_graph_partition([g.size(r) for r in range(3)], b)
    text = open(b).read()
    assert text == open(a).read()
    assert text.startswith("#Rank\tNumberOfKmers\tIdealNumberOfKmers\tDifference\tRelativeDifference\n#TotalKmers: %d\n#Ranks: 3\n" % g.size())
    # one rank, nothing stored: 0 of 0 is 0% (EdgePurger.cpp:205-208 skips the division)
    ray_amd.write_graph_partition([0], b)
    assert open(b).read().splitlines()[-1] == "0\t0\t0\t0\t0%"


def test_sequence_file_counts(tmp_path):
    """FastaLoader.cpp:26-64 / FastqLoader.cpp:84-119 record counting (host-side part of the ingest)"""
    import ray_amd
    fa, fq = tmp_path / "x.fasta", tmp_path / "x.fastq"
    fa.write_text(">r1 some text\nACGT\nACGT\n>r2\nNNNN\n>r3\nTTTT")
    fq.write_text("@r1\nACGT\n+\nIIII\n@r2\nAC\n+\nII\n")
    assert ray_amd.count_sequences(str(fa)) == 3 and ray_amd.count_sequences(str(fq)) == 2
    with pytest.raises(ray_amd.RayGraphError):
        ray_amd.count_sequences(str(tmp_path / "x.sff"))


@pytest.mark.parametrize("k,W", [(21, 1), (31, 2), (63, 2)])
def test_genome_graph_checkpoint_round_trip(oracle, tmp_path, k, W):
    """the written checkpoint through two readers: the oracle's restatement of the reference's reader
    (MessageProcessor.cpp:596-655, with its parents / children / size assertions :633-641,650) rebuilds the graph it was
    written from, and the product's reader (rg_read_genome_graph) returns the same records"""
    import ray_amd
    packed, offsets, lengths = make_reads_numpy(genome=4000, read_len=90, n_reads=900, seed=50 + k, error=0.02)
    g = oracle.Graph(k, nranks=1, bloom_bits=40000, W=W)
    g.set_reads(packed, offsets, lengths)
    g.count_kmers()
    g.add_edges()
    g.purge_edges()
    rec = g.export()
    path = str(tmp_path / "Rank0.Checkpoint.GenomeGraph.ray")
    ray_amd.write_genome_graph(rec, k, path)
    # the reference's reader, restated
    g2 = oracle.Graph(k, nranks=1, bloom_bits=0, W=W)
    assert g2.read_genome_graph(path) == 0, "the reader's assertions (MessageProcessor.cpp:633-641,650) fail on the written file"
    back = g2.export()
    assert g2.size() == g.size() and back.tobytes() == rec.tobytes()
    # the product's reader: same records, in file order (the writer's = sorted here)
    got = ray_amd.read_genome_graph(path, k, W)
    assert got.tobytes() == rec.tobytes()
    # entries in another order, reverse-complement strand first: the reference takes the coverage from the entry that
    # creates the vertex only when that entry is the lower key (Vertex::setCoverage, Vertex.cpp:80-92) -- both readers agree
    entries = read_genome_graph(path, W)
    swapped = str(tmp_path / "swapped.ray")
    with open(swapped, "wb") as f:
        f.write(struct.pack("<Q", len(entries)))
        order = list(range(0, len(entries), 2))[::-1]
        for i in order:
            for key, cov, parents, children in (entries[i + 1], entries[i]) if i % 4 == 0 else (entries[i], entries[i + 1]):
                f.write(struct.pack("<%dQ" % W, *key) + struct.pack("<ii", cov, len(parents)))
                for nb in parents:
                    f.write(struct.pack("<%dQ" % W, *nb))
                f.write(struct.pack("<i", len(children)))
                for nb in children:
                    f.write(struct.pack("<%dQ" % W, *nb))
    g3 = oracle.Graph(k, nranks=1, bloom_bits=0, W=W)
    assert g3.read_genome_graph(swapped) == 0
    want3 = g3.export()
    got3 = ray_amd.read_genome_graph(swapped, k, W)
    got3 = got3[np.lexsort(tuple(got3["key"][:, j] for j in reversed(range(W))))]
    assert got3.tobytes() == want3.tobytes()
    assert (want3["coverage"] == 0).sum() > 0 and (want3["coverage"] > 0).sum() > 0   # both cases occurred
    assert np.array_equal(want3["edges"], rec["edges"])
    # a corrupted file: a parent that is not a neighbour, a wrong count
    data = bytearray(open(path, "rb").read())
    first_with_parent = next(i for i, e in enumerate(entries) if e[2])
    pos = 8
    for e in entries[:first_with_parent]:
        pos += 8 * W + 8 + 8 * W * len(e[2]) + 4 + 8 * W * len(e[3])
    pos += 8 * W + 8     # first parent of that entry
    data[pos + 1] ^= 0x40
    bad = str(tmp_path / "bad.ray")
    open(bad, "wb").write(bytes(data))
    with pytest.raises(ray_amd.RayGraphError):
        ray_amd.read_genome_graph(bad, k, W)
    data = bytearray(open(path, "rb").read())
    data[0:8] = struct.pack("<Q", len(entries) + 2)
    open(bad, "wb").write(bytes(data))
    with pytest.raises(ray_amd.RayGraphError):
        ray_amd.read_genome_graph(bad, k, W)


def test_coverage_distribution_analysis_file(oracle, tmp_path):
    """CoverageDistributionAnalysis.txt (MachineHelper.cpp:283-327): the product's writer against the oracle's, byte for byte"""
    import ray_amd
    rng = np.random.default_rng(7)
    cases = []
    packed, offsets, lengths = make_reads_numpy(genome=20000, read_len=100, n_reads=8000, seed=3, error=0.01)
    g = oracle.Graph(31, nranks=1)
    g.set_reads(packed, offsets, lengths)
    g.count_kmers()
    cases.append(g.histogram())
    for _ in range(30):
        n = int(rng.integers(1, 300))
        cov = np.sort(rng.choice(np.arange(1, 20000), size=n, replace=False)).astype(np.uint32)
        cnt = (rng.integers(1, 1 << int(rng.integers(2, 40)), size=n) * 2).astype(np.uint64)
        cases.append((cov, cnt))
    cases.append((np.array([12000, 13000], dtype=np.uint32), np.array([2, 4], dtype=np.uint64)))   # nothing below the loop's 9999
    for i, (cov, cnt) in enumerate(cases):
        mn, pk, rp = oracle.coverage_distribution(cov, cnt)
        assert ray_amd.coverage_distribution(cov, cnt) == (mn, pk, rp)
        a, b = str(tmp_path / "a.txt"), str(tmp_path / "b.txt")
        oracle.write_coverage_analysis(cov, cnt, 31, mn, pk, rp, "RayOutput/CoverageDistribution.txt", a)
        ray_amd.write_coverage_analysis(cov, cnt, 31, mn, pk, rp, "RayOutput/CoverageDistribution.txt", b)
        assert open(a, "rb").read() == open(b, "rb").read(), "case %d" % i
    text = open(b).read().splitlines()
    assert text[0] == "k-mer length:\t31" and text[2] == "Lowest coverage observed:\t9999" and text[-1] == "DistributionFile: RayOutput/CoverageDistribution.txt"
    assert text[7] == "Percentage of vertices with coverage 9999:\t0 %"
