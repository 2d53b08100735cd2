"""SURVEY.md 8f rows on the GPU, through the C ABI: read ingest (FASTA/FASTQ text -> packed reads in HBM)
and the SequencesIndexer marker look-ups, both against the oracle's restatement of the reference."""
import numpy as np
import pytest

from helpers import assert_records_equal
from ray_amd.exchange import partition_reads
from ray_amd.synth import make_reads_numpy

pytestmark = pytest.mark.gpu


def _messy_sequences(n, seed):
    rng = np.random.default_rng(seed)
    seqs = []
    for i in range(n):
        L = int(rng.integers(0, 140))
        s = "".join(rng.choice(list("ACGT"), size=L))
        if i % 3 == 0:
            s = s.lower()
        if i % 5 == 0 and L > 10:   # inner junk becomes A, leading / trailing junk is trimmed
            s = "NN" + s[:5] + "N" + s[6:9] + "x" + s[10:] + "NNN"
        if i % 7 == 0:
            s = "N" * int(rng.integers(0, 4))  # nothing left after the trim
        seqs.append(s)
    return seqs


def test_ingest_fasta_and_fastq(oracle, tmp_path):
    import ray_amd
    seqs = _messy_sequences(4000, 11)
    fa, fq = str(tmp_path / "reads.fasta"), str(tmp_path / "reads.fastq")
    with open(fa, "w") as f:
        for i, s in enumerate(seqs):
            f.write(">read%d extra words\n" % i)
            for j in range(0, len(s), 60):          # multi-line records
                f.write(s[j:j + 60] + "\n")
    with open(fq, "w") as f:
        for i, s in enumerate(seqs):
            f.write("@read%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)))
    for path, is_fq in ((fa, 0), (fq, 1)):
        p, o, l = oracle.load_sequences(path, is_fq)
        assert ray_amd.count_sequences(path) == len(l) == len(seqs)
        b = ray_amd.GraphBuilder(21)
        assert b.load_sequences(path) == len(l)
        gp, go, gl = b.get_reads()
        assert np.array_equal(gl, l) and np.array_equal(go, o) and np.array_equal(gp, p)
        # the partition of rank 1 of 3 (SequencesLoader.cpp:287-297), added in two pieces
        first, last = partition_reads(len(l), 1, 3)
        mid = (first + last) // 2
        b2 = ray_amd.GraphBuilder(21)
        assert b2.load_sequences(path, first, mid) + b2.load_sequences(path, mid, last) == last - first
        gp2, go2, gl2 = b2.get_reads()
        assert np.array_equal(gl2, l[first:last])
        for i in (0, 1, len(gl2) // 2, len(gl2) - 1):
            nb = (int(gl2[i]) + 3) // 4
            assert np.array_equal(gp2[int(go2[i]):int(go2[i]) + nb], p[int(o[first + i]):int(o[first + i]) + nb])
        # and the graph built from the ingested reads is the oracle's graph of the same file
        g = oracle.Graph(21, bloom_bits=40000)
        g.set_reads(p, o, l)
        g.count_kmers()
        g.add_edges()
        b3 = ray_amd.GraphBuilder(21, bloom_bits=40000)
        b3.load_sequences(path)
        b3.count_kmers()
        b3.add_edges()
        assert_records_equal(b3.export(), g.export(), path)


def test_add_reads_text_rejects_oversized_sequences():
    import ray_amd
    b = ray_amd.GraphBuilder(21)
    with pytest.raises(ray_amd.RayGraphError):
        b.add_reads_text(b"A" * 70000, [0], [70000])


@pytest.mark.parametrize("k,bloom", [(21, 30000), (31, None), (63, 0)])
def test_read_markers_match_indexer_worker(oracle, k, bloom):
    import ray_amd
    from ray_amd.graph import RG_BLOOM_DEFAULT
    packed, offsets, lengths = make_reads_numpy(genome=4000, read_len=120 if k < 63 else 230, n_reads=3000, seed=17 + k, error=0.03, ragged=True)
    g = oracle.Graph(k, bloom_bits=oracle.BLOOM_DEFAULT if bloom is None else bloom)
    g.set_reads(packed, offsets, lengths)
    g.count_kmers()
    wf, wr = g.read_markers()
    b = ray_amd.GraphBuilder(k, bloom_bits=RG_BLOOM_DEFAULT if bloom is None else bloom)
    half = len(lengths) // 2
    b.add_reads(packed, offsets[:half], lengths[:half])      # two batches: marker arrays follow the order added
    b.add_reads(packed, offsets[half:], lengths[half:])
    b.count_kmers()
    gf, gr = b.read_markers(len(lengths))
    assert (wf >= 0).sum() > 100 and (wf < 0).sum() > 0, "both outcomes must occur in the test data"
    assert np.array_equal(gf, wf) and np.array_equal(gr, wr)


def test_standalone_driver_writes_rays_files(oracle, tmp_path):
    """ray_amd/raygraph_build (C++ over the C ABI, no Python in the product path): FASTQ pair + FASTA in, Ray's
    graph-construction files out; every file against the oracle's restatement of the reference writer."""
    import os
    import subprocess
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(here, "ray_amd", "raygraph_build")
    assert os.path.exists(exe), "build it with make -C ray_amd/csrc"
    rng = np.random.default_rng(5)
    genome = "".join(rng.choice(list("ACGT"), size=6000))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}

    def reads(n, seed):
        r = np.random.default_rng(seed)
        out = []
        for _ in range(n):
            p = int(r.integers(0, len(genome) - 100))
            s = list(genome[p:p + 100])
            for j in range(100):
                if r.random() < 0.01:
                    s[j] = "ACGT"[int(r.integers(0, 4))]
            s = "".join(s)
            if r.random() < 0.5:
                s = "".join(comp[c] for c in reversed(s))
            out.append(s)
        return out
    left, right, single = reads(1500, 1), reads(1500, 2), reads(700, 3)
    files = []
    for name, rs in (("left.fastq", left), ("right.fastq", right)):
        with open(tmp_path / name, "w") as f:
            for i, s in enumerate(rs):
                f.write("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)))
        files.append(str(tmp_path / name))
    with open(tmp_path / "single.fasta", "w") as f:
        for i, s in enumerate(single):
            f.write(">s%d\n%s\n%s\n" % (i, s[:50], s[50:]))
    files.append(str(tmp_path / "single.fasta"))
    out = str(tmp_path) + "/out_"
    ck = tmp_path / "ck"
    ck.mkdir()
    res = subprocess.run([exe, "-k", "31", "-p", files[0], files[1], "-s", files[2], "-o", out, "-write-kmers",
                          "-write-checkpoints", str(ck), "-bloom-filter-bits", "400000"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr

    # the oracle on the same files in the same order (read ids run through the files in command-line order)
    parts = [oracle.load_sequences(files[0], 1), oracle.load_sequences(files[1], 1), oracle.load_sequences(files[2], 0)]
    packed = np.concatenate([p[0] for p in parts])
    base = np.cumsum([0] + [len(p[0]) for p in parts])
    offsets = np.concatenate([p[1] + np.uint64(base[i]) for i, p in enumerate(parts)])
    lengths = np.concatenate([p[2] for p in parts])
    g = oracle.Graph(31, bloom_bits=400000)
    g.set_reads(packed, offsets, lengths)
    g.count_kmers()
    cov, cnt = g.histogram()
    want_dist = str(tmp_path / "want_dist.txt")
    mpr = oracle.coverage_distribution(cov, cnt, want_dist)
    assert open(out + "CoverageDistribution.txt", "rb").read() == open(want_dist, "rb").read()
    assert "minimum coverage %d, peak coverage %d, repeat coverage %d" % tuple(mpr) in res.stdout
    g.add_edges()
    g.purge_edges()
    body = [l for l in open(out + "kmers.txt").read().splitlines() if not l.startswith("#")]
    assert sorted(body) == sorted(g.kmers_txt().strip().split("\n"))
    deg = [int(l.split("\t")[2]) for l in open(out + "degreeDistribution.txt").read().splitlines() if not l.startswith("#")]
    assert deg == [int(x) for x in g.degree_distribution().reshape(-1)]
    want_part = str(tmp_path / "want_part.txt")
    g.write_partition(want_part)
    assert open(out + "GraphPartition.txt").read() == open(want_part).read()
    # checkpoints: same set of entries as the reference writer's (the driver writes in table order, as Ray does)
    from test_formats_cpu import read_genome_graph
    want_gg = str(tmp_path / "want_gg")
    g.write_genome_graph(0, want_gg)
    got = read_genome_graph(str(ck / "Rank0.Checkpoint.GenomeGraph.ray"), 1)
    want = read_genome_graph(want_gg, 1)
    assert len(got) == len(want) and sorted(got) == sorted(want)
    want_cc = str(tmp_path / "want_cc")
    oracle.write_coverage_checkpoint(cov, cnt, want_cc)
    assert open(ck / "Rank0.Checkpoint.CoverageDistribution.ray", "rb").read() == open(want_cc, "rb").read()
    # CoverageDistributionAnalysis.txt (MachineHelper.cpp:283-327), last line = the distribution file's path as given
    want_an = str(tmp_path / "want_an.txt")
    oracle.write_coverage_analysis(cov, cnt, 31, mpr[0], mpr[1], mpr[2], out + "CoverageDistribution.txt", want_an)
    assert open(out + "CoverageDistributionAnalysis.txt", "rb").read() == open(want_an, "rb").read()

    # one rank with a communicator all the same: RCCL initialised by the C++ driver, every exchange a send to itself
    out1 = str(tmp_path) + "/rccl1_"
    res = subprocess.run([exe, "-k", "31", "-p", files[0], files[1], "-s", files[2], "-o", out1, "-write-kmers", "-bloom-filter-bits", "400000",
                          "-gpus", "1", "-transport", "rccl", "-route", "records", "-parts", "3"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "collectives" in res.stdout
    for name in ("CoverageDistribution.txt", "GraphPartition.txt", "degreeDistribution.txt"):
        assert open(out1 + name, "rb").read() == open(out + name, "rb").read(), name
    assert sorted(open(out1 + "kmers.txt").read().splitlines()) == sorted(open(out + "kmers.txt").read().splitlines())

    # ---- the same job on G ranks, one process each (here: sharing the one GPU, bytes through host shared memory; on a node
    # -transport rccl), as `mpiexec -n G Ray` lays it out: per-rank checkpoints against the G-rank oracle's
    for G, route in ((2, "auto"), (3, "records")):
        outg = str(tmp_path) + "/out%d_" % G
        ckg = tmp_path / ("ck%d" % G)
        ckg.mkdir()
        res = subprocess.run([exe, "-k", "31", "-p", files[0], files[1], "-s", files[2], "-o", outg, "-write-kmers", "-write-checkpoints", str(ckg),
                              "-bloom-filter-bits", "400000", "-gpus", str(G), "-devices", "0", "-transport", "shm", "-route", route, "-parts", "2"],
                             capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stdout + res.stderr
        gg = oracle.Graph(31, nranks=G, bloom_bits=400000)
        gg.set_reads(packed, offsets, lengths)
        gg.count_kmers()
        covg, cntg = gg.histogram()
        oracle.coverage_distribution(covg, cntg, want_dist)
        assert open(outg + "CoverageDistribution.txt", "rb").read() == open(want_dist, "rb").read()
        gg.add_edges()
        gg.purge_edges()
        for r in range(G):
            gg.write_genome_graph(r, want_gg)
            got = read_genome_graph(str(ckg / ("Rank%d.Checkpoint.GenomeGraph.ray" % r)), 1)
            want = read_genome_graph(want_gg, 1)
            assert len(got) == len(want) > 100 and sorted(got) == sorted(want), "rank %d of %d" % (r, G)
            assert "Rank %d : partition is" % r in res.stdout
        gg.write_partition(want_part)
        assert open(outg + "GraphPartition.txt").read() == open(want_part).read()
        lines = open(outg + "kmers.txt").read().splitlines()
        assert lines[0].startswith("# Generated by Ray") and len([l for l in lines if l.startswith("#")]) == 13
        assert sorted(l for l in lines if not l.startswith("#")) == sorted(gg.kmers_txt().strip().split("\n"))
        deg = [int(l.split("\t")[2]) for l in open(outg + "degreeDistribution.txt").read().splitlines() if not l.startswith("#")]
        assert deg == [int(x) for x in gg.degree_distribution().reshape(-1)]
        # and back: every rank loads its own checkpoint the way the reference's reader does; same files again
        outr = str(tmp_path) + "/back%d_" % G
        res = subprocess.run([exe, "-k", "31", "-read-checkpoints", str(ckg), "-o", outr, "-write-kmers", "-gpus", str(G), "-devices", "0", "-transport", "shm"],
                             capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stdout + res.stderr
        assert open(outr + "CoverageDistribution.txt", "rb").read() == open(want_dist, "rb").read()
        assert sorted(open(outr + "kmers.txt").read().splitlines()) == sorted(lines)
        assert open(outr + "GraphPartition.txt").read() == open(want_part).read()


@pytest.mark.parametrize("k", [31, 63])
def test_import_vertices_resumes_from_exported_records(oracle, k):
    """what Ray does when it reads the GenomeGraph checkpoint (MessageProcessor.cpp:596-655): the exported records
    become the vertex store again, and the later stages (purge, look-ups, degree table) give the same results"""
    import ray_amd
    packed, offsets, lengths = make_reads_numpy(genome=8000, read_len=150, n_reads=2500, seed=31 + k, error=0.02)
    a = ray_amd.GraphBuilder(k, bloom_bits=200000)
    a.add_reads(packed, offsets, lengths)
    a.count_kmers()
    a.add_edges()
    rec = a.export()
    b = ray_amd.GraphBuilder(k)
    b.import_vertices(rec)
    assert b.size_() == a.size_()
    assert_records_equal(b.export(), rec, "imported")
    a.purge_edges()
    b.purge_edges()
    assert_records_equal(b.export(), a.export(), "purged after import")
    assert np.array_equal(b.degree_distribution(), a.degree_distribution())
    g = oracle.Graph(k, bloom_bits=200000)
    g.set_reads(packed, offsets, lengths)
    g.count_kmers()
    g.add_edges()
    g.purge_edges()
    assert_records_equal(b.export(), g.export(), "vs oracle")
