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
