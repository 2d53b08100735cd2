"""Pins the CPU restatement (oracle/) against every known answer the reference holds for this path
(tests/golden/reference_known_answers.json, each entry cites its source)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))


def k2(t):
    return (C.c_uint64 * 2)(*t)


def test_word_id(oracle):
    for case in GOLD["word_id"]:
        w = oracle.word_id(case["kmer"])
        assert [hex(w[0]), hex(w[1])] == case["words"], case["kmer"]
        assert oracle.id_to_word(w, len(case["kmer"])) == case["kmer"]


def test_edge_bitmap_example(oracle):
    """Vertex.h:84-115: ATGAC <- GATGA, -> TGACT, TGACA is 1001 0100; its twin GTCAT reads 0010 1001"""
    ex = GOLD["edge_bitmap_example"]
    k, L = ex["k"], oracle.lib()
    v = oracle.word_id(ex["vertex"])
    stored = 0
    for p in ex["parents"]:
        stored = L.ro_add_ingoing_edge(stored, k2(v), k2(oracle.word_id(p)), k, 1)
    for c in ex["children"]:
        stored = L.ro_add_outgoing_edge(stored, k2(v), k2(oracle.word_id(c)), k, 1)
    assert L.ro_get_edges(stored, k2(v), k, 1) == int(ex["bitmap"], 16)
    twin = oracle.word_id(ex["twin"])
    assert oracle.complement_vertex(v, k) == twin
    assert L.ro_get_edges(stored, k2(twin), k, 1) == int(ex["twin_bitmap"], 16)
    assert oracle.convert_bitmap(int(ex["bitmap"], 16)) == int(ex["twin_bitmap"], 16)
    names = lambda ks: sorted(oracle.id_to_word(x, k) for x in ks)
    assert names(oracle.ingoing_edges(v, int(ex["bitmap"], 16), k, 1)) == sorted(ex["parents"])
    assert names(oracle.outgoing_edges(v, int(ex["bitmap"], 16), k, 1)) == sorted(ex["children"])
    assert names(oracle.ingoing_edges(twin, int(ex["twin_bitmap"], 16), k, 1)) == sorted(ex["twin_parents"])
    assert names(oracle.outgoing_edges(twin, int(ex["twin_bitmap"], 16), k, 1)) == sorted(ex["twin_children"])
    # adding the same edges through the twin gives the same stored byte (the reverse-strand records of
    # VerticesExtractor.cpp:217-251 are redundant)
    stored2 = 0
    for p in ex["twin_parents"]:
        stored2 = L.ro_add_ingoing_edge(stored2, k2(twin), k2(oracle.word_id(p)), k, 1)
    for c in ex["twin_children"]:
        stored2 = L.ro_add_outgoing_edge(stored2, k2(twin), k2(oracle.word_id(c)), k, 1)
    assert stored2 == stored


def test_convert_bitmap_is_an_involution(oracle):
    """Vertex.cpp:403-412 post-conditions"""
    for m in range(256):
        c = oracle.convert_bitmap(m)
        assert oracle.convert_bitmap(c) == m
        assert bin(c).count("1") == bin(m).count("1")
        for base in range(4):  # out bit 4+c <-> in bit 3-c
            assert bool(m & (1 << (4 + base))) == bool(c & (1 << (3 - base)))


def test_hashes_survey_values(oracle):
    """hash-derived values: PARITY UNPINNED (crypto.cpp absent); these pin the restated mixes to what
    SURVEY.md 8c recorded with the same assumed mixes, and the two-word quirk of Kmer.cpp:288-293"""
    for case in GOLD["hashes"]:
        w = oracle.word_id(case["kmer"])
        assert hex(oracle.hash1(w, case["W"])) == case["hash1"]
        assert hex(oracle.hash2(w, case["W"])) == case["hash2"]
        assert oracle.vertex_rank(w, case["k"], case["W"], 8) == case["vertex_rank_8"]


def test_bloom_constants_and_sizing(oracle):
    w = oracle.word_id("ACGTACGTACGTACGTACGTACGTACGTACG")
    h2 = oracle.hash2(w, 1)
    m = 1000003
    assert oracle.bloom_bit_indices(w, 1, m) == [(h2 ^ int(x, 16)) % m for x in GOLD["bloom_magic"]["values"]]
    for case in GOLD["default_bloom_bits"]["cases"]:
        assert oracle.lib().ro_default_bloom_bits(case["k"], case["reads"]) == case["bits"]


def test_kmers_txt_grammar(oracle):
    """the reference's example lines parse under the grammar the oracle emits"""
    g = oracle.Graph(5, bloom_bits=0)
    seqs = ["GATGACT", "GATGACA"]
    packed = b"".join(oracle.read_pack(s)[0] for s in seqs)
    offs = [0, len(oracle.read_pack(seqs[0])[0])]
    g.set_reads(np.frombuffer(packed, dtype=np.uint8), offs, [7, 7])
    g.count_kmers()
    g.add_edges()
    lines = g.kmers_txt().strip().split("\n")
    assert "ATGAC;2;G;A T" in lines  # the Vertex.h example vertex, children enumerated A,C,G,T
    assert "GTCAT;2;A T;C" in lines
    for ex in GOLD["kmers_txt_lines"]:
        seq, cov, par, chi = ex["line"].split(";")
        assert set(seq) <= set("ACGT") and int(cov) > 0
        assert all(t in "ACGT" for t in par.split()) and all(t in "ACGT" for t in chi.split())


def test_complement_and_order(oracle):
    rng = np.random.default_rng(7)
    for k in (5, 21, 31, 33, 47, 63):
        for _ in range(50):
            s = "".join("ACGT"[i] for i in rng.integers(0, 4, k))
            rc = "".join({"A": "T", "C": "G", "G": "C", "T": "A"}[c] for c in reversed(s))
            w = oracle.word_id(s)
            assert oracle.id_to_word(oracle.complement_vertex(w, k), k) == rc
            assert oracle.complement_vertex(oracle.complement_vertex(w, k), k) == w
            assert w != oracle.complement_vertex(w, k)  # odd k: MessageProcessor.cpp:730-732


def test_read_packing(oracle):
    """Read::constructor / trim / getSeq (SequencesLoader/Read.cpp:31-174)"""
    cases = {"ACGT": ("ACGT", bytes([0b11100100])), "NNACGTNN": ("ACGT", None), "acgtn": ("ACGT", None),
             "ACNNGT": ("ACAAGT", None), "NNNN": ("", b""), "": ("", b""), "A": ("A", bytes([0])),
             "TTTTT": ("TTTTT", bytes([0xFF, 0x03])), "xACGTx": ("ACGT", None)}
    for raw, (want, want_bytes) in cases.items():
        packed, n = oracle.read_pack(raw)
        assert n == len(want), raw
        assert oracle.read_get_seq(packed, n) == want, raw
        if want_bytes is not None:
            assert packed == want_bytes, raw
    packed, n = oracle.read_pack("NNACGTNN", trim=False)  # no trimming: every non-ACGT symbol becomes A
    assert oracle.read_get_seq(packed, n) == "AAACGTAA"


def test_coverage_distribution_peak_finder(oracle, tmp_path):
    """CoverageDistribution.cpp:33-112 on hand-checked histograms"""
    # error tail at 2..4, valley at 6, peak at 30
    cov = np.arange(2, 61, dtype=np.uint32)
    cnt = np.array([int(9000 * np.exp(-((int(c) - 30) ** 2) / 50.0)) * 2 + (200000 >> int(c)) * 2 + 2 for c in cov], dtype=np.uint64)
    path = str(tmp_path / "CoverageDistribution.txt")
    mn, pk, rp = oracle.coverage_distribution(cov, cnt, path)
    assert pk == 30
    valley = int(cov[np.argmin(cnt[: 29])])
    assert mn == valley
    assert rp == min(2 * pk, pk + (pk - mn))
    text = open(path).read().split("\n")
    assert text[0] == "# KmerCoverage\tFrequency"
    assert text[1] == "# Any frequency is a even number because of odd k-mer length"
    assert text[2] == "%d %d" % (cov[0], cnt[0]) and text[-1] == ""
    # nothing above the vote threshold of 8192: the reference reads begin() of an empty map; we define position 0
    mn, pk, rp = oracle.coverage_distribution(np.array([2, 3, 4], np.uint32), np.array([10, 50, 20], np.uint64))
    assert (mn, pk, rp) == (2, 2, 2)
    # ties take the leftmost minimum (<=, walking left)
    mn, pk, rp = oracle.coverage_distribution(np.array([1, 2, 3, 4], np.uint32), np.array([9000, 9000, 9000, 20000], np.uint64))
    assert (mn, pk) == (1, 4)
