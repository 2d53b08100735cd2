"""ctypes binding of oracle/libray_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/ray_oracle.h).  The product package ray_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libray_oracle.so")

BLOOM_DEFAULT = (1 << 64) - 1


def build(force=False):
    newest = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("ray_oracle.c", "ro_synth.c", "ray_oracle.h"))
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < newest:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        u64p = C.POINTER(C.c_uint64)
        u32p = C.POINTER(C.c_uint32)
        u8p = C.POINTER(C.c_uint8)
        L.ro_read_pack.argtypes = [C.c_char_p, C.c_int, u8p]
        L.ro_read_pack.restype = C.c_int
        L.ro_read_get_seq.argtypes = [u8p, C.c_int, C.c_char_p]
        L.ro_read_required_bytes.argtypes = [C.c_int]
        L.ro_read_required_bytes.restype = C.c_int
        L.ro_word_id.argtypes = [C.c_char_p, u64p]
        L.ro_complement_vertex.argtypes = [u64p, C.c_int, u64p]
        L.ro_less.argtypes = [u64p, u64p, C.c_int]
        L.ro_less.restype = C.c_int
        for f in (L.ro_mix1, L.ro_mix2):
            f.argtypes = [C.c_uint64]
            f.restype = C.c_uint64
        for f in (L.ro_hash1, L.ro_hash2):
            f.argtypes = [u64p, C.c_int]
            f.restype = C.c_uint64
        L.ro_vertex_rank.argtypes = [u64p, C.c_int, C.c_int, C.c_int]
        L.ro_vertex_rank.restype = C.c_int
        L.ro_id_to_word.argtypes = [u64p, C.c_int, C.c_char_p]
        L.ro_outgoing_edges.argtypes = [u64p, C.c_uint8, C.c_int, C.c_int, u64p]
        L.ro_outgoing_edges.restype = C.c_int
        L.ro_ingoing_edges.argtypes = [u64p, C.c_uint8, C.c_int, C.c_int, u64p]
        L.ro_ingoing_edges.restype = C.c_int
        L.ro_bloom_bit_indices.argtypes = [u64p, C.c_int, C.c_uint64, u64p]
        L.ro_default_bloom_bits.argtypes = [C.c_int, C.c_uint64]
        L.ro_default_bloom_bits.restype = C.c_uint64
        L.ro_convert_bitmap.argtypes = [C.c_uint8]
        L.ro_convert_bitmap.restype = C.c_uint8
        for f in (L.ro_add_outgoing_edge, L.ro_add_ingoing_edge):
            f.argtypes = [C.c_uint8, u64p, u64p, C.c_int, C.c_int]
            f.restype = C.c_uint8
        L.ro_get_edges.argtypes = [C.c_uint8, u64p, C.c_int, C.c_int]
        L.ro_get_edges.restype = C.c_uint8
        L.ro_graph_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64]
        L.ro_graph_create.restype = C.c_void_p
        L.ro_graph_destroy.argtypes = [C.c_void_p]
        L.ro_graph_set_reads.argtypes = [C.c_void_p, u8p, u64p, u32p, C.c_uint64]
        for name in ("ro_graph_count_kmers", "ro_graph_add_edges", "ro_graph_purge_edges"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_int
        for name in ("ro_graph_size", "ro_graph_bloom_bits", "ro_graph_bloom_set_bits", "ro_graph_bloom_insertions"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_int]
            getattr(L, name).restype = C.c_uint64
        L.ro_graph_occurrences.argtypes = [C.c_void_p]
        L.ro_graph_occurrences.restype = C.c_uint64
        L.ro_graph_export.argtypes = [C.c_void_p, C.c_int, u8p, C.c_uint64]
        L.ro_graph_export.restype = C.c_uint64
        L.ro_graph_find.argtypes = [C.c_void_p, u64p, u32p, u8p]
        L.ro_graph_find.restype = C.c_int
        L.ro_graph_histogram.argtypes = [C.c_void_p, u32p, u64p, C.c_uint64]
        L.ro_graph_histogram.restype = C.c_uint64
        L.ro_graph_degree_distribution.argtypes = [C.c_void_p, u64p]
        L.ro_graph_kmers_txt.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
        L.ro_graph_kmers_txt.restype = C.c_uint64
        L.ro_coverage_distribution.argtypes = [u32p, u64p, C.c_uint64, C.c_char_p,
                                               C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ro_coverage_distribution.restype = C.c_int
        L.ro_graph_write_genome_graph.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
        L.ro_graph_write_genome_graph.restype = C.c_int
        L.ro_write_coverage_checkpoint.argtypes = [u32p, u64p, C.c_uint64, C.c_char_p]
        L.ro_write_coverage_checkpoint.restype = C.c_int
        L.ro_graph_write_partition.argtypes = [C.c_void_p, C.c_char_p]
        L.ro_graph_write_partition.restype = C.c_int
        L.ro_load_sequences.argtypes = [C.c_char_p, C.c_int, u8p, C.c_uint64, u64p, u32p, C.c_uint64, u64p]
        L.ro_load_sequences.restype = C.c_uint64
        L.ro_graph_read_markers.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.ro_graph_read_markers.restype = C.c_int
        L.ro_write_coverage_analysis.argtypes = [u32p, u64p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p]
        L.ro_write_coverage_analysis.restype = C.c_int
        L.ro_graph_read_genome_graph.argtypes = [C.c_void_p, C.c_char_p]
        L.ro_graph_read_genome_graph.restype = C.c_int
        L.ro_synth_generate.argtypes = [u8p, u64p, u32p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_double]
        L.ro_synth_generate.restype = C.c_int
        _lib = L
    return _lib


def _k2(words):
    a = (C.c_uint64 * 2)(int(words[0]), int(words[1]) if len(words) > 1 else 0)
    return a


def word_id(s):
    out = (C.c_uint64 * 2)()
    lib().ro_word_id(s.encode(), out)
    return (out[0], out[1])


def complement_vertex(words, k):
    out = (C.c_uint64 * 2)()
    lib().ro_complement_vertex(_k2(words), k, out)
    return (out[0], out[1])


def id_to_word(words, k):
    buf = C.create_string_buffer(k + 1)
    lib().ro_id_to_word(_k2(words), k, buf)
    return buf.value.decode()


def hash1(words, W):
    return lib().ro_hash1(_k2(words), W)


def hash2(words, W):
    return lib().ro_hash2(_k2(words), W)


def vertex_rank(words, k, W, size):
    return lib().ro_vertex_rank(_k2(words), k, W, size)


def bloom_bit_indices(lower, W, m_bits):
    out = (C.c_uint64 * 8)()
    lib().ro_bloom_bit_indices(_k2(lower), W, m_bits, out)
    return list(out)


def convert_bitmap(b):
    return lib().ro_convert_bitmap(b)


def read_pack(seq, trim=True):
    buf = (C.c_uint8 * (len(seq) // 4 + 2))()
    n = lib().ro_read_pack(seq.encode(), 1 if trim else 0, buf)
    nbytes = lib().ro_read_required_bytes(n)
    return bytes(buf[:nbytes]), n


def read_get_seq(packed, length):
    buf = (C.c_uint8 * max(1, len(packed))).from_buffer_copy(packed if packed else b"\0")
    out = C.create_string_buffer(length + 1)
    lib().ro_read_get_seq(buf, length, out)
    return out.value.decode()


def outgoing_edges(words, edges, k, W):
    out = (C.c_uint64 * 8)()
    n = lib().ro_outgoing_edges(_k2(words), edges, k, W, out)
    return [(out[2 * i], out[2 * i + 1]) for i in range(n)]


def ingoing_edges(words, edges, k, W):
    out = (C.c_uint64 * 8)()
    n = lib().ro_ingoing_edges(_k2(words), edges, k, W, out)
    return [(out[2 * i], out[2 * i + 1]) for i in range(n)]


def coverage_distribution(cov, cnt, path=None):
    cov = np.ascontiguousarray(cov, dtype=np.uint32)
    cnt = np.ascontiguousarray(cnt, dtype=np.uint64)
    mn, pk, rp = C.c_int(), C.c_int(), C.c_int()
    rc = lib().ro_coverage_distribution(
        cov.ctypes.data_as(C.POINTER(C.c_uint32)), cnt.ctypes.data_as(C.POINTER(C.c_uint64)),
        len(cov), path.encode() if path else None, C.byref(mn), C.byref(pk), C.byref(rp))
    if rc != 0:
        raise RuntimeError("ro_coverage_distribution rc=%d" % rc)
    return mn.value, pk.value, rp.value


class Graph:
    """R-rank sequential model of the reference path (ray_oracle.h)."""

    def __init__(self, k, nranks=1, bloom_bits=BLOOM_DEFAULT, W=None):
        self.k = k
        self.W = W if W is not None else (1 if k <= 31 else 2)
        self.nranks = nranks
        self._g = lib().ro_graph_create(k, self.W, nranks, bloom_bits)
        if not self._g:
            raise ValueError("bad oracle configuration")
        self._keep = None

    def close(self):
        if self._g:
            lib().ro_graph_destroy(self._g)
            self._g = None

    def __del__(self):
        self.close()

    def set_reads(self, packed, offsets, lengths):
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
        self._keep = (packed, offsets, lengths)
        lib().ro_graph_set_reads(self._g, packed.ctypes.data_as(C.POINTER(C.c_uint8)),
                                 offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
                                 lengths.ctypes.data_as(C.POINTER(C.c_uint32)), len(lengths))

    def count_kmers(self):
        lib().ro_graph_count_kmers(self._g)

    def add_edges(self):
        lib().ro_graph_add_edges(self._g)

    def purge_edges(self):
        lib().ro_graph_purge_edges(self._g)

    def size(self, rank=-1):
        return lib().ro_graph_size(self._g, rank)

    def occurrences(self):
        return lib().ro_graph_occurrences(self._g)

    def bloom_bits(self, rank):
        return lib().ro_graph_bloom_bits(self._g, rank)

    def bloom_set_bits(self, rank):
        return lib().ro_graph_bloom_set_bits(self._g, rank)

    def bloom_insertions(self, rank):
        return lib().ro_graph_bloom_insertions(self._g, rank)

    def export(self, rank=-1):
        """structured array (key[W] u64, coverage u32, edges u8) sorted by key."""
        n = self.size(rank) // 2
        rec = self.W * 8 + 5
        buf = np.zeros(max(1, n) * rec, dtype=np.uint8)
        got = lib().ro_graph_export(self._g, rank, buf.ctypes.data_as(C.POINTER(C.c_uint8)), n)
        assert got == n
        return records_to_struct(buf[: n * rec], self.W)

    def histogram(self):
        cap = 1 << 16
        while True:
            cov = np.zeros(cap, dtype=np.uint32)
            cnt = np.zeros(cap, dtype=np.uint64)
            n = lib().ro_graph_histogram(self._g, cov.ctypes.data_as(C.POINTER(C.c_uint32)),
                                         cnt.ctypes.data_as(C.POINTER(C.c_uint64)), cap)
            if n <= cap:
                return cov[:n].copy(), cnt[:n].copy()
            cap = int(n)

    def degree_distribution(self):
        out = np.zeros(25, dtype=np.uint64)
        lib().ro_graph_degree_distribution(self._g, out.ctypes.data_as(C.POINTER(C.c_uint64)))
        return out.reshape(5, 5)

    def kmers_txt(self):
        need = lib().ro_graph_kmers_txt(self._g, None, 0)
        buf = C.create_string_buffer(int(need) + 1)
        lib().ro_graph_kmers_txt(self._g, buf, need)
        return buf.raw[:need].decode()

    def write_genome_graph(self, rank, path):
        assert lib().ro_graph_write_genome_graph(self._g, rank, path.encode()) == 0

    def read_genome_graph(self, path):
        """the reference's checkpoint reader (MessageProcessor.cpp:596-655) into this (empty, one-rank) graph; 0 = ok,
        -2 / -3 / -4 = the reader's parents / children / size assertion (:633-641,650) would have failed"""
        return lib().ro_graph_read_genome_graph(self._g, path.encode())

    def write_partition(self, path):
        assert lib().ro_graph_write_partition(self._g, path.encode()) == 0

    def read_markers(self):
        n = len(self._keep[2])
        fwd = np.zeros(max(1, n), dtype=np.int32)
        rev = np.zeros(max(1, n), dtype=np.int32)
        lib().ro_graph_read_markers(self._g, fwd.ctypes.data_as(C.POINTER(C.c_int32)), rev.ctypes.data_as(C.POINTER(C.c_int32)))
        return fwd[:n], rev[:n]

    def find(self, words):
        cov = C.c_uint32()
        e = C.c_uint8()
        found = lib().ro_graph_find(self._g, _k2(words), C.byref(cov), C.byref(e))
        return bool(found), cov.value, e.value


def write_coverage_checkpoint(cov, cnt, path):
    cov = np.ascontiguousarray(cov, dtype=np.uint32)
    cnt = np.ascontiguousarray(cnt, dtype=np.uint64)
    assert lib().ro_write_coverage_checkpoint(cov.ctypes.data_as(C.POINTER(C.c_uint32)), cnt.ctypes.data_as(C.POINTER(C.c_uint64)),
                                              len(cov), path.encode()) == 0


def write_coverage_analysis(cov, cnt, k, minimum, peak, repeat, distribution_file, path):
    """CoverageDistributionAnalysis.txt (MachineHelper.cpp:283-327)"""
    cov = np.ascontiguousarray(cov, dtype=np.uint32)
    cnt = np.ascontiguousarray(cnt, dtype=np.uint64)
    assert lib().ro_write_coverage_analysis(cov.ctypes.data_as(C.POINTER(C.c_uint32)), cnt.ctypes.data_as(C.POINTER(C.c_uint64)), len(cov),
                                            k, minimum, peak, repeat, distribution_file.encode(), path.encode()) == 0


def load_sequences(path, is_fastq):
    """(packed, offsets, lengths) of a FASTA / FASTQ file as the reference's loaders build them"""
    used = C.c_uint64()
    n = lib().ro_load_sequences(path.encode(), int(is_fastq), None, 0, None, None, 0, C.byref(used))
    packed = np.zeros(max(1, used.value), dtype=np.uint8)
    offsets = np.zeros(max(1, n), dtype=np.uint64)
    lengths = np.zeros(max(1, n), dtype=np.uint32)
    lib().ro_load_sequences(path.encode(), int(is_fastq), packed.ctypes.data_as(C.POINTER(C.c_uint8)), len(packed),
                            offsets.ctypes.data_as(C.POINTER(C.c_uint64)), lengths.ctypes.data_as(C.POINTER(C.c_uint32)),
                            n, C.byref(used))
    return packed[: used.value], offsets[:n], lengths[:n]


def synth_generate(n, genome, read_len, seed, error=0.0, first=0):
    """reads first .. first+n-1 of the counter-based generator (oracle/ro_synth.c = tools/rg_synth.hip on the device)
    -> (packed, offsets, lengths)"""
    bpr = (read_len + 3) // 4
    packed = np.zeros(max(1, n * bpr), dtype=np.uint8)
    offsets = np.zeros(max(1, n), dtype=np.uint64)
    lengths = np.zeros(max(1, n), dtype=np.uint32)
    rc = lib().ro_synth_generate(packed.ctypes.data_as(C.POINTER(C.c_uint8)), offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
                                 lengths.ctypes.data_as(C.POINTER(C.c_uint32)), n, first, genome, read_len, seed, error)
    if rc != 0:
        raise ValueError("ro_synth_generate rc=%d" % rc)
    return packed[: n * bpr], offsets[:n], lengths[:n]


def record_dtype(W):
    return np.dtype({"names": ["key", "coverage", "edges"],
                     "formats": [("<u8", (W,)), "<u4", "u1"],
                     "offsets": [0, W * 8, W * 8 + 4], "itemsize": W * 8 + 5})


def records_to_struct(buf, W):
    """Vertex::dump records (W*8 + 4 + 1 bytes each) -> structured numpy array."""
    return np.frombuffer(bytes(buf), dtype=record_dtype(W))
