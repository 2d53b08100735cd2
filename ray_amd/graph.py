"""Thin object wrapper over the C ABI (include/raygraph.h).  No compute happens in Python."""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import RG_BLOOM_DEFAULT, RG_ENTRY_BYTES, UINT64_MAX, RgConfig, RgStats


class RayGraphError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("raygraph error %d: %s" % (code, message))
        self.code = code


def record_dtype(kmer_words):
    """Vertex::dump layout (VerticesExtractor/Vertex.cpp:438-454): key words, u32 coverage, u8 edges."""
    W = kmer_words
    return np.dtype({"names": ["key", "coverage", "edges"], "formats": [("<u8", (W,)), "<u4", "u1"],
                     "offsets": [0, W * 8, W * 8 + 4], "itemsize": W * 8 + 5})


def pack_read(sequence, trim=True):
    """Read::constructor packing (SequencesLoader/Read.cpp:31-159) -> (bytes, length)."""
    L = _lib.load()
    raw = sequence.encode()
    buf = (C.c_uint8 * (len(raw) // 4 + 2))()
    n = L.rg_pack_read(raw, 1 if trim else 0, buf)
    if n < 0:
        raise RayGraphError(n, "rg_pack_read")
    return bytes(buf[: (n + 3) // 4]), n


def pack_reads(sequences, trim=True):
    """ArrayOfReads as the three flat arrays rg_add_reads takes."""
    chunks, offsets, lengths, pos = [], [], [], 0
    for s in sequences:
        b, n = pack_read(s, trim)
        chunks.append(b)
        offsets.append(pos)
        lengths.append(n)
        pos += len(b)
    packed = np.frombuffer(b"".join(chunks), dtype=np.uint8).copy() if pos else np.zeros(0, np.uint8)
    return packed, np.asarray(offsets, dtype=np.uint64), np.asarray(lengths, dtype=np.uint32)


def coverage_distribution(coverage, count, path=None):
    """CoverageDistribution (CoverageGatherer/CoverageDistribution.cpp:33-112) -> (min, peak, repeat)."""
    L = _lib.load()
    cov = np.ascontiguousarray(coverage, dtype=np.uint32)
    cnt = np.ascontiguousarray(count, dtype=np.uint64)
    mn, pk, rp = C.c_int32(), C.c_int32(), C.c_int32()
    rc = L.rg_coverage_distribution(cov.ctypes.data_as(C.POINTER(C.c_uint32)), cnt.ctypes.data_as(C.POINTER(C.c_uint64)),
                                    len(cov), path.encode() if path else None, C.byref(mn), C.byref(pk), C.byref(rp))
    if rc != 0:
        raise RayGraphError(rc, "no k-mers found in reads" if len(cov) == 0 else "rg_coverage_distribution")
    return mn.value, pk.value, rp.value


def count_sequences(path):
    """number of records of a FASTA / FASTQ file as Ray's loaders count them"""
    n = C.c_uint64()
    rc = _lib.load().rg_count_sequences(os.fsencode(path), C.byref(n))
    if rc:
        raise RayGraphError(rc, "cannot read %s (plain .fasta/.fa/.fna/.fas/.fastq/.fq only)" % path)
    return n.value


def write_genome_graph(records, k, path):
    """Rank<r>.Checkpoint.GenomeGraph.ray from GraphBuilder.export() records (MessageProcessor.cpp:658-678)"""
    records = np.ascontiguousarray(records)
    W = records.dtype["key"].shape[0]
    rc = _lib.load().rg_write_genome_graph(records.ctypes.data, len(records), k, W, os.fsencode(path))
    if rc:
        raise RayGraphError(rc, "cannot write %s" % path)


def write_coverage_checkpoint(coverage, count, path):
    """Rank0.Checkpoint.CoverageDistribution.ray (MachineHelper.cpp:243-259)"""
    cov = np.ascontiguousarray(coverage, dtype=np.uint32)
    cnt = np.ascontiguousarray(count, dtype=np.uint64)
    rc = _lib.load().rg_write_coverage_checkpoint(cov.ctypes.data_as(C.POINTER(C.c_uint32)), cnt.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                  len(cov), os.fsencode(path))
    if rc:
        raise RayGraphError(rc, "cannot write %s" % path)


def write_coverage_analysis(coverage, count, k, minimum, peak, repeat, distribution_file, path):
    """<prefix>CoverageDistributionAnalysis.txt (MachineHelper.cpp:283-327)"""
    cov = np.ascontiguousarray(coverage, dtype=np.uint32)
    cnt = np.ascontiguousarray(count, dtype=np.uint64)
    rc = _lib.load().rg_write_coverage_analysis(cov.ctypes.data_as(C.POINTER(C.c_uint32)), cnt.ctypes.data_as(C.POINTER(C.c_uint64)), len(cov),
                                                k, minimum, peak, repeat, os.fsencode(distribution_file), os.fsencode(path))
    if rc:
        raise RayGraphError(rc, "cannot write %s" % path)


def read_genome_graph(path, k, kmer_words=0):
    """Rank<r>.Checkpoint.GenomeGraph.ray -> Vertex::dump records, as the reference's reader builds its GridTable
    (MessageProcessor.cpp:596-655); raises when the reader's assertions would fail"""
    W = kmer_words if kmer_words else (1 if k <= 31 else 2)
    n = C.c_uint64()
    rc = _lib.load().rg_read_genome_graph(os.fsencode(path), k, W, None, 0, C.byref(n))
    if rc:
        raise RayGraphError(rc, "cannot read %s" % path)
    buf = np.zeros(max(1, n.value), dtype=record_dtype(W))
    rc = _lib.load().rg_read_genome_graph(os.fsencode(path), k, W, buf.ctypes.data, n.value, C.byref(n))
    if rc:
        raise RayGraphError(rc, "%s is not a consistent GenomeGraph checkpoint" % path)
    return buf[: n.value]


def write_graph_partition(sizes, path):
    """<prefix>GraphPartition.txt (EdgePurger.cpp:172-217); sizes[r] = GridTable size of rank r"""
    a = (C.c_uint64 * len(sizes))(*[int(x) for x in sizes])
    rc = _lib.load().rg_write_graph_partition(a, len(sizes), os.fsencode(path))
    if rc:
        raise RayGraphError(rc, "cannot write %s" % path)


class Comm:
    """rg_comm: this rank's end of the all-to-all(v) between the ranks of a job (include/raygraph.h, "multi-GPU exchange").
    Comm.rccl(id, rank, size, device): RCCL over xGMI, `id` = Comm.unique_id() made on rank 0 and sent to the others;
    Comm.transport(alltoall_u64, alltoallv, rank, size, device): the caller's collectives (callables, see rg_transport)."""

    def __init__(self, handle, keep=None):
        self._L = _lib.load()
        self._c = handle
        self._keep = keep

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(_lib.RG_COMM_ID_BYTES)
        L = _lib.load()
        rc = L.rg_comm_unique_id(buf)
        if rc:
            raise RayGraphError(rc, L.rg_comm_last_error(None).decode())
        return buf.raw

    @classmethod
    def rccl(cls, unique_id, rank, size, device=0):
        L = _lib.load()
        c = C.c_void_p()
        rc = L.rg_comm_create_rccl(C.c_char_p(bytes(unique_id)), rank, size, device, C.byref(c))
        if rc:
            raise RayGraphError(rc, L.rg_comm_last_error(None).decode())
        return cls(c)

    @classmethod
    def transport(cls, alltoall_u64, alltoallv, rank, size, device=0):
        """alltoall_u64(send: list of size*n ints, n) -> list of size*n ints; alltoallv(d_send, send_counts, d_recv, recv_counts,
        unit_bytes) -> None (device pointers as ints, counts as lists)"""
        L = _lib.load()

        def _a2a(user, send, recv, n):
            try:
                got = alltoall_u64([int(send[i]) for i in range(size * n)], n)
                for i in range(size * n):
                    recv[i] = int(got[i])
                return 0
            except Exception:  # noqa: BLE001  (must not propagate into C)
                import traceback
                traceback.print_exc()
                return 1

        def _a2av(user, d_send, scount, d_recv, rcount, unit):
            try:
                alltoallv(d_send or 0, [int(scount[r]) for r in range(size)], d_recv or 0, [int(rcount[r]) for r in range(size)], unit)
                return 0
            except Exception:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                return 1

        t = _lib.RgTransport(None, _lib.ALLTOALL_U64_FN(_a2a), _lib.ALLTOALLV_FN(_a2av))
        c = C.c_void_p()
        rc = L.rg_comm_create_transport(C.byref(t), rank, size, device, C.byref(c))
        if rc:
            raise RayGraphError(rc, L.rg_comm_last_error(None).decode())
        return cls(c, keep=t)

    def stats(self):
        s = _lib.RgCommStats()
        self._L.rg_comm_get_stats(self._c, C.byref(s))
        return {"bytes_sent": int(s.bytes_sent), "bytes_received": int(s.bytes_received), "collectives": int(s.collectives),
                "wait_seconds": float(s.wait_seconds)}

    def close(self):
        if getattr(self, "_c", None):
            self._L.rg_comm_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _comm_handle(comm):
    return comm._c if comm is not None else None


class GraphBuilder:
    """One rank's share of the distributed de Bruijn graph on one GPU."""

    def __init__(self, k, rank=0, size=1, device=0, bloom_bits=RG_BLOOM_DEFAULT, kmer_words=0, table_slots=0,
                 first_read_id=0, stream=None, sweep=0):
        self._L = _lib.load()
        cfg = RgConfig(k=k, kmer_words=kmer_words, rank=rank, size=size, device=device, reserved0=0,
                       bloom_bits=bloom_bits, table_slots=table_slots, first_read_id=first_read_id,
                       stream=C.c_void_p(stream) if stream else None)
        self._h = C.c_void_p()
        rc = self._L.rg_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            msg = self._L.rg_last_error(None).decode()
            self._h = None
            raise RayGraphError(rc, msg)
        if sweep:   # tests and experiments: force internal routes (rg_debug_set)
            self._check(self._L.rg_debug_set(self._h, sweep))
        self.k, self.rank, self.size, self.device = k, rank, size, device
        self.kmer_words = kmer_words if kmer_words else (1 if k <= 31 else 2)
        self._keep = []

    # -- plumbing
    def _check(self, rc):
        if rc != 0:
            raise RayGraphError(rc, self._L.rg_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.rg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- input
    def add_reads(self, packed, offsets, lengths):
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
        self._check(self._L.rg_add_reads(self._h, packed.ctypes.data, packed.size, offsets.ctypes.data,
                                         lengths.ctypes.data, lengths.size))

    def add_reads_device(self, packed_ptr, packed_bytes, offsets_ptr, lengths_ptr, n_reads, max_length, keep=None):
        """device-resident arrays (e.g. torch tensors' data_ptr()); `keep` pins their owners."""
        if keep is not None:
            self._keep.append(keep)
        self._check(self._L.rg_add_reads_device(self._h, packed_ptr, packed_bytes, offsets_ptr, lengths_ptr, n_reads,
                                                max_length))

    def add_reads_text(self, text, starts, raw_lengths):
        """raw sequences as they stand in a file; trimmed and packed on the device (Read.cpp:31-159)"""
        text = bytes(text)
        starts = np.ascontiguousarray(starts, dtype=np.uint64)
        raw = np.ascontiguousarray(raw_lengths, dtype=np.uint32)
        self._check(self._L.rg_add_reads_text(self._h, text, len(text), starts.ctypes.data, raw.ctypes.data, len(raw)))

    def load_sequences(self, path, first=0, last=UINT64_MAX):
        """sequences [first, last) of a FASTA / FASTQ file (SequencesLoader.cpp:287-297 partitions by these ids)"""
        n = C.c_uint64()
        self._check(self._L.rg_load_sequences(self._h, os.fsencode(path), first, last, C.byref(n)))
        return n.value

    def get_reads(self):
        """(packed, byte offsets, lengths) of everything added so far, back on the host"""
        n, nb = C.c_uint64(), C.c_uint64()
        self._check(self._L.rg_get_reads(self._h, None, 0, None, None, 0, C.byref(n), C.byref(nb)))
        packed = np.zeros(max(1, nb.value), dtype=np.uint8)
        off = np.zeros(max(1, n.value), dtype=np.uint64)
        ln = np.zeros(max(1, n.value), dtype=np.uint32)
        self._check(self._L.rg_get_reads(self._h, packed.ctypes.data, len(packed), off.ctypes.data, ln.ctypes.data, len(ln),
                                         C.byref(n), C.byref(nb)))
        return packed[: nb.value], off[: n.value], ln[: n.value]

    def pack_reads_noalloc(self, text_ptr, text_bytes, starts_ptr, raw_lengths_ptr, n_reads, packed_ptr, packed_cap,
                           offsets_ptr, lengths_ptr):
        """rg_pack_reads_noalloc: device pointers only, nothing allocated, queued on the builder's stream"""
        self._check(self._L.rg_pack_reads_noalloc(self._h, text_ptr, text_bytes, starts_ptr, raw_lengths_ptr, n_reads,
                                                  packed_ptr, packed_cap, offsets_ptr, lengths_ptr))

    def import_vertices(self, records):
        """load exported records (or a parsed GenomeGraph checkpoint) into a fresh builder: coverage and edges final"""
        records = np.ascontiguousarray(records)
        self._check(self._L.rg_import_vertices(self._h, records.ctypes.data, len(records)))

    def read_markers(self, n_reads):
        """SequencesIndexer marker positions (forward, reverse) of every read, -1 = none (IndexerWorker.cpp:67-248)"""
        import torch
        dev = "cuda:%d" % self.device
        fwd = torch.empty(max(1, n_reads), dtype=torch.int32, device=dev)
        rev = torch.empty(max(1, n_reads), dtype=torch.int32, device=dev)
        self._check(self._L.rg_read_markers(self._h, fwd.data_ptr(), rev.data_ptr()))
        return fwd[:n_reads].cpu().numpy(), rev[:n_reads].cpu().numpy()

    # -- pass 1
    def count_kmers(self):
        self._check(self._L.rg_count_kmers(self._h))

    def count_local(self):
        self._check(self._L.rg_count_local(self._h))

    # -- record exchange (exchange.exchange_records drives these)
    def records_local(self, part=0, nparts=1):
        """reads -> records per destination rank; part p of P covers a P-th of every batch (see exchange_records)"""
        self._check(self._L.rg_records_local_part(self._h, part, nparts))

    def record_counts(self):
        """(records per destination rank, bytes per record)"""
        out = (C.c_uint64 * self.size)()
        nb = C.c_uint32()
        self._check(self._L.rg_record_counts(self._h, out, C.byref(nb)))
        return [int(x) for x in out], nb.value

    def export_records(self, device_ptr, capacity_records):
        self._check(self._L.rg_export_records(self._h, device_ptr, capacity_records))

    def count_records(self, device_ptr, n_records):
        self._check(self._L.rg_count_records(self._h, device_ptr, n_records))

    def owner_counts(self):
        out = (C.c_uint64 * self.size)()
        self._check(self._L.rg_owner_counts(self._h, out))
        return [int(x) for x in out]

    def export_owner_buckets(self, device_ptr, capacity_entries):
        self._check(self._L.rg_export_owner_buckets(self._h, device_ptr, capacity_entries))

    def merge_entries(self, device_ptr, n_entries):
        self._check(self._L.rg_merge_entries(self._h, device_ptr, n_entries))

    def release_local(self):
        self._check(self._L.rg_release_local(self._h))

    def adopt_entries(self, device_ptr, n_entries, keepalive=None):
        """rg_adopt_entries: admission runs in the caller's buffer; `keepalive` (the tensor that owns it) is held until
        resolve_admission has returned"""
        self._check(self._L.rg_adopt_entries(self._h, device_ptr, n_entries))
        self._borrowed = keepalive

    def resolve_admission(self, local_reads=UINT64_MAX):
        self._check(self._L.rg_resolve_admission(self._h, local_reads))
        self._borrowed = None

    # -- the exchanges between ranks, inside the library (rg_comm; comm may be None on one rank)
    def exchange_records(self, comm, parts=1):
        self._check(self._L.rg_exchange_records(self._h, _comm_handle(comm), parts))

    def exchange_owner_buckets(self, comm):
        self._check(self._L.rg_exchange_owner_buckets(self._h, _comm_handle(comm)))

    def count_kmers_exchange(self, comm, route="records", parts=4, local_reads=UINT64_MAX):
        """RAY_SLAVE_MODE_ADD_VERTICES across ranks in one call"""
        r = {"records": _lib.RG_ROUTE_RECORDS, "entries": _lib.RG_ROUTE_ENTRIES}[route]
        self._check(self._L.rg_count_kmers_exchange(self._h, _comm_handle(comm), r, parts, local_reads))

    def histogram_exchange(self, comm):
        """the ranks' histograms merged (MessageProcessor.cpp:1048-1061), on every rank"""
        n = C.c_uint64()
        # the merged histogram has at most as many bins as all ranks hold together; ask with room for them
        cap = 1 << 16
        while True:
            cov = np.zeros(cap, dtype=np.uint32)
            cnt = np.zeros(cap, dtype=np.uint64)
            self._check(self._L.rg_histogram_exchange(self._h, _comm_handle(comm), cov.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                      cnt.ctypes.data_as(C.POINTER(C.c_uint64)), cap, C.byref(n)))
            if n.value <= cap:
                return cov[: n.value], cnt[: n.value]
            cap = int(n.value)

    def purge_edges_exchange(self, comm):
        self._check(self._L.rg_purge_edges_exchange(self._h, _comm_handle(comm)))

    def read_markers_exchange(self, comm, n_reads, parts=1):
        import torch
        dev = "cuda:%d" % self.device
        fwd = torch.empty(max(1, n_reads), dtype=torch.int32, device=dev)
        rev = torch.empty(max(1, n_reads), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        self._check(self._L.rg_read_markers_exchange(self._h, _comm_handle(comm), parts, fwd.data_ptr(), rev.data_ptr()))
        return fwd[:n_reads].cpu().numpy(), rev[:n_reads].cpu().numpy()

    def allgather_u64(self, comm, value):
        out = (C.c_uint64 * self.size)()
        self._check(self._L.rg_allgather_u64(self._h, _comm_handle(comm), int(value), out))
        return [int(x) for x in out]

    # -- histogram / pass 2 / purge
    def histogram(self):
        n = C.c_uint64()
        self._check(self._L.rg_histogram(self._h, None, None, 0, C.byref(n)))
        cov = np.zeros(max(1, n.value), dtype=np.uint32)
        cnt = np.zeros(max(1, n.value), dtype=np.uint64)
        m = C.c_uint64()
        self._check(self._L.rg_histogram(self._h, cov.ctypes.data_as(C.POINTER(C.c_uint32)),
                                         cnt.ctypes.data_as(C.POINTER(C.c_uint64)), n.value, C.byref(m)))
        return cov[: m.value], cnt[: m.value]

    def add_edges(self):
        self._check(self._L.rg_add_edges(self._h))

    def purge_edges(self):
        self._check(self._L.rg_purge_edges(self._h))

    # -- purge across ranks (exchange.purge_edges_distributed drives these)
    def purge_query_counts(self):
        out = (C.c_uint64 * self.size)()
        self._check(self._L.rg_purge_query_counts(self._h, out))
        return [int(x) for x in out]

    def purge_export_queries(self, device_ptr, capacity_queries):
        self._check(self._L.rg_purge_export_queries(self._h, device_ptr, capacity_queries))

    def purge_answer(self, queries_ptr, n, answers_ptr):
        self._check(self._L.rg_purge_answer(self._h, queries_ptr, n, answers_ptr))

    def purge_apply(self, answers_ptr, n):
        self._check(self._L.rg_purge_apply(self._h, answers_ptr, n))

    # -- the hashes against the caller's (rg_check_hashes / rg_hash_kmers)
    def check_hashes(self, hash1, hash2, n_samples=4096):
        """hash1 / hash2: callables taking a tuple of kmer_words uint64 words, returning Ray's hash_function_1 / _2.
        Raises RayGraphError (RG_ERR_HASH) when the device's hashes differ."""
        from ._lib import HASH_FN
        W = self.kmer_words
        f1 = HASH_FN(lambda words, user: int(hash1(tuple(int(words[j]) for j in range(W)))) & 0xFFFFFFFFFFFFFFFF)
        f2 = HASH_FN(lambda words, user: int(hash2(tuple(int(words[j]) for j in range(W)))) & 0xFFFFFFFFFFFFFFFF)
        self._check(self._L.rg_check_hashes(self._h, f1, f2, None, n_samples))

    def hash_kmers(self, kmers):
        q = np.ascontiguousarray(kmers, dtype=np.uint64).reshape(-1, self.kmer_words)
        h1 = np.zeros(len(q), dtype=np.uint64)
        h2 = np.zeros(len(q), dtype=np.uint64)
        u64p = C.POINTER(C.c_uint64)
        self._check(self._L.rg_hash_kmers(self._h, q.ctypes.data_as(u64p), len(q), h1.ctypes.data_as(u64p), h2.ctypes.data_as(u64p)))
        return h1, h2

    # -- markers across ranks (exchange.read_markers_distributed drives these)
    def markers_query_counts(self, part=0, nparts=1):
        """k-mers of the reads' slice `part` of `nparts` per owner rank (the slices of records_local)"""
        out = (C.c_uint64 * self.size)()
        self._check(self._L.rg_markers_query_counts_part(self._h, part, nparts, out))
        return [int(x) for x in out]

    def markers_export_queries(self, device_ptr, capacity_queries):
        self._check(self._L.rg_markers_export_queries(self._h, device_ptr, capacity_queries))

    def markers_answer(self, queries_ptr, n, coverage_ptr):
        self._check(self._L.rg_markers_answer(self._h, queries_ptr, n, coverage_ptr))

    def markers_apply(self, coverage_ptr, n, forward_ptr, reverse_ptr):
        self._check(self._L.rg_markers_apply(self._h, coverage_ptr, n, forward_ptr, reverse_ptr))

    # -- the store
    def size_(self):
        n = C.c_uint64()
        self._check(self._L.rg_size(self._h, C.byref(n)))
        return n.value

    def export_into(self, buf):
        """rg_export into the caller's record array (record_dtype(kmer_words)); returns the number of records written"""
        m = C.c_uint64()
        self._check(self._L.rg_export(self._h, buf.ctypes.data, len(buf), C.byref(m)))
        return int(m.value)

    def export_stream(self, consume, chunk_records=0, user=None):
        """rg_export_stream: `consume` is a callable(records: structured numpy array, valid during the call) or a C function
        pointer of type rg_export_fn (then `user` is passed through)"""
        if callable(consume) and not isinstance(consume, (C._CFuncPtr, C.c_void_p)):
            dt = record_dtype(self.kmer_words)

            def _cb(_user, ptr, n):
                try:
                    buf = (C.c_uint8 * (n * dt.itemsize)).from_address(ptr)
                    return int(consume(np.frombuffer(buf, dtype=dt, count=n)) or 0)
                except Exception:  # noqa: BLE001  (must not propagate into C)
                    import traceback
                    traceback.print_exc()
                    return 1
            fn = _lib.EXPORT_FN(_cb)
        else:
            fn = consume
        self._check(self._L.rg_export_stream(self._h, chunk_records, fn if isinstance(fn, C.c_void_p) else C.cast(fn, C.c_void_p), user))

    def export(self, sort=True):
        """stored vertices as a structured array (key, coverage, edges); sorted by key if asked."""
        n = C.c_uint64()
        self._check(self._L.rg_export(self._h, None, 0, C.byref(n)))
        dt = record_dtype(self.kmer_words)
        buf = np.zeros(max(1, n.value), dtype=dt)
        m = C.c_uint64()
        self._check(self._L.rg_export(self._h, buf.ctypes.data, n.value, C.byref(m)))
        rec = buf[: m.value]
        if sort and len(rec):
            keys = rec["key"]
            order = np.lexsort(tuple(keys[:, j] for j in reversed(range(self.kmer_words))))
            rec = rec[order]
        return rec

    def find(self, kmers):
        """bulk GridTable::find: kmers (n, kmer_words) uint64, either strand -> (coverage, edges)."""
        q = np.ascontiguousarray(kmers, dtype=np.uint64).reshape(-1, self.kmer_words)
        n = q.shape[0]
        cov = np.zeros(n, dtype=np.uint32)
        e = np.zeros(n, dtype=np.uint8)
        self._check(self._L.rg_find(self._h, q.ctypes.data_as(C.POINTER(C.c_uint64)), n,
                                    cov.ctypes.data_as(C.POINTER(C.c_uint32)), e.ctypes.data_as(C.POINTER(C.c_uint8))))
        return cov, e

    def degree_distribution(self):
        out = np.zeros(25, dtype=np.uint64)
        self._check(self._L.rg_degree_distribution(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out.reshape(5, 5)

    def set_profiling(self, enable=True):
        self._check(self._L.rg_set_profiling(self._h, 1 if enable else 0))

    def kernel_profile(self):
        """{kernel name: (total ms, launches)} from HIP events on the launch stream"""
        buf = C.create_string_buffer(8192)
        self._check(self._L.rg_get_kernel_profile(self._h, buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            name, ms, n = line.split()
            out[name] = (float(ms), int(n))
        return out

    def stats(self):
        s = RgStats()
        self._check(self._L.rg_get_stats(self._h, C.byref(s)))
        return {n: int(getattr(s, n)) for n, _ in RgStats._fields_}


__all__ = ["GraphBuilder", "Comm", "RayGraphError", "record_dtype", "pack_read", "pack_reads", "coverage_distribution",
           "RG_BLOOM_DEFAULT", "RG_ENTRY_BYTES"]
