"""Python model of the *device algorithm* (test infrastructure, small inputs only).

Where oracle/ restates the reference's sequential handlers, this restates what the HIP kernels do:
aggregate every canonical k-mer to (first-arrival stamp, count, edge bits), move the entries to their
owner, then resolve Bloom admission from stamps alone:

    firstTouch(bit) = min over keys touching the bit of their first-arrival stamp
    falsePositive(x) <=> every bit of x has firstTouch < firstArrival(x)

tests/test_model_vs_oracle.py checks this against the oracle on CPU (so the order-free formulation is
proven equal to the order-dependent reference before any GPU runs); the same class doubles as the
CPU stand-in engine for the gloo exchange test.
"""
import ctypes as C

import numpy as np

ENTRY = np.dtype([("k0", "<u8"), ("k1", "<u8"), ("stamp", "<u8"), ("meta", "<u8")])
COUNT_MASK = (1 << 48) - 1
M64 = (1 << 64) - 1


def convert_bitmap(m):
    s = ((m << 4) | (m >> 4)) & 0xFF
    s = ((s & 0xAA) >> 1) | ((s & 0x55) << 1)
    s = ((s & 0xCC) >> 2) | ((s & 0x33) << 2)
    return s


def to_internal(w0, w1, W):
    if W == 1:
        return w0, M64  # the kernels leave k1 of one-word entries untouched (EMPTY)
    return w0 & ~(1 << 63) & M64, ((w1 << 1) | (w0 >> 63)) & M64


def from_internal(i0, i1, W):
    if W == 1:
        return i0, 0
    return (i0 | (i1 << 63)) & M64, i1 >> 1


def read_codes(packed, off, length):
    return [(int(packed[off + (i >> 2)]) >> ((i & 3) * 2)) & 3 for i in range(length)]


def occurrences(packed, offsets, lengths, k, first_read_id=0):
    """yields ((w0, w1), stamp, edge mask in the lower k-mer's orientation) per k-mer occurrence"""
    mask = (1 << (2 * k)) - 1
    for r in range(len(lengths)):
        L = int(lengths[r])
        if L < k:
            continue
        codes = read_codes(packed, int(offsets[r]), L)
        n = L - k + 1
        fw = rc = 0
        for j in range(k):
            fw |= codes[j] << (2 * j)
            rc = ((rc << 2) | (3 - codes[j])) & mask
        for p in range(n):
            if p:
                c = codes[p + k - 1]
                fw = (fw >> 2) | (c << (2 * (k - 1)))
                rc = ((rc << 2) | (3 - c)) & mask
            f = (fw & M64, fw >> 64)
            b = (rc & M64, rc >> 64)
            use_rc = b < f  # Kmer.cpp:38-47: word 0 first
            em = 0
            if p > 0:
                a = codes[p - 1]
                em |= (1 << (7 - a)) if use_rc else (1 << a)
            if p + 1 < n:
                a = codes[p + k]
                em |= (1 << (3 - a)) if use_rc else (1 << (4 + a))
            yield (b if use_rc else f), ((first_read_id + r) << 16) | p, em


class ModelEngine:
    """GraphBuilder's exchange surface on the CPU (host pointers instead of device pointers)"""

    def __init__(self, ro, k, rank=0, size=1, bloom_bits=None, W=None, first_read_id=0):
        self.ro, self.k, self.rank, self.size = ro, k, rank, size
        self.W = W if W else (1 if k <= 31 else 2)
        self.bloom_bits = ro.BLOOM_DEFAULT if bloom_bits is None else bloom_bits
        self.first_read_id = first_read_id
        self.n_reads = 0
        self.local = {}   # key -> [stamp, count, edges]
        self.table = {}
        self.final = None
        self._reads = []

    def add_reads(self, packed, offsets, lengths):
        for key, stamp, em in occurrences(packed, offsets, lengths, self.k, self.first_read_id + self.n_reads):
            e = self.local.setdefault(key, [stamp, 0, 0])
            e[0] = min(e[0], stamp)
            e[1] += 1
            e[2] |= em
        self._reads.append((packed, np.asarray(offsets), np.asarray(lengths)))
        self.n_reads += len(lengths)

    def count_local(self):
        if self.size == 1:
            self.table = self.local

    # ---- record exchange surface: here a "record" is one locally aggregated 32-byte entry and its destination
    # any function of the key (the product sends super-k-mer records by minimizer bucket: invisible in results)
    def _dest(self, key):
        return self.ro.hash2(key, self.W) % self.size

    def records_local(self, part=0, nparts=1):
        self._rec = [[] for _ in range(self.size)]
        for i, (key, (stamp, count, em)) in enumerate(sorted(self.local.items())):
            if i % nparts != part:
                continue
            i0, i1 = to_internal(key[0], key[1], self.W)
            self._rec[self._dest(key)].append((i0, i1, stamp, count | (em << 48)))
        if part == nparts - 1:
            self.local = {}

    def record_counts(self):
        return [len(r) for r in self._rec], ENTRY.itemsize

    def export_records(self, ptr, capacity):
        flat = [e for per in self._rec for e in per]
        assert len(flat) <= capacity
        out = np.array(flat, dtype=ENTRY) if flat else np.zeros(0, dtype=ENTRY)
        if len(flat):
            C.memmove(ptr, out.ctypes.data, out.nbytes)

    def count_records(self, ptr, n):
        self.local = {}
        if n:
            buf = (C.c_uint8 * (n * ENTRY.itemsize)).from_address(ptr)
            for k0, k1, stamp, meta in np.frombuffer(buf, dtype=ENTRY).tolist():
                key = from_internal(k0, k1, self.W)
                e = self.local.setdefault(key, [stamp, 0, 0])
                e[0] = min(e[0], stamp)
                e[1] += meta & COUNT_MASK
                e[2] |= (meta >> 48) & 0xFF

    def _owner(self, key):
        return self.ro.hash1(key, self.W) % self.size

    def owner_counts(self):
        c = [0] * self.size
        for key in self.local:
            c[self._owner(key)] += 1
        self._counts = c
        return c

    def export_owner_buckets(self, ptr, capacity):
        total = sum(self._counts)
        assert total <= capacity
        out = np.zeros(total, dtype=ENTRY)
        cur = np.concatenate([[0], np.cumsum(self._counts)[:-1]]).astype(np.int64)
        for key, (stamp, count, em) in self.local.items():
            o = self._owner(key)
            i0, i1 = to_internal(key[0], key[1], self.W)
            out[cur[o]] = (i0, i1, stamp, count | (em << 48))
            cur[o] += 1
        if total:
            C.memmove(ptr, out.ctypes.data, out.nbytes)

    def merge_entries(self, ptr, n):
        if n == 0:
            return
        buf = (C.c_uint8 * (n * ENTRY.itemsize)).from_address(ptr)
        for k0, k1, stamp, meta in np.frombuffer(buf, dtype=ENTRY).tolist():
            key = from_internal(k0, k1, self.W)
            e = self.table.setdefault(key, [stamp, 0, 0])
            e[0] = min(e[0], stamp)
            e[1] += meta & COUNT_MASK
            e[2] |= (meta >> 48) & 0xFF

    def resolve_admission(self, local_reads=None):
        reads = self.n_reads if local_reads is None else local_reads
        m = self.bloom_bits
        if m == self.ro.BLOOM_DEFAULT:
            m = min(16 * self.k * reads, 1 << 33)
        self.m_bits = m
        touch = {}
        bits = {}
        if m:
            for key, (stamp, _, _) in self.table.items():
                bl = self.ro.bloom_bit_indices(key, self.W, m)
                bits[key] = bl
                for b in bl:
                    if stamp < touch.get(b, M64):
                        touch[b] = stamp
        final = {}
        self.insertions = 0
        for key, (stamp, n, em) in self.table.items():
            if m:
                fp = all(touch[b] < stamp for b in bits[key])
                if fp:
                    cov = n + 1
                else:
                    self.insertions += 1
                    cov = n if n >= 2 else 0
            else:
                cov = n
            cov = min(cov, 0xFFFFFFFF)
            if cov:
                final[key] = (cov, em)
        self.set_bits = len(touch)
        self.final = final

    # ---- purge across ranks (same surface as GraphBuilder.purge_*; host pointers) ----
    @property
    def kmer_words(self):
        return self.W

    def _neighbour(self, key, bit):
        """canonical k-mer named by edge bit `bit` of stored key (bits 0-3 parents, 4-7 children)"""
        k, c = self.k, bit & 3
        K = key[0] | (key[1] << 64)
        if bit < 4:
            N = ((K << 2) | c) & ((1 << (2 * k)) - 1)
        else:
            N = (K >> 2) | (c << (2 * (k - 1)))
        n = (N & M64, N >> 64)
        rc = tuple(self.ro.complement_vertex(n, k))
        return n if (n[0], n[1]) < (rc[0], rc[1]) or n == rc else rc

    def purge_query_counts(self):
        self._queries = [[] for _ in range(self.size)]   # per owner: (canonical neighbour, key, bit)
        for key in sorted(self.final):
            cov, em = self.final[key]
            for bit in range(8):
                if em >> bit & 1:
                    nb = self._neighbour(key, bit)
                    self._queries[self._owner(nb)].append((nb, key, bit))
        return [len(q) for q in self._queries]

    def purge_export_queries(self, ptr, capacity):
        flat = [q for per in self._queries for q in per]
        assert len(flat) <= capacity
        out = np.zeros((len(flat), self.W), dtype=np.uint64)
        for i, (nb, _, _) in enumerate(flat):
            out[i] = to_internal(nb[0], nb[1], self.W)[: self.W]
        if len(flat):
            C.memmove(ptr, out.ctypes.data, out.nbytes)
        self._flat = flat

    def purge_answer(self, qptr, n, aptr):
        if n == 0:
            return
        buf = (C.c_uint8 * (n * 8 * self.W)).from_address(qptr)
        q = np.frombuffer(buf, dtype=np.uint64).reshape(n, self.W)
        ans = np.zeros(n, dtype=np.uint8)
        for i in range(n):
            key = from_internal(int(q[i, 0]), int(q[i, 1]) if self.W == 2 else 0, self.W)
            ans[i] = 1 if key in self.final else 0
        C.memmove(aptr, ans.ctypes.data, n)

    def purge_apply(self, aptr, n):
        assert n == len(self._flat)
        ans = np.frombuffer((C.c_uint8 * max(1, n)).from_address(aptr), dtype=np.uint8)[:n] if n else []
        for a, (_, key, bit) in zip(ans, self._flat):
            if not a:
                cov, em = self.final[key]
                self.final[key] = (cov, em & ~(1 << bit))

    # ---- markers across ranks (same surface as GraphBuilder.markers_*; host pointers) ----
    def markers_query_counts(self):
        self._mq = [[] for _ in range(self.size)]   # per owner: (canonical k-mer, read, position)
        r0 = 0
        for packed, offsets, lengths in self._reads:
            for key, stamp, _ in occurrences(packed, offsets, lengths, self.k, r0):
                self._mq[self._owner(key)].append((key, stamp >> 16, stamp & 0xFFFF))
            r0 += len(lengths)
        return [len(q) for q in self._mq]

    def markers_export_queries(self, ptr, capacity):
        flat = [q for per in self._mq for q in per]
        assert len(flat) <= capacity
        out = np.zeros((len(flat), self.W), dtype=np.uint64)
        for i, (key, _, _) in enumerate(flat):
            out[i] = to_internal(key[0], key[1], self.W)[: self.W]
        if len(flat):
            C.memmove(ptr, out.ctypes.data, out.nbytes)
        self._mflat = flat

    def markers_answer(self, qptr, n, aptr):
        if n == 0:
            return
        buf = (C.c_uint8 * (n * 8 * self.W)).from_address(qptr)
        q = np.frombuffer(buf, dtype=np.uint64).reshape(n, self.W)
        ans = np.zeros(n, dtype=np.uint32)
        for i in range(n):
            key = from_internal(int(q[i, 0]), int(q[i, 1]) if self.W == 2 else 0, self.W)
            ans[i] = self.final[key][0] if key in self.final else 0
        C.memmove(aptr, ans.ctypes.data, 4 * n)

    def markers_apply(self, aptr, n, fptr, rptr):
        """IndexerWorker.cpp:97-248 on the coverages that came back"""
        assert n == len(self._mflat)
        ans = np.frombuffer((C.c_uint8 * max(4, 4 * n)).from_address(aptr), dtype=np.uint32)[:n] if n else []
        lengths = np.concatenate([l for _, _, l in self._reads]) if self._reads else np.zeros(0, dtype=np.uint32)
        cov = [dict() for _ in range(len(lengths))]
        for a, (_, read, pos) in zip(ans, self._mflat):
            cov[read][pos] = int(a)
        fwd = np.full(max(1, len(lengths)), -1, dtype=np.int32)
        rev = np.full(max(1, len(lengths)), -1, dtype=np.int32)
        for r, L in enumerate(lengths.tolist()):
            n_k = L - self.k + 1 if L >= self.k else 0
            best, best_score = -1, 0
            for value in range(10, 0, -1):
                score = sum(1 for p in range(n_k) if cov[r][p] <= value or cov[r][p] >= 2 * value)
                if best == -1 or score > best_score:
                    best, best_score = value, score
            good = [p for p in range(n_k) if cov[r][p] >= 2 * best]
            if good:
                fwd[r] = good[0]
                rev[r] = L - self.k - good[-1]
        C.memmove(fptr, fwd.ctypes.data, 4 * len(lengths))
        C.memmove(rptr, rev.ctypes.data, 4 * len(lengths))

    def records(self, with_edges=True):
        dt = self.ro.record_dtype(self.W)
        out = np.zeros(len(self.final), dtype=dt)
        for i, key in enumerate(sorted(self.final)):
            cov, em = self.final[key]
            out[i] = (tuple(key[: self.W]), cov, em if with_edges else 0)
        return out

    def histogram(self):
        h = {}
        for cov, _ in self.final.values():
            h[cov] = h.get(cov, 0) + 2
        ks = sorted(h)
        return np.asarray(ks, dtype=np.uint32), np.asarray([h[c] for c in ks], dtype=np.uint64)
