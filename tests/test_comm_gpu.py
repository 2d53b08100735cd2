"""The exchanges between ranks INSIDE the library (rg_comm_*, rg_exchange_*, rg_*_exchange: include/raygraph.h), as a C++
Ray rank would call them.

* R ranks as R host threads of this process, each with its own handle and its own communicator over the caller's-transport
  interface (rg_transport): the callbacks move exactly the bytes an all-to-all(v) moves, through device copies, with a barrier
  where a collective synchronises.  The whole C path -- record or entry route, owner exchange, admission, merged histogram,
  markers in slices, purge, allgather -- must equal the R-rank oracle.  (RCCL wants one GPU per rank; a GPU box has one.)
* The same path on a real RCCL communicator of one rank: rg_comm_unique_id, ncclCommInitRank, grouped ncclSend / ncclRecv of
  the rank's records / entries / queries to itself, on RCCL's own stream.
"""
import ctypes as C
import threading

import numpy as np
import pytest

from helpers import assert_records_equal, oracle_run
from ray_amd.exchange import partition_reads
from ray_amd.synth import make_reads_numpy

pytestmark = pytest.mark.gpu


class ThreadWorld:
    """what the ranks share: a barrier and one slot per rank where it posts what it sends"""

    def __init__(self, size):
        self.size = size
        self.barrier = threading.Barrier(size, timeout=120)
        self.words = [None] * size
        self.posted = [None] * size
        self.hip = C.CDLL("libamdhip64.so.7")   # the HIP runtime this process already holds
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipDeviceSynchronize.argtypes = []

    def callbacks(self, rank):
        size = self.size

        def alltoall_u64(send, n):
            self.words[rank] = send
            self.barrier.wait()
            got = []
            for src in range(size):
                got += self.words[src][rank * n:(rank + 1) * n]
            self.barrier.wait()
            return got

        def alltoallv(d_send, scount, d_recv, rcount, unit):
            self.posted[rank] = (d_send, scount, unit)
            self.barrier.wait()
            at = 0
            for src in range(size):
                p, sc, u = self.posted[src]
                assert u == unit and sc[rank] == rcount[src], "rank %d expects %d items from %d, which sends %d" % (rank, rcount[src], src, sc[rank])
                nbytes = sc[rank] * unit
                if nbytes:
                    assert self.hip.hipMemcpy(d_recv + at, p + sum(sc[:rank]) * unit, nbytes, 3) == 0   # hipMemcpyDeviceToDevice
                at += nbytes
            assert self.hip.hipDeviceSynchronize() == 0
            self.barrier.wait()   # nobody's send buffer goes before everybody has copied out of it

        return alltoall_u64, alltoallv


def run_world(packed, offsets, lengths, k, nranks, bloom, route, parts, marker_parts, sweep=0, W=0):
    """every rank's results: dict(rank -> ...); raises the first rank's exception"""
    import ray_amd
    from ray_amd.graph import RG_BLOOM_DEFAULT
    n = len(lengths)
    world = ThreadWorld(nranks)
    out, errors = {}, []

    def rank_main(r):
        try:
            first, last = partition_reads(n, r, nranks)
            b = ray_amd.GraphBuilder(k, rank=r, size=nranks, bloom_bits=RG_BLOOM_DEFAULT if bloom is None else bloom, kmer_words=W,
                                     first_read_id=first, sweep=sweep)
            b.add_reads(packed, offsets[first:last], lengths[first:last])
            comm = ray_amd.Comm.transport(*world.callbacks(r), rank=r, size=nranks)
            b.count_kmers_exchange(comm, route=route, parts=parts, local_reads=last - first)   # RAY_SLAVE_MODE_ADD_VERTICES
            res = {"stats": b.stats(), "pass1": b.export()}
            res["hist"] = b.histogram_exchange(comm)                                         # RAY_SLAVE_MODE_SEND_DISTRIBUTION
            b.add_edges()                                                                       # RAY_SLAVE_MODE_ADD_EDGES
            res["pass2"] = b.export()
            res["markers"] = b.read_markers_exchange(comm, last - first, parts=marker_parts)   # SequencesIndexer
            b.purge_edges_exchange(comm)                                                       # RAY_SLAVE_MODE_PURGE_NULL_EDGES
            res["purged"] = b.export()
            res["sizes"] = b.allgather_u64(comm, b.size_())
            res["comm"] = comm.stats()
            out[r] = res
            b.close()
            comm.close()
        except BaseException as e:  # noqa: BLE001
            errors.append((r, e))
            world.barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        real = [e for e in errors if not isinstance(e[1], threading.BrokenBarrierError)] or errors
        raise AssertionError("rank %d failed: %r" % real[0])
    return out


def check_world(out, want, nranks, n_reads, what):
    assert sum(out[r]["stats"]["occurrences"] for r in range(nranks)) == want["occ"], what
    g = want["graph"]
    for r in range(nranks):
        st = out[r]["stats"]
        assert (st["bloom_bits"], st["bloom_set_bits"], st["bloom_insertions"]) == want["bloom"][r], "%s rank %d Bloom statistics" % (what, r)
        assert np.array_equal(out[r]["hist"][0], want["hist"][0]) and np.array_equal(out[r]["hist"][1], want["hist"][1]), "%s rank %d merged histogram" % (what, r)
        assert out[r]["sizes"] == [g.size(q) for q in range(nranks)], "%s rank %d allgather of the GridTable sizes" % (what, r)
    wf, wr = want["markers"]
    gf = np.concatenate([out[r]["markers"][0] for r in range(nranks)])
    gr = np.concatenate([out[r]["markers"][1] for r in range(nranks)])
    assert len(gf) == n_reads and np.array_equal(gf, wf) and np.array_equal(gr, wr), what + ": markers"


@pytest.mark.parametrize("k,nranks,bloom,route,parts,marker_parts", [
    (31, 2, None, "records", 3, 2), (31, 4, 60000, "records", 1, 1), (31, 8, 0, "records", 2, 3), (63, 2, 90000, "entries", 1, 2),
    (21, 3, 20000, "entries", 1, 1), (63, 3, 90000, "records", 4, 1)])
def test_c_exchange_path_matches_oracle(oracle, k, nranks, bloom, route, parts, marker_parts):
    packed, offsets, lengths = make_reads_numpy(genome=15000, read_len=100 if k < 63 else 150, n_reads=5003, seed=k + nranks, error=0.015)
    g = oracle.Graph(k, nranks=nranks, bloom_bits=oracle.BLOOM_DEFAULT if bloom is None else bloom)
    g.set_reads(packed, offsets, lengths)
    g.count_kmers()
    want = {"graph": g, "occ": g.occurrences(), "hist": g.histogram(),
            "bloom": [(g.bloom_bits(r), g.bloom_set_bits(r), g.bloom_insertions(r)) for r in range(nranks)]}
    pass1 = [g.export(r) for r in range(nranks)]
    g.add_edges()
    pass2 = [g.export(r) for r in range(nranks)]
    want["markers"] = g.read_markers()
    want["sizes"] = [g.size(r) for r in range(nranks)]
    for sweep in (1 | 16 | (10 << 8), 0):
        out = run_world(packed, offsets, lengths, k, nranks, bloom, route, parts, marker_parts, sweep=sweep)
        what = "%s route, %d ranks, sweep 0x%x" % (route, nranks, sweep)
        check_world(out, want, nranks, len(lengths), what)
        for r in range(nranks):
            assert_records_equal(out[r]["pass1"], pass1[r], "%s rank %d after pass 1" % (what, r))
            assert_records_equal(out[r]["pass2"], pass2[r], "%s rank %d after pass 2" % (what, r))
        assert sum(out[r]["comm"]["bytes_sent"] for r in range(nranks)) == sum(out[r]["comm"]["bytes_received"] for r in range(nranks)) > 0
    g.purge_edges()
    for r in range(nranks):
        assert_records_equal(out[r]["purged"], g.export(r), "rank %d after the purge" % r)


def test_c_exchange_with_ranks_that_have_no_reads(oracle):
    """fewer reads than ranks (SequencesLoader.cpp:287-297 gives them all to the last rank): every rank still takes part in
    every collective with nothing to send"""
    packed, offsets, lengths = make_reads_numpy(genome=300, read_len=80, n_reads=3, seed=5, error=0.0)
    want = oracle_run(oracle, packed, offsets, lengths, 21, 4, 0)
    want["markers"] = want["graph"].read_markers()
    for route in ("records", "entries"):
        out = run_world(packed, offsets, lengths, 21, 4, 0, route, 2, 2)
        check_world(out, want, 4, len(lengths), route)
        for r in range(4):
            assert_records_equal(out[r]["pass2"], want["graph"].export(r), "%s rank %d" % (route, r))


def test_markers_one_very_long_read_among_short_ones(oracle):
    """the multi-rank marker route holds memory per k-mer that exists (an exclusive scan of the reads' k-mer counts), not
    per read x longest read: one 60 000-base read among 20 000 reads of 60 (a dense matrix would be 20 001 x 59 980
    cells here, and 10 TB at 40 M reads); slices bound the buffers further"""
    rng = np.random.default_rng(11)
    genome = rng.integers(0, 4, size=70000, dtype=np.uint8)
    from ray_amd.synth import pack_codes
    short_starts = rng.integers(0, 70000 - 60, size=20000)
    sp, so, sl = pack_codes(genome[short_starts[:, None] + np.arange(60)[None, :]], np.full(20000, 60, dtype=np.uint32))
    lp, _, ll = pack_codes(genome[None, 5000:65000], np.array([60000], dtype=np.uint32))
    at = 1234                                        # the long read sits between short ones
    cut = int(so[at])
    packed = np.concatenate([sp[:cut], lp, sp[cut:]])
    offsets = np.concatenate([so[:at], [cut], so[at:] + len(lp)]).astype(np.uint64)
    lengths = np.concatenate([sl[:at], ll, sl[at:]]).astype(np.uint32)
    k, nranks = 21, 2
    want = oracle_run(oracle, packed, offsets, lengths, k, nranks, 0)
    want["markers"] = want["graph"].read_markers()
    out = run_world(packed, offsets, lengths, k, nranks, 0, "records", 1, 3)
    check_world(out, want, nranks, len(lengths), "long read")
    assert (want["markers"][0] >= 0).sum() > 100


@pytest.mark.parametrize("k,parts,route", [(31, 3, "records"), (63, 1, "records"), (31, 1, "entries")])
def test_c_exchange_path_on_a_one_rank_rccl_communicator(oracle, k, parts, route):
    import ray_amd
    packed, offsets, lengths = make_reads_numpy(genome=30000, read_len=150, n_reads=9000, seed=70 + k, error=0.01)
    bloom = 2000000
    want = oracle_run(oracle, packed, offsets, lengths, k, 1, bloom, purge=True)
    comm = ray_amd.Comm.rccl(ray_amd.Comm.unique_id(), 0, 1, 0)
    b = ray_amd.GraphBuilder(k, bloom_bits=bloom, sweep=1)
    b.add_reads(packed, offsets, lengths)
    if route == "records":
        b.exchange_records(comm, parts)          # grouped ncclSend / ncclRecv of the 20- / 40-byte records, slice by slice
        b.exchange_owner_buckets(comm)           # ... of the 32-byte entries
        b.resolve_admission(len(lengths))
    else:
        b.count_kmers_exchange(comm, route="entries", local_reads=len(lengths))   # one rank, entry route: nothing to move
    cov, cnt = b.histogram_exchange(comm)
    assert np.array_equal(cov, want["hist"][0]) and np.array_equal(cnt, want["hist"][1])
    assert_records_equal(b.export(), want["pass1"], "after pass 1")
    b.add_edges()
    assert_records_equal(b.export(), want["pass2"], "after pass 2")
    wf, wr = want["graph"].read_markers()
    gf, gr = b.read_markers_exchange(comm, len(lengths), parts=2)    # keys out, 4-byte coverages back
    assert np.array_equal(gf, wf) and np.array_equal(gr, wr)
    b.purge_edges_exchange(comm)                                      # keys out, one-byte answers back
    assert_records_equal(b.export(), want["purged"], "after purge")
    assert b.allgather_u64(comm, 42) == [42]
    st = comm.stats()
    assert st["collectives"] >= (4 if route == "entries" else 6) and st["bytes_sent"] == 0   # a rank's traffic to itself is not counted
    b.close()
    comm.close()


def test_comm_errors_are_reported(oracle):
    import ray_amd
    packed, offsets, lengths = make_reads_numpy(genome=3000, read_len=80, n_reads=500, seed=5, error=0.0)
    b = ray_amd.GraphBuilder(21, rank=0, size=2)
    b.add_reads(packed, offsets, lengths)
    with pytest.raises(ray_amd.RayGraphError) as e:     # two ranks, no communicator
        b.exchange_records(None, 1)
    assert e.value.code == -1 and "communicator" in str(e.value)
    comm = ray_amd.Comm.transport(lambda s, n: s, lambda *a: None, rank=1, size=2)   # another rank's communicator
    with pytest.raises(ray_amd.RayGraphError) as e:
        b.exchange_records(comm, 1)
    assert e.value.code == -1
    comm.close()

    def broken(*a):
        raise RuntimeError("link down")
    comm = ray_amd.Comm.transport(broken, broken, rank=0, size=2)
    with pytest.raises(ray_amd.RayGraphError) as e:     # the caller's collective fails: RG_ERR_COMM, not a hang or a crash
        b.exchange_records(comm, 1)
    assert e.value.code == -7
    comm.close()
    b.close()
