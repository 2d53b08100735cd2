"""The owner exchange on real hardware without needing N GPUs: R GraphBuilders (rank r of R) share
device 0 in one process and swap their per-owner buckets through device buffers, exactly the bytes an
all-to-all would move.  Result must equal the R-rank oracle (Bloom filter per rank sized from that
rank's own read count, arrival order = source rank order)."""
import numpy as np
import pytest

from helpers import assert_records_equal, oracle_run
from ray_amd.exchange import partition_reads
from ray_amd.synth import make_reads_numpy

pytestmark = pytest.mark.gpu


def run_ranks(packed, offsets, lengths, k, nranks, bloom, W=0, sweep=0, route="entries", parts=1, one_merge=False):
    """route "entries": every rank aggregates its own reads, then the entries go to their owners;
    route "records": the super-k-mer records go to the rank that aggregates their bucket first (the preferred
    multi-GPU route), then the entries go to their owners.
    one_merge: an owner gets everything it receives in ONE rg_merge_entries call ("adopt": one rg_adopt_entries call after
    rg_release_local, as exchange_owner_buckets delivers it: no copy, admission runs in the receive buffer)
    (record route: the entries are unique and are kept as a log that is admitted before anything is inserted);
    otherwise one call per source rank (the first is kept as a log, the second sends it to the table: classic merge)"""
    import torch
    import ray_amd
    from ray_amd.graph import RG_BLOOM_DEFAULT, RG_ENTRY_BYTES
    n = len(lengths)
    builders = []
    for r in range(nranks):
        first, last = partition_reads(n, r, nranks)
        b = ray_amd.GraphBuilder(k, rank=r, size=nranks, bloom_bits=RG_BLOOM_DEFAULT if bloom is None else bloom,
                                 kmer_words=W, first_read_id=first, sweep=sweep)
        b.add_reads(packed, offsets[first:last], lengths[first:last])
        builders.append(b)
    if route == "records":
        sent = []   # parts > 1: every rank cuts its reads in slices whose records travel separately (any order on arrival)
        for b in builders:
            for part in range(parts):
                b.records_local(part, parts)
                counts, unit = b.record_counts()
                buf = torch.empty(max(1, sum(counts)) * unit, dtype=torch.uint8, device="cuda")
                b.export_records(buf.data_ptr(), max(1, sum(counts)))
                sent.append((counts, buf, unit))
        assert sum(sum(c) for c, _, _ in sent) > 0
        for dst, b in enumerate(builders):
            parts = []
            for counts, buf, unit in sent:
                start = sum(counts[:dst])
                parts.append(buf[start * unit:(start + counts[dst]) * unit])
            recv = torch.cat(parts) if parts else torch.empty(0, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            b.count_records(recv.data_ptr() if recv.numel() else 0, recv.numel() // sent[0][2])
    else:
        for b in builders:
            b.count_local()
    sends = []
    for b in builders:
        counts = b.owner_counts()
        buf = torch.empty(max(1, sum(counts)) * RG_ENTRY_BYTES, dtype=torch.uint8, device="cuda")
        b.export_owner_buckets(buf.data_ptr(), max(1, sum(counts)))
        torch.cuda.synchronize()
        if one_merge == "adopt":
            b.release_local()   # the aggregation goes before the exchange, not at the first delivery
        sends.append((counts, buf))
    for dst, b in enumerate(builders):
        got = []
        for counts, buf in sends:
            start = sum(counts[:dst])
            part = buf[start * RG_ENTRY_BYTES:(start + counts[dst]) * RG_ENTRY_BYTES]
            if one_merge:
                got.append(part)
            else:
                b.merge_entries(part.data_ptr(), counts[dst])
        if one_merge:
            allgot = torch.cat(got)
            torch.cuda.synchronize()
            if one_merge == "adopt":   # admission in the receive buffer itself
                b.adopt_entries(allgot.data_ptr() if allgot.numel() else 0, allgot.numel() // RG_ENTRY_BYTES, keepalive=allgot)
            else:
                b.merge_entries(allgot.data_ptr() if allgot.numel() else 0, allgot.numel() // RG_ENTRY_BYTES)
    for r, b in enumerate(builders):
        first, last = partition_reads(n, r, nranks)
        b.resolve_admission(last - first)
        b.add_edges()
    return builders


@pytest.mark.parametrize("k,nranks,bloom", [(31, 2, None), (31, 4, 60000), (31, 8, 0), (63, 2, 90000), (21, 3, 20000)])
def test_multirank_matches_oracle(oracle, k, nranks, bloom):
    packed, offsets, lengths = make_reads_numpy(genome=15000, read_len=100 if k < 63 else 150, n_reads=5001,
                                                seed=k + nranks, error=0.015)
    want = oracle_run(oracle, packed, offsets, lengths, k, nranks, bloom)
    for sweep, route, parts, one in ((1 | 16 | (17 << 8), "entries", 1, False), (2 | 32, "entries", 1, True), (1 | 16 | (10 << 8), "records", 1, True),
                                     (1 | 32 | (10 << 8), "records", 1, True), (1 | 16 | (10 << 8) | 0x80000, "records", 1, True),
                                     (1 | 16 | (10 << 8), "records", 3, False), (1 | 16 | 0x2000000, "records", 1, True), (1 | 16 | (11 << 8), "records", 2, "adopt"),
                                     (1 | 32 | (9 << 8), "entries", 1, "adopt"), (2 | 32, "records", 1, False)):
        builders = run_ranks(packed, offsets, lengths, k, nranks, bloom, sweep=sweep, route=route, parts=parts, one_merge=one)
        check_ranks(builders, want, route)
    # SequencesIndexer's markers across ranks (IndexerWorker.cpp:67-248): k-mers to their owners, coverages back
    wf, wr = want["graph"].read_markers()
    gf, gr = markers_ranks(builders, len(lengths))
    assert (wf >= 0).sum() > 100 and (wf < 0).sum() > 0, "both outcomes must occur in the test data"
    assert np.array_equal(gf, wf) and np.array_equal(gr, wr)
    # RAY_SLAVE_MODE_PURGE_NULL_EDGES across ranks: keys to the owners, one-byte answers back
    want["graph"].purge_edges()
    purge_ranks(builders)
    for r, b in enumerate(builders):
        assert_records_equal(b.export(), want["graph"].export(r), "rank %d after purge" % r)


def markers_ranks(builders, n_reads):
    """rg_markers_*: the bytes two all-to-all(v) would move, through device buffers; returns the markers of all reads in
    global read order (rank r holds the reads of partition_reads(n, r, R))"""
    import torch
    R = len(builders)
    unit = 8 * builders[0].kmer_words
    asked = []
    for b in builders:
        counts = b.markers_query_counts()
        keys = torch.empty(max(1, sum(counts)) * unit, dtype=torch.uint8, device="cuda")
        b.markers_export_queries(keys.data_ptr(), max(1, sum(counts)))
        asked.append((counts, keys))
    answers = [[None] * R for _ in range(R)]  # answers[src][dst]
    for dst, b in enumerate(builders):
        for src, (counts, keys) in enumerate(asked):
            start, n = sum(counts[:dst]), counts[dst]
            part = keys[start * unit:(start + n) * unit].clone()
            ans = torch.full((max(1, n),), 0x7FFFFFFF, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()   # the builder works on a stream of its own: torch's copies and fills must have landed
            b.markers_answer(part.data_ptr(), n, ans.data_ptr())
            answers[src][dst] = ans[:n]
    torch.cuda.synchronize()
    fwd, rev = [], []
    for src, b in enumerate(builders):
        first, last = partition_reads(n_reads, src, R)
        total = sum(asked[src][0])
        back = torch.cat(answers[src]) if total else torch.zeros(1, dtype=torch.int32, device="cuda")
        f = torch.full((max(1, last - first),), -7, dtype=torch.int32, device="cuda")
        r = torch.full((max(1, last - first),), -7, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        b.markers_apply(back.data_ptr(), total, f.data_ptr(), r.data_ptr())
        fwd.append(f[:last - first].cpu().numpy())
        rev.append(r[:last - first].cpu().numpy())
    return np.concatenate(fwd), np.concatenate(rev)


def purge_ranks(builders):
    import torch
    R = len(builders)
    unit = 8 * builders[0].kmer_words
    asked = []
    for b in builders:
        counts = b.purge_query_counts()
        keys = torch.empty(max(1, sum(counts)) * unit, dtype=torch.uint8, device="cuda")
        b.purge_export_queries(keys.data_ptr(), max(1, sum(counts)))
        asked.append((counts, keys))
    answers = [[None] * R for _ in range(R)]  # answers[src][dst]
    for dst, b in enumerate(builders):
        for src, (counts, keys) in enumerate(asked):
            start, n = sum(counts[:dst]), counts[dst]
            part = keys[start * unit:(start + n) * unit].clone()
            ans = torch.zeros(max(1, n), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()   # (as above)
            b.purge_answer(part.data_ptr(), n, ans.data_ptr())
            answers[src][dst] = ans[:n]
    torch.cuda.synchronize()
    for src, b in enumerate(builders):
        back = torch.cat(answers[src]) if sum(asked[src][0]) else torch.zeros(1, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        b.purge_apply(back.data_ptr(), sum(asked[src][0]))


def check_ranks(builders, want, route=""):
    assert sum(b.stats()["occurrences"] for b in builders) == want["occ"]
    for r, b in enumerate(builders):
        assert_records_equal(b.export(), want["graph"].export(r), "%s rank %d" % (route, r))
        st = b.stats()
        assert (st["bloom_bits"], st["bloom_set_bits"], st["bloom_insertions"]) == want["bloom"][r]
    # merged histogram, as rank 0 accumulates RAY_MPI_TAG_COVERAGE_DATA (MessageProcessor.cpp:1048-1061)
    merged = {}
    for b in builders:
        for c, n in zip(*b.histogram()):
            merged[int(c)] = merged.get(int(c), 0) + int(n)
    assert sorted(merged) == want["hist"][0].tolist()
    assert [merged[c] for c in sorted(merged)] == want["hist"][1].tolist()


def test_ranks_without_reads(oracle):
    """fewer reads than ranks: SequencesLoader gives them all to the last rank (SequencesLoader.cpp:287-297); the others
    still take part in every exchange -- records, entries, markers, purge -- with nothing to send"""
    packed, offsets, lengths = make_reads_numpy(genome=300, read_len=80, n_reads=3, seed=5, error=0.0)
    want = oracle_run(oracle, packed, offsets, lengths, 21, 4, 0)
    for route in ("records", "entries"):
        builders = run_ranks(packed, offsets, lengths, 21, 4, 0, route=route, one_merge=True)
        check_ranks(builders, want, route)
    wf, wr = want["graph"].read_markers()
    gf, gr = markers_ranks(builders, len(lengths))
    assert np.array_equal(gf, wf) and np.array_equal(gr, wr)
    want["graph"].purge_edges()
    purge_ranks(builders)
    for r, b in enumerate(builders):
        assert_records_equal(b.export(), want["graph"].export(r), "rank %d after purge" % r)
