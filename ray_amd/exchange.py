"""Owner exchange between ranks: the one data-path collective of each pass.

Replaces RayPlatform's per-message RAY_MPI_TAG_VERTICES_DATA / OUT_EDGES_DATA / IN_EDGES_DATA
point-to-point traffic (KmerAcademyBuilder.cpp:187-196, VerticesExtractor.cpp:175-251) with one
all-to-all(v) of per-owner buckets: every rank aggregates its own reads per canonical k-mer,
groups the 32-byte entries by owner = hash1(lower) % size (Kmer.cpp:165-170), and the owners merge
what they receive.

On the GPU box the exchanges run INSIDE the library (rg_comm_* / rg_exchange_*: grouped ncclSend / ncclRecv over xGMI on
RCCL's own stream, ray_amd/csrc/rg_comm.cpp): `create_comm` makes this rank's communicator -- torch.distributed only carries
the 256-byte id from rank 0 to the others, as one Ray message would -- and GraphBuilder.count_kmers_exchange /
histogram_exchange / purge_edges_exchange / read_markers_exchange take it.  The functions below it are the same exchanges
driven step by step from Python over torch.distributed collectives: the "gloo" route of the CPU tests and of rehearsals
with more ranks than GPUs (host-staged), where RCCL cannot be used.

`engine` is anything with the GraphBuilder exchange surface: size, owner_counts(),
export_owner_buckets(ptr, capacity), merge_entries(ptr, n).
"""
import torch
import torch.distributed as dist

from ._lib import RG_ENTRY_BYTES


def create_comm(rank, size, device_index, group=None):
    """this rank's rg_comm over RCCL: rank 0 makes the id (rg_comm_unique_id), the process group carries it, every rank
    joins (rg_comm_create_rccl is a collective)"""
    from .graph import Comm
    ids = [Comm.unique_id() if rank == 0 else None]
    if size > 1:
        dist.broadcast_object_list(ids, src=0, group=group)
    return Comm.rccl(ids[0], rank, size, device_index)


def partition_reads(total_reads, rank, size):
    """SequencesLoader.cpp:287-297: contiguous ranges of floor(T/size) reads, the last rank takes the rest."""
    per = total_reads // size
    first = rank * per
    last = total_reads if rank == size - 1 else first + per
    return first, last


def _staged(group):
    """gloo cannot move device memory: stage through the host (tests / single-GPU rehearsals only)"""
    return dist.get_backend(group) == "gloo"


def _a2a(recv, src, out_counts, in_counts, unit, group, async_op=False):
    """all-to-all(v) of `unit`-byte items held in uint8 tensors.  The items travel as the widest integer type that divides
    them (32-byte entries as int64, 20-byte records as int32), so that no per-peer element count comes near 2^31
    (3.2 GB of entries, 6.9 GB of records to one peer at N = 2)."""
    for es, dt in ((8, torch.int64), (4, torch.int32), (2, torch.int16)):
        if unit % es == 0 and recv.numel() % es == 0 and src.numel() % es == 0 and recv.storage_offset() % es == 0 and src.storage_offset() % es == 0:
            q = unit // es
            return dist.all_to_all_single(recv.view(dt), src.view(dt), output_split_sizes=[c * q for c in out_counts],
                                          input_split_sizes=[c * q for c in in_counts], group=group, async_op=async_op)
    return dist.all_to_all_single(recv, src, output_split_sizes=[c * unit for c in out_counts],
                                  input_split_sizes=[c * unit for c in in_counts], group=group, async_op=async_op)


def exchange_owner_buckets(engine, group=None, device="cuda", always=False):
    """all-to-all(v) of the per-owner buckets; returns (entries sent, entries received).
    always: run the exchange on one rank too (after exchange_records: the rank's entries travel to itself)."""
    size = engine.size
    if size == 1 and not always:
        return 0, 0  # the local table already is the partition; nothing moves
    counts = engine.owner_counts()
    total = sum(counts)
    send = torch.empty(max(1, total) * RG_ENTRY_BYTES, dtype=torch.uint8, device=device)
    engine.export_owner_buckets(send.data_ptr(), max(1, total))
    if hasattr(engine, "release_local"):
        engine.release_local()   # the buckets hold everything the local aggregation knew: its memory is free for the receive buffer
    staged = _staged(group) and str(device) != "cpu"
    xdev = "cpu" if staged else device
    send_counts = torch.tensor(counts, dtype=torch.int64, device=xdev)
    recv_counts = torch.empty(size, dtype=torch.int64, device=xdev)
    dist.all_to_all_single(recv_counts, send_counts, group=group)
    recv_list = [int(x) for x in recv_counts.cpu().tolist()]
    n_recv = sum(recv_list)
    recv = torch.empty(max(1, n_recv) * RG_ENTRY_BYTES, dtype=torch.uint8, device=xdev)
    src = send[: total * RG_ENTRY_BYTES].cpu() if staged else send[: total * RG_ENTRY_BYTES]
    _a2a(recv[: n_recv * RG_ENTRY_BYTES], src, recv_list, counts, RG_ENTRY_BYTES, group)
    if staged:
        recv = recv.to(device)
    if str(device) != "cpu":
        torch.cuda.current_stream().synchronize()
    del send, src
    if hasattr(engine, "adopt_entries"):
        engine.adopt_entries(recv.data_ptr(), n_recv, keepalive=recv)   # no copy: admission runs in the receive buffer
    else:
        engine.merge_entries(recv.data_ptr(), n_recv)
    return total, n_recv


def _all_to_all_bytes(send, send_counts, unit, group, device, staged):
    """all-to-all(v) of `unit`-byte items; returns (received bytes on `device`, received counts)"""
    size = len(send_counts)
    xdev = "cpu" if staged else device
    sc = torch.tensor(send_counts, dtype=torch.int64, device=xdev)
    rc = torch.empty(size, dtype=torch.int64, device=xdev)
    dist.all_to_all_single(rc, sc, group=group)
    recv_counts = [int(x) for x in rc.cpu().tolist()]
    n_send, n_recv = sum(send_counts), sum(recv_counts)
    recv = torch.empty(max(1, n_recv) * unit, dtype=torch.uint8, device=xdev)
    src = send[: n_send * unit].cpu() if staged else send[: n_send * unit]
    _a2a(recv[: n_recv * unit], src, recv_counts, send_counts, unit, group)
    if staged:
        recv = recv.to(device)
    return recv, recv_counts


def exchange_records(engine, group=None, device="cuda", parts=1):
    """first exchange of the record route: the local reads cut into super-k-mer records go to the rank that
    aggregates their bucket, which then counts them; follow with exchange_owner_buckets().
    parts > 1: the reads are cut in `parts` slices; the all-to-all of one slice's records runs (asynchronously, on
    the collective's own stream) while the kernels of the next slice produce theirs.
    Returns (records sent, records received)."""
    size = engine.size
    staged = _staged(group) and str(device) != "cpu"
    xdev = "cpu" if staged else device
    pending = []          # (work, send buffer, receive view): kept alive until the collective has finished
    big, big_used, unit = None, 0, 0
    sent = 0
    for part in range(parts):
        engine.records_local(part, parts)
        counts, unit = engine.record_counts()
        total = sum(counts)
        send = torch.empty(max(1, total) * unit, dtype=torch.uint8, device=device)
        engine.export_records(send.data_ptr(), max(1, total))   # returns once the buffer is complete
        sent += total
        sc = torch.tensor(counts, dtype=torch.int64, device=xdev)
        rc = torch.empty(size, dtype=torch.int64, device=xdev)
        dist.all_to_all_single(rc, sc, group=group)
        recv_counts = [int(x) for x in rc.cpu().tolist()]
        n_recv = sum(recv_counts)
        if big is None:   # one receive buffer for all parts (sized from the first), so that nothing is copied together later
            big = torch.empty(max(1, int(n_recv * parts * 1.1) + 4096 * parts) * unit, dtype=torch.uint8, device=xdev)
        if big_used + n_recv * unit <= big.numel():
            recv = big[big_used: big_used + n_recv * unit]
            big_used += n_recv * unit
            in_big = True
        else:
            recv = torch.empty(max(1, n_recv) * unit, dtype=torch.uint8, device=xdev)[: n_recv * unit]
            in_big = False
        src = send[: total * unit].cpu() if staged else send[: total * unit]
        work = _a2a(recv, src, recv_counts, counts, unit, group, async_op=True)
        pending.append((work, src, recv, in_big))
    for work, _, _, _ in pending:
        work.wait()
    if all(p[3] for p in pending):
        allrec = big[:big_used]
    else:
        allrec = torch.cat([big[:big_used]] + [p[2] for p in pending if not p[3]])
    n_all = allrec.numel() // unit if unit else 0
    if staged:
        allrec = allrec.to(device)
    if str(device) != "cpu":
        torch.cuda.current_stream().synchronize()
    del pending
    engine.count_records(allrec.data_ptr(), n_all)
    return sent, n_all


def purge_edges_distributed(engine, group=None, device="cuda", always=False):
    """RAY_SLAVE_MODE_PURGE_NULL_EDGES across ranks: one all-to-all of neighbour keys, one of one-byte
    answers back (EdgePurgerWorker.cpp:28-102 asks one key per message, MessageProcessor.cpp:1224-1256
    answers with the coverage).  Returns the number of queries this rank asked."""
    if engine.size == 1 and not always:
        engine.purge_edges()
        return 0
    unit = 8 * engine.kmer_words
    counts = engine.purge_query_counts()
    total = sum(counts)
    keys = torch.empty(max(1, total) * unit, dtype=torch.uint8, device=device)
    engine.purge_export_queries(keys.data_ptr(), max(1, total))
    staged = _staged(group) and str(device) != "cpu"
    asked, asked_counts = _all_to_all_bytes(keys, counts, unit, group, device, staged)
    n_asked = sum(asked_counts)
    answers = torch.zeros(max(1, n_asked), dtype=torch.uint8, device=device)
    if str(device) != "cpu":
        torch.cuda.current_stream().synchronize()
    engine.purge_answer(asked.data_ptr(), n_asked, answers.data_ptr())
    back, back_counts = _all_to_all_bytes(answers, asked_counts, 1, group, device, staged)
    assert back_counts == counts
    if str(device) != "cpu":
        torch.cuda.current_stream().synchronize()
    engine.purge_apply(back.data_ptr(), total)
    return total


def read_markers_distributed(engine, n_reads, group=None, device="cuda"):
    """SequencesIndexer marker selection across ranks (IndexerWorker.cpp:67-248): one all-to-all of the k-mers of this
    rank's reads to their owners, one of 4-byte coverages back (the reference sends one
    RAY_MPI_TAG_REQUEST_VERTEX_COVERAGE message per k-mer, :76-96).  Returns (forward, reverse) numpy arrays, -1 = none."""
    unit = 8 * engine.kmer_words
    counts = engine.markers_query_counts()
    total = sum(counts)
    keys = torch.empty(max(1, total) * unit, dtype=torch.uint8, device=device)
    engine.markers_export_queries(keys.data_ptr(), max(1, total))
    staged = _staged(group) and str(device) != "cpu"
    asked, asked_counts = _all_to_all_bytes(keys, counts, unit, group, device, staged)
    n_asked = sum(asked_counts)
    answers = torch.zeros(max(1, n_asked) * 4, dtype=torch.uint8, device=device)
    if str(device) != "cpu":
        torch.cuda.current_stream().synchronize()
    engine.markers_answer(asked.data_ptr(), n_asked, answers.data_ptr())
    back, back_counts = _all_to_all_bytes(answers, asked_counts, 4, group, device, staged)
    assert back_counts == counts
    fwd = torch.empty(max(1, n_reads), dtype=torch.int32, device=device)
    rev = torch.empty(max(1, n_reads), dtype=torch.int32, device=device)
    if str(device) != "cpu":
        torch.cuda.current_stream().synchronize()
    engine.markers_apply(back.data_ptr(), total, fwd.data_ptr(), rev.data_ptr())
    return fwd[:n_reads].cpu().numpy(), rev[:n_reads].cpu().numpy()


HIST_DENSE = 65536


def reduce_histogram(cov, cnt, group=None, device="cuda", size=1, always=False):
    """merge the ranks' (coverage, count) pairs like rank 0 does on RAY_MPI_TAG_COVERAGE_DATA
    (MessageProcessor.cpp:1048-1061): a dense all-reduce for coverage < 65536, the rare tail by object
    gather.  Every rank gets the merged, ascending arrays."""
    import numpy as np
    if size == 1 and not always:
        return cov, cnt
    dense = torch.zeros(HIST_DENSE, dtype=torch.int64)
    low = cov < HIST_DENSE
    dense[torch.from_numpy(cov[low].astype(np.int64))] = torch.from_numpy(cnt[low].astype(np.int64))
    if not _staged(group):
        dense = dense.to(device)
    dist.all_reduce(dense, op=dist.ReduceOp.SUM, group=group)
    dense = dense.cpu().numpy()
    tail = list(zip(cov[~low].tolist(), cnt[~low].tolist()))
    tails = [None] * size
    dist.all_gather_object(tails, tail, group=group)
    merged = {}
    for t in tails:
        for c, n in t:
            merged[c] = merged.get(c, 0) + n
    idx = np.nonzero(dense)[0]
    out_cov = np.concatenate([idx.astype(np.uint32), np.asarray(sorted(merged), dtype=np.uint32)])
    out_cnt = np.concatenate([dense[idx].astype(np.uint64),
                              np.asarray([merged[c] for c in sorted(merged)], dtype=np.uint64)])
    return out_cov, out_cnt
