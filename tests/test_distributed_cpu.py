"""N>1 path on CPU: two processes over gloo run ray_amd.exchange (partitioning, count/bucket
all-to-all(v), merge, histogram reduction) with the CPU stand-in engine of tests/model.py, and the
result must equal the 2-rank oracle."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, k, bloom, out_dir, route):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist
    from model import ModelEngine
    from oracle import ray_oracle as ro
    from ray_amd.exchange import (exchange_owner_buckets, exchange_records, partition_reads, purge_edges_distributed,
                                  read_markers_distributed, reduce_histogram)
    from ray_amd.synth import make_reads_numpy

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    packed, offsets, lengths = make_reads_numpy(genome=1200, read_len=60, n_reads=151, seed=77, error=0.02)
    first, last = partition_reads(len(lengths), rank, world)
    e = ModelEngine(ro, k, rank, world, bloom, first_read_id=first)
    e.add_reads(packed, offsets[first:last], lengths[first:last])
    if route == "records":   # records to the rank that aggregates their bucket, then entries to their owners
        n_local = len(e.local)
        sent, received = exchange_records(e, device="cpu", parts=3 if k == 21 and bloom is None else 1)
        assert sent == n_local
    else:                    # every rank aggregates its own reads, then entries to their owners
        e.count_local()
    sent, received = exchange_owner_buckets(e, device="cpu")
    assert sent == len(e.local)
    e.resolve_admission(last - first)
    cov, cnt = reduce_histogram(*e.histogram(), device="cpu", size=world)
    np.save(os.path.join(out_dir, "rec%d.npy" % rank), e.records())
    fwd, rev = read_markers_distributed(e, last - first, device="cpu")   # SequencesIndexer: k-mers out, coverages back
    np.save(os.path.join(out_dir, "markers%d.npy" % rank), np.stack([fwd, rev]))
    asked = purge_edges_distributed(e, device="cpu")   # RAY_SLAVE_MODE_PURGE_NULL_EDGES: key all-to-all, answers back
    assert asked == sum(bin(int(em)).count("1") for em in np.load(os.path.join(out_dir, "rec%d.npy" % rank))["edges"])
    np.save(os.path.join(out_dir, "purged%d.npy" % rank), e.records())
    if rank == 0:
        np.save(os.path.join(out_dir, "hist.npy"), np.stack([cov.astype(np.uint64), cnt]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("k,bloom,route", [(21, 4000, "entries"), (21, None, "records"), (21, 4000, "records")])
def test_two_rank_exchange_over_gloo(oracle, tmp_path, k, bloom, route):
    sys.path.insert(0, HERE)
    from helpers import assert_records_equal, oracle_run
    from ray_amd.synth import make_reads_numpy
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, k, bloom, str(tmp_path), route), nprocs=2, join=True)
    packed, offsets, lengths = make_reads_numpy(genome=1200, read_len=60, n_reads=151, seed=77, error=0.02)
    want = oracle_run(oracle, packed, offsets, lengths, k, 2, bloom)
    for r in range(2):
        got = np.load(str(tmp_path / ("rec%d.npy" % r)))
        assert_records_equal(got, want["graph"].export(r), "rank %d" % r)
    wf, wr = want["graph"].read_markers()
    got = np.concatenate([np.load(str(tmp_path / ("markers%d.npy" % r))) for r in range(2)], axis=1)
    assert np.array_equal(got[0], wf) and np.array_equal(got[1], wr)
    want["graph"].purge_edges()
    changed = 0
    for r in range(2):
        got = np.load(str(tmp_path / ("purged%d.npy" % r)))
        assert_records_equal(got, want["graph"].export(r), "rank %d after purge" % r)
        changed += int((got["edges"] != np.load(str(tmp_path / ("rec%d.npy" % r)))["edges"]).sum())
    assert changed > 0, "the test data must contain edges towards vertices that were not admitted"
    hist = np.load(str(tmp_path / "hist.npy"))
    assert np.array_equal(hist[0], want["hist"][0].astype(np.uint64)) and np.array_equal(hist[1], want["hist"][1])


def test_partition_reads_matches_sequences_loader():
    """SequencesLoader.cpp:287-297"""
    from ray_amd.exchange import partition_reads
    assert [partition_reads(10, r, 3) for r in range(3)] == [(0, 3), (3, 6), (6, 10)]
    assert [partition_reads(2, r, 4) for r in range(4)] == [(0, 0), (0, 0), (0, 0), (0, 2)]
    assert partition_reads(7, 0, 1) == (0, 7)
