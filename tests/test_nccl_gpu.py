"""The exchanges of ray_amd/exchange.py on a real RCCL communicator ("nccl" backend).  A GPU box has one GPU, so the
process group has one rank: every all-to-all(v) sends the rank's records / entries / purge queries to itself, but the
communicator is initialised, the collectives run on device tensors and the library consumes what they delivered.
Results must equal the single-rank oracle."""
import os
import socket

import numpy as np
import pytest

from helpers import assert_records_equal, oracle_run
from ray_amd.synth import make_reads_numpy

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_group():
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda:0"))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("k,parts", [(31, 3), (63, 1)])
def test_exchanges_run_on_rccl(oracle, nccl_group, k, parts):
    import torch
    import ray_amd
    from ray_amd.exchange import exchange_owner_buckets, exchange_records, purge_edges_distributed, read_markers_distributed, reduce_histogram
    assert nccl_group.get_backend() == "nccl"
    packed, offsets, lengths = make_reads_numpy(genome=30000, read_len=150, n_reads=9000, seed=70 + k, error=0.01)
    bloom = 2000000
    want = oracle_run(oracle, packed, offsets, lengths, k, 1, bloom, purge=True)
    b = ray_amd.GraphBuilder(k, bloom_bits=bloom, sweep=1)
    b.add_reads(packed, offsets, lengths)
    sent, got = exchange_records(b, device="cuda:0", parts=parts)          # all_to_all_single of 24 / 40-byte records
    assert sent == got > 0
    sent, got = exchange_owner_buckets(b, device="cuda:0", always=True)    # ... of 32-byte entries
    assert sent == got > 0
    b.resolve_admission(len(lengths))
    cov, cnt = b.histogram()
    cov, cnt = reduce_histogram(cov, cnt, device="cuda:0", size=1, always=True)   # all_reduce
    assert np.array_equal(cov, want["hist"][0]) and np.array_equal(cnt, want["hist"][1])
    assert_records_equal(b.export(), want["pass1"], "after pass 1")
    b.add_edges()
    assert_records_equal(b.export(), want["pass2"], "after pass 2")
    wf, wr = want["graph"].read_markers()                                   # markers before the purge, as SequencesIndexer runs
    gf, gr = read_markers_distributed(b, len(lengths), device="cuda:0")   # keys out, 4-byte coverages back
    assert np.array_equal(gf, wf) and np.array_equal(gr, wr)
    asked = purge_edges_distributed(b, device="cuda:0", always=True)      # keys out, one-byte answers back
    assert asked > 0
    assert_records_equal(b.export(), want["purged"], "after purge")
    torch.cuda.synchronize()
    b.close()
