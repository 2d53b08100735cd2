"""CPU proof that the order-free formulation the kernels use (aggregate -> stamps -> Bloom resolution)
gives exactly what the reference's order-dependent handlers give, for 1, 2 and 3 ranks, with the
filter at its default size, squeezed until false positives are common, and off."""
import numpy as np
import pytest

from helpers import assert_records_equal, oracle_run
from model import ModelEngine
from ray_amd.exchange import partition_reads
from ray_amd.synth import make_reads_numpy


def run_model(ro, packed, offsets, lengths, k, nranks, bloom_bits, W=None):
    n = len(lengths)
    engines = []
    for r in range(nranks):
        first, last = partition_reads(n, r, nranks)
        e = ModelEngine(ro, k, r, nranks, bloom_bits, W, first_read_id=first)
        e.add_reads(packed, offsets[first:last], lengths[first:last])
        e.count_local()
        engines.append(e)
    if nranks > 1:
        import ctypes as C
        from model import ENTRY
        bufs = []
        for e in engines:
            counts = e.owner_counts()
            buf = np.zeros(max(1, sum(counts)), dtype=ENTRY)
            e.export_owner_buckets(buf.ctypes.data, len(buf))
            bufs.append((counts, buf))
        for dst, e in enumerate(engines):
            for counts, buf in bufs:
                start = sum(counts[:dst])
                part = np.ascontiguousarray(buf[start:start + counts[dst]])
                e.merge_entries(part.ctypes.data, len(part))
    for r, e in enumerate(engines):
        first, last = partition_reads(n, r, nranks)
        e.resolve_admission(last - first)
    return engines


@pytest.mark.parametrize("k,nranks,bloom", [(21, 1, None), (21, 1, 0), (21, 1, 20000), (21, 1, 3000), (21, 2, 8000),
                                           (31, 3, 5000), (33, 1, 6000), (63, 2, None), (5, 1, 64), (21, 2, 0)])
def test_stamp_formulation_equals_sequential_reference(oracle, k, nranks, bloom):
    read_len = 100 if k > 31 else 60
    packed, offsets, lengths = make_reads_numpy(genome=1500, read_len=read_len, n_reads=160, seed=k * 10 + nranks, error=0.02)
    want = oracle_run(oracle, packed, offsets, lengths, k, nranks, bloom)
    engines = run_model(oracle, packed, offsets, lengths, k, nranks, bloom)
    got = np.concatenate([e.records() for e in engines])
    order = np.lexsort(tuple(got["key"][:, j] for j in reversed(range(got["key"].shape[1]))))
    assert_records_equal(got[order], want["pass2"], "model vs oracle")
    for r, e in enumerate(engines):
        bits, set_bits, insertions = want["bloom"][r]
        assert e.m_bits == bits
        if bits:
            assert (e.set_bits, e.insertions) == (set_bits, insertions)
    # false positives must actually occur in the squeezed cases, otherwise the test proves nothing
    if bloom is not None and 0 < bloom <= 8000:
        distinct = len(oracle_run(oracle, packed, offsets, lengths, k, nranks, 0)["pass2"])
        assert sum(b[2] for b in want["bloom"]) < distinct
