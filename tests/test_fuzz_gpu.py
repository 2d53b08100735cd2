"""Seeded random cases through the C ABI against the oracle: k, read length (short reads for the lane-per-read kernels,
reads beyond 1024 bases for the wavefront-per-read one), coverage, error rate, filter size (default, off, powers of two and
odd sizes from far too small to ample), reads added in one to three batches, two-word keys for a k that one word holds, and
a random mix of the switches that force internal routes.  Every case is
deterministic (the seed is the case number)."""
import os

import numpy as np
import pytest

from helpers import gpu_vs_oracle
from ray_amd.synth import make_reads_numpy

pytestmark = pytest.mark.gpu

FLAG_CHOICES = (0x4, 0x8, 0x80, 0x10000, 0x20000, 0x40000, 0x100000, 0x400000, 0x800000, 0x1000000, 0x2000000, 0x4000000, 0x8000000, 0x10000000, 0x20000000, 0x40000000, 0x80000000)


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    k = int(rng.choice([15, 17, 21, 27, 31, 33, 37, 39, 45, 47, 55, 63]))
    long_reads = rng.random() < 0.2
    read_len = int(rng.integers(1100, 1600)) if long_reads else int(rng.integers(max(k + 5, 40), 320))
    genome = int(rng.integers(2000, 30000))
    coverage = float(rng.choice([3, 8, 20, 45]))
    n_reads = max(50, int(genome * coverage / read_len))
    error = float(rng.choice([0.0, 0.002, 0.01, 0.03]))
    ragged = bool(rng.random() < 0.3)
    bloom = rng.choice(["default", "off", "pow2", "odd"])
    if bloom == "default":
        bloom_bits = None
    elif bloom == "off":
        bloom_bits = 0
    elif bloom == "pow2":
        bloom_bits = 1 << int(rng.integers(10, 30))
    else:
        bloom_bits = int(rng.integers(5000, 3_000_000))
    sweeps = []
    for _ in range(3):
        route = int(rng.choice([1, 1, 1, 2]))                      # partitioned sweep most of the time
        s = route | int(rng.choice([0, 16, 32]))
        if route == 1 and rng.random() < 0.7:
            s |= int(rng.integers(8, 21)) << 8                     # forced bucket count; otherwise the library's own choice
        for f in FLAG_CHOICES:
            if rng.random() < 0.25:
                s |= f
        if (s & 0x400000) and (s & 0x800000):
            s &= ~0x800000
        sweeps.append(s)
    # (drawn last, so that the cases keep what they had before these were added)
    batches = int(rng.choice([1, 1, 2, 3]))
    W = 2 if (k <= 31 and rng.random() < 0.15) else None
    return dict(k=k, read_len=read_len, genome=genome, n_reads=n_reads, error=error, ragged=ragged, bloom_bits=bloom_bits, sweeps=tuple(sweeps),
                batches=batches, W=W)


# RG_FUZZ_SEEDS=N widens the run for a one-off check (24 cases are the suite's share)
# case 281 (k = 15: one k-mer per record, cut by the wavefront-per-read kernel) once overflowed the record buffer's bound
@pytest.mark.parametrize("seed", sorted(set(range(int(os.environ.get("RG_FUZZ_SEEDS", "24")))) | {281}))
def test_random_case(oracle, seed):
    c = _case(seed)
    packed, offsets, lengths = make_reads_numpy(c["genome"], c["read_len"], c["n_reads"], seed=500 + seed, error=c["error"], ragged=c["ragged"])
    gpu_vs_oracle(oracle, packed, offsets, lengths, c["k"], bloom_bits=c["bloom_bits"], sweeps=c["sweeps"], batches=c["batches"], W=c["W"])


ITEM_FLAG_CHOICES = (0x4, 0x8, 0x80, 16, 32, 0x10000, 0x400000, 0x800000, 0x4000000, 0x8000000, 0x10000000, 0x20000000, 0x40000000, 0x80000000)


def _item_case(seed):
    """the routes of large one-rank inputs on small ones: the partition sized from a record sample, a filter of 2^28 bits or
    more (the aggregation makes the admission's items and, for one-word keys, a 16-byte log), a table large enough for the
    one-pass admission half of the time, and a random mix of the switches that turn parts of that off or force another route"""
    rng = np.random.default_rng(9000 + seed)
    k = int(rng.choice([17, 21, 27, 31, 31, 33, 47, 63]))
    big = k > 32
    read_len = int(rng.integers(120, 260))
    genome = int(rng.integers(15000, 40000)) * (4 if big else 1)
    n_reads = int(genome * float(rng.choice([20, 40])) / read_len)
    sweeps = []
    for _ in range(2):
        s = 1 | 0x2000000 | 0x1000000
        for f in ITEM_FLAG_CHOICES:
            if rng.random() < 0.2:
                s |= f
        if (s & 16) and (s & 32):
            s &= ~32
        if (s & 0x400000) and (s & 0x800000):
            s &= ~0x800000
        sweeps.append(s)
    return dict(k=k, read_len=read_len, genome=genome, n_reads=n_reads, error=float(rng.choice([0.0, 0.01, 0.03])), ragged=bool(rng.random() < 0.3),
                bloom_bits=1 << int(rng.choice([28, 28, 29, 30, 31, 33])), sweeps=tuple(sweeps), batches=int(rng.choice([1, 2])),
                table_slots=int(rng.choice([0, 1 << 21])), W=2 if (not big and rng.random() < 0.1) else None)


# RG_FUZZ_ITEM_SEEDS=N widens the run
@pytest.mark.parametrize("seed", range(int(os.environ.get("RG_FUZZ_ITEM_SEEDS", "10"))))
def test_random_case_with_items_from_the_aggregation(oracle, seed):
    c = _item_case(seed)
    packed, offsets, lengths = make_reads_numpy(c["genome"], c["read_len"], c["n_reads"], seed=900 + seed, error=c["error"], ragged=c["ragged"])
    gpu_vs_oracle(oracle, packed, offsets, lengths, c["k"], bloom_bits=c["bloom_bits"], sweeps=c["sweeps"], batches=c["batches"], W=c["W"],
                  table_slots=c["table_slots"])


@pytest.mark.parametrize("k", [21, 31, 47])
def test_record_header_extremes(oracle, k):
    """what a super-k-mer record's header must hold at its largest: read ids just below 2^33 (the arrival stamps only order
    the occurrences, so the results are those of ids from 0) and positions up to the longest read the reference stores
    (65534 bases: the wavefront-per-read kernel), in the 20-byte records of k <= 31 and in the full-size ones; a Bloom filter
    small enough that false positives, which depend on the order, do occur"""
    rng = np.random.default_rng(k)
    genome = rng.integers(0, 4, size=70000, dtype=np.uint8)
    reads = [genome[:65534], genome[3000:68534], genome[::-1][:40000].copy()]
    for _ in range(300):
        a = int(rng.integers(0, 69000))
        reads.append(genome[a:a + int(rng.integers(k, 900))])
    lengths = np.array([len(r) for r in reads], dtype=np.uint32)
    nbytes = (lengths.astype(np.int64) + 3) // 4
    offsets = np.concatenate([[0], np.cumsum(nbytes)[:-1]]).astype(np.uint64)
    packed = np.zeros(int(nbytes.sum()) + 16, dtype=np.uint8)
    for r, o in zip(reads, offsets):
        pad = np.zeros((len(r) + 3) // 4 * 4, dtype=np.uint8)
        pad[:len(r)] = r
        q = pad.reshape(-1, 4)
        packed[int(o):int(o) + len(q)] = q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6)
    first = (1 << 33) - 2 - len(reads)
    gpu_vs_oracle(oracle, packed, offsets, lengths, k, bloom_bits=300_000, first_read_id=first,
                  sweeps=(1 | 16 | (12 << 8), 1 | 32 | (14 << 8) | 0x8000000, 1 | 16 | (9 << 8) | 0x10000, 2 | 32))


def _rank_case(seed):
    rng = np.random.default_rng(7000 + seed)
    k = int(rng.choice([15, 21, 31, 33, 47, 63]))
    nranks = int(rng.choice([2, 3, 5, 8]))
    read_len = int(rng.integers(max(k + 5, 50), 260))
    genome = int(rng.integers(3000, 20000))
    n_reads = max(4 * nranks, int(genome * float(rng.choice([4, 12, 30])) / read_len))
    bloom = [None, 0, int(rng.integers(20000, 400000)), 1 << int(rng.integers(14, 24))][int(rng.integers(0, 4))]
    route = str(rng.choice(["records", "entries"]))
    parts = int(rng.choice([1, 1, 2, 3])) if route == "records" else 1
    one = [False, True, "adopt"][int(rng.integers(0, 3))]
    sweep = 1 | int(rng.choice([16, 32])) | (int(rng.integers(8, 18)) << 8)
    for f in (0x10000, 0x100000, 0x400000, 0x800000, 0x1000000, 0x4000000, 0x8000000):
        if rng.random() < 0.2:
            sweep |= f
    if (sweep & 0x400000) and (sweep & 0x800000):
        sweep &= ~0x800000
    return dict(k=k, nranks=nranks, read_len=read_len, genome=genome, n_reads=n_reads, bloom=bloom, route=route, parts=parts, one=one,
                sweep=sweep, error=float(rng.choice([0.0, 0.01])))


@pytest.mark.parametrize("seed", range(int(os.environ.get("RG_FUZZ_RANK_SEEDS", "8"))))
def test_random_ranks(oracle, seed):
    """R builders on one device swapping records / entries, purge queries and answers through device buffers, against the
    R-rank oracle: random k, rank count, route, slices, delivery (several merges, one merge, adoption in place) and switches"""
    from helpers import assert_records_equal, oracle_run
    from test_multirank_gpu import check_ranks, purge_ranks, run_ranks
    c = _rank_case(seed)
    packed, offsets, lengths = make_reads_numpy(c["genome"], c["read_len"], c["n_reads"], seed=900 + seed, error=c["error"])
    want = oracle_run(oracle, packed, offsets, lengths, c["k"], c["nranks"], c["bloom"])
    builders = run_ranks(packed, offsets, lengths, c["k"], c["nranks"], c["bloom"], sweep=c["sweep"], route=c["route"], parts=c["parts"], one_merge=c["one"])
    check_ranks(builders, want, "%s" % c)
    want["graph"].purge_edges()
    purge_ranks(builders)
    for r, b in enumerate(builders):
        assert_records_equal(b.export(), want["graph"].export(r), "rank %d after purge: %s" % (r, c))
