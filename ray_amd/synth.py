"""Synthetic read sets for the parity tests and the bench.

Genome: i.i.d. uniform ACGT; reads: uniform start, random strand, i.i.d. substitution errors, no N.
Pairing is not modelled: it does not enter the graph-construction path (reads are k-mer sources only).
Output is Ray's packed layout (SequencesLoader/Read.cpp:135-141): base i -> bits 2(i%4).. of byte i/4,
every read starting on a byte boundary.
"""
import numpy as np

# BASELINE.json configs (read length of configs 3 and 5 assumed 150)
CONFIGS = {
    "C1_ecoli_50x_100bp_k31": dict(genome=4_600_000, read_len=100, coverage=50, k=31, error=0.0, seed=1),
    "C2_100Mbp_60x_150bp_k31": dict(genome=100_000_000, read_len=150, coverage=60, k=31, error=0.0, seed=2),
    "C3_500Mbp_40x_150bp_k63": dict(genome=500_000_000, read_len=150, coverage=40, k=63, error=0.0, seed=3),
    "C5_3.1Gbp_30x_150bp_k31_1pct": dict(genome=3_100_000_000, read_len=150, coverage=30, k=31, error=0.01, seed=5),
}


def n_reads_for(genome, read_len, coverage):
    return int(genome * coverage // read_len)


def make_reads_numpy(genome, read_len, n_reads, seed, error=0.0, ragged=False, genomes=None):
    """-> (packed u8, byte offsets u64, lengths u32).  `genomes`: optional list of (codes, weight) for a
    metagenome; otherwise one genome of `genome` bases.  ragged=True draws lengths in [1, read_len]."""
    rng = np.random.default_rng(seed)
    if genomes is None:
        g = rng.integers(0, 4, size=genome, dtype=np.uint8)
        starts = rng.integers(0, genome - read_len + 1, size=n_reads, dtype=np.int64)
        idx = starts[:, None] + np.arange(read_len, dtype=np.int64)[None, :]
        codes = g[idx]
    else:
        weights = np.asarray([w for _, w in genomes], dtype=np.float64)
        which = rng.choice(len(genomes), size=n_reads, p=weights / weights.sum())
        codes = np.zeros((n_reads, read_len), dtype=np.uint8)
        for gi, (g, _) in enumerate(genomes):
            sel = np.nonzero(which == gi)[0]
            starts = rng.integers(0, len(g) - read_len + 1, size=len(sel), dtype=np.int64)
            codes[sel] = g[starts[:, None] + np.arange(read_len, dtype=np.int64)[None, :]]
    rev = rng.random(n_reads) < 0.5
    codes[rev] = 3 - codes[rev][:, ::-1]
    if error > 0:
        err = rng.random(codes.shape) < error
        codes = np.where(err, (codes + rng.integers(1, 4, size=codes.shape, dtype=np.uint8)) & 3, codes).astype(np.uint8)
    lengths = np.full(n_reads, read_len, dtype=np.uint32)
    if ragged:
        lengths = rng.integers(1, read_len + 1, size=n_reads, dtype=np.uint32)
        codes = np.where(np.arange(read_len)[None, :] < lengths[:, None], codes, 0).astype(np.uint8)
    return pack_codes(codes, lengths)


def pack_codes(codes, lengths):
    """codes (n, L) uint8 in 0..3 (zero beyond each length) -> Ray packed layout, reads back to back"""
    n, L = codes.shape
    Lp = (L + 3) // 4 * 4
    if Lp != L:
        codes = np.concatenate([codes, np.zeros((n, Lp - L), dtype=np.uint8)], axis=1)
    q = codes.reshape(n, Lp // 4, 4)
    by = (q[:, :, 0] | (q[:, :, 1] << 2) | (q[:, :, 2] << 4) | (q[:, :, 3] << 6)).astype(np.uint8)
    nbytes = (lengths.astype(np.int64) + 3) // 4
    offsets = np.zeros(n, dtype=np.uint64)
    if n:
        offsets[1:] = np.cumsum(nbytes)[:-1].astype(np.uint64)
    if (nbytes == Lp // 4).all():
        packed = by.reshape(-1)
    else:
        keep = np.arange(Lp // 4)[None, :] < nbytes[:, None]
        packed = by[keep]
    return np.ascontiguousarray(packed), offsets, lengths.astype(np.uint32)


def make_reads_torch(genome, read_len, n_reads, seed, error=0.0, device="cuda", chunk=2_000_000, read_seed=None):
    """same distribution, generated directly in HBM with torch (plumbing only): fixed-length reads.
    -> (packed u8 tensor padded by 32 bytes, offsets u64-as-int64 tensor, lengths int32 tensor)"""
    import torch
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    g = torch.randint(0, 4, (genome,), dtype=torch.uint8, device=device, generator=gen)
    if read_seed is not None:  # same genome on every rank, different reads
        gen.manual_seed(read_seed)
    bpr = (read_len + 3) // 4
    packed = torch.zeros(n_reads * bpr + 32, dtype=torch.uint8, device=device)
    ar = torch.arange(read_len, device=device, dtype=torch.int64)
    Lp = bpr * 4
    for c0 in range(0, n_reads, chunk):
        m = min(chunk, n_reads - c0)
        starts = torch.randint(0, genome - read_len + 1, (m,), device=device, generator=gen, dtype=torch.int64)
        codes = g[starts[:, None] + ar[None, :]]
        rev = torch.rand(m, device=device, generator=gen) < 0.5
        codes = torch.where(rev[:, None], 3 - codes.flip(1), codes)
        if error > 0:
            err = torch.rand(codes.shape, device=device, generator=gen) < error
            sub = torch.randint(1, 4, codes.shape, device=device, generator=gen, dtype=torch.uint8)
            codes = torch.where(err, (codes + sub) & 3, codes)
        if Lp != read_len:
            codes = torch.cat([codes, torch.zeros(m, Lp - read_len, dtype=torch.uint8, device=device)], dim=1)
        q = codes.view(m, bpr, 4)
        by = q[:, :, 0] | (q[:, :, 1] << 2) | (q[:, :, 2] << 4) | (q[:, :, 3] << 6)
        packed[c0 * bpr:(c0 + m) * bpr] = by.reshape(-1)
        del codes, q, by, starts, rev
    offsets = torch.arange(n_reads, device=device, dtype=torch.int64) * bpr
    lengths = torch.full((n_reads,), read_len, dtype=torch.int32, device=device)
    return packed, offsets, lengths
