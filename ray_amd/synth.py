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
    # Ray Meta: 200 genomes of 2-6 Mbp, abundance lognormal(0, 1) normalised to a mean coverage of 30x; `genome`
    # is the nominal total (200 x 4 Mbp), the generated one is the sum of the drawn lengths
    "C4_meta_200sp_lognormal_30x_150bp_k31": dict(genome=800_000_000, read_len=150, coverage=30, k=31, error=0.0, seed=4,
                                                   species=200, genome_min=2_000_000, genome_max=6_000_000, sigma=1.0),
}


def n_reads_for(genome, read_len, coverage):
    return int(genome * coverage // read_len)


def metagenome_plan(cfg, scale=1.0, species_factor=1):
    """C4: -> (lengths int64 [S], coverage float64 [S], n_reads).  Species i has a genome of lengths[i] bases (uniform in
    [genome_min, genome_max] x scale) and is sequenced coverage[i] deep: lognormal(0, sigma) abundances scaled so that their
    mean is cfg["coverage"].  A read comes from species i with probability ~ coverage[i] * lengths[i]."""
    rng = np.random.default_rng(cfg["seed"])
    S = cfg["species"] * species_factor
    lo = max(cfg["read_len"] + 1, int(cfg["genome_min"] * scale))
    hi = max(lo + 1, int(cfg["genome_max"] * scale))
    lengths = rng.integers(lo, hi + 1, size=S, dtype=np.int64)
    a = rng.lognormal(mean=0.0, sigma=cfg["sigma"], size=S)
    coverage = cfg["coverage"] * a / a.mean()
    n_reads = int((coverage * lengths).sum() // cfg["read_len"])
    return lengths, coverage, n_reads


def metagenome_numpy(cfg, scale):
    """C4 for make_reads_numpy(genomes=...): -> ([(codes, weight)], n_reads)"""
    lengths, coverage, n_reads = metagenome_plan(cfg, scale)
    rng = np.random.default_rng(cfg["seed"] + 1000)
    return [(rng.integers(0, 4, size=int(n), dtype=np.uint8), float(c * n)) for n, c in zip(lengths, coverage)], n_reads


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


def make_reads_torch(genome, read_len, n_reads, seed, error=0.0, device="cuda", chunk=2_000_000, read_seed=None, species=None):
    """same distribution, generated directly in HBM with torch (plumbing only): fixed-length reads.
    species: (lengths, coverage) of metagenome_plan -- the genomes lie back to back in one array of sum(lengths) bases and a
    read starts inside the genome of a species drawn with probability ~ coverage * length.
    -> (packed u8 tensor padded by 32 bytes, offsets u64-as-int64 tensor, lengths int32 tensor)"""
    import torch
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    if species is not None:
        sp_len = torch.as_tensor(species[0], dtype=torch.int64, device=device)
        sp_off = torch.cumsum(sp_len, 0) - sp_len
        sp_w = torch.as_tensor(species[1] * species[0], dtype=torch.float64, device=device)
        genome = int(species[0].sum())
    g = torch.randint(0, 4, (genome,), dtype=torch.uint8, device=device, generator=gen)
    if read_seed is not None:  # same genome on every rank, different reads
        gen.manual_seed(read_seed)
    bpr = (read_len + 3) // 4
    packed = torch.zeros(n_reads * bpr + 32, dtype=torch.uint8, device=device)
    ar = torch.arange(read_len, device=device, dtype=torch.int64)
    Lp = bpr * 4
    for c0 in range(0, n_reads, chunk):
        m = min(chunk, n_reads - c0)
        if species is None:
            starts = torch.randint(0, genome - read_len + 1, (m,), device=device, generator=gen, dtype=torch.int64)
        else:
            which = torch.multinomial(sp_w.float(), m, replacement=True, generator=gen)
            room = (sp_len[which] - read_len + 1).double()
            starts = sp_off[which] + (torch.rand(m, device=device, generator=gen, dtype=torch.float64) * room).long()
            starts = torch.minimum(starts, sp_off[which] + sp_len[which] - read_len)
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


_synth_lib = None


def make_reads_device(genome, read_len, n_reads, seed, error=0.0, first=0, device="cuda"):
    """The counter-based generator (tools/rg_synth.hip; the tests hold a host twin of it): reads first .. first+n_reads-1,
    made in HBM -- every value is a pure function of (seed, index), so the same reads can be made anywhere.
    -> (packed u8 tensor padded by 32 bytes, offsets int64 tensor, lengths int32 tensor), as make_reads_torch"""
    import ctypes as C
    import os
    import torch
    global _synth_lib
    if _synth_lib is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "librg_synth.so")
        if not os.path.exists(path):
            raise ImportError("%s is missing: build it with `make -C tools` (or __graft_entry__.build())" % path)
        _synth_lib = C.CDLL(path)
        _synth_lib.rgs_generate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64,
                                            C.c_double, C.c_void_p]
        _synth_lib.rgs_generate.restype = C.c_int
    bpr = (read_len + 3) // 4
    packed = torch.zeros(n_reads * bpr + 32, dtype=torch.uint8, device=device)
    offsets = torch.empty(max(1, n_reads), dtype=torch.int64, device=device)
    lengths = torch.empty(max(1, n_reads), dtype=torch.int32, device=device)
    torch.cuda.synchronize()
    with torch.cuda.device(packed.device):
        rc = _synth_lib.rgs_generate(packed.data_ptr(), offsets.data_ptr(), lengths.data_ptr(), n_reads, first, genome, read_len, seed, error, None)
    if rc != 0:
        raise RuntimeError("rgs_generate failed (%d)" % rc)
    return packed, offsets[:n_reads], lengths[:n_reads]
