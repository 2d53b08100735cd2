"""development aid: one configuration through the HIP path and the oracle, differences printed"""
import sys
import numpy as np
sys.path.insert(0, ".")
import ray_amd
from oracle import ray_oracle as ro
from ray_amd.synth import make_reads_numpy

# usage: python tools/debug_parity.py K BLOOM_BITS SWEEP_FLAGS READ_LENGTH N_READS   (on the GPU box)
k = int(sys.argv[1]); bloom = int(sys.argv[2]); sweep = int(sys.argv[3], 0); rl = int(sys.argv[4]); nreads = int(sys.argv[5])
packed, offsets, lengths = make_reads_numpy(genome=15000, read_len=rl, n_reads=nreads, seed=k + 2, error=0.015)
g = ro.Graph(k, nranks=1, bloom_bits=bloom)
g.set_reads(packed, offsets, lengths); g.count_kmers(); want = g.export()
b = ray_amd.GraphBuilder(k, bloom_bits=bloom, sweep=sweep)
b.add_reads(packed, offsets, lengths); b.count_kmers(); got = b.export()
print("k", k, "bloom", bloom, "sweep", hex(sweep), "vertices", len(got), len(want), "stats", {x: b.stats()[x] for x in ("occurrences", "candidates", "log_entries", "superkmer_records")})
if len(got) == len(want) and np.array_equal(got["key"], want["key"]):
    d = np.nonzero(got["coverage"] != want["coverage"])[0]
    print("coverage differs at", len(d), "of", len(got), [(int(got["coverage"][i]), int(want["coverage"][i])) for i in d[:12]])
else:
    print("key sets differ")
print("sum of coverages", int(got["coverage"].sum()), int(want["coverage"].sum()))
