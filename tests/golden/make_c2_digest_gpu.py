"""tests/golden/full_size_C2.json when the oracle's own run (make_full_size_digests.py C2: hours on one core) is not at hand:
the same fields computed by the HIP path on the GPU box, so that later kernel changes cannot move C2's full-size results
silently.  A regression digest, NOT an oracle result -- the file says so in "_made_by"; the oracle pins the same code at full
size through C1 (full_size_C1.json) and at every smaller size through the parity tests.

  gpurun -- python tests/golden/make_c2_digest_gpu.py     (writes gpurun_out/full_size_C2_gpu.json)
"""
import hashlib
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import ray_amd  # noqa: E402
from ray_amd.synth import CONFIGS, make_reads_device, n_reads_for  # noqa: E402

name = "C2_100Mbp_60x_150bp_k31"
cfg = CONFIGS[name]
n = n_reads_for(cfg["genome"], cfg["read_len"], cfg["coverage"])
packed, offsets, lengths = make_reads_device(cfg["genome"], cfg["read_len"], n, cfg["seed"], cfg["error"])
b = ray_amd.GraphBuilder(cfg["k"], stream=torch.cuda.current_stream().cuda_stream)
b.add_reads_device(packed.data_ptr(), n * ((cfg["read_len"] + 3) // 4), offsets.data_ptr(), lengths.data_ptr(), n, cfg["read_len"])
b.count_kmers()
st = b.stats()
out = {"_made_by": "tests/golden/make_c2_digest_gpu.py: the HIP path itself (regression digest, not an oracle result)", "workload": name,
       "generator": "counter v1 (tools/rg_synth.hip, oracle/ro_synth.c)", "config": cfg, "reads": n, "occurrences": st["occurrences"], "size": b.size_(),
       "bloom": [st["bloom_bits"], st["bloom_set_bits"], st["bloom_insertions"]]}
cov, cnt = b.histogram()
out["histogram"] = [cov.tolist(), cnt.tolist()]
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "CoverageDistribution.txt")
    out["minimum_peak_repeat"] = list(ray_amd.coverage_distribution(cov, cnt, path))
    out["sha256_CoverageDistribution_txt"] = hashlib.sha256(open(path, "rb").read()).hexdigest()
rec = b.export(sort=False)
order = torch.from_numpy(np.ascontiguousarray(rec["key"][:, 0]).view(np.int64)).cuda().argsort().cpu().numpy()
out["sha256_sorted_records_after_pass1"] = hashlib.sha256(rec[order].tobytes()).hexdigest()
b.add_edges()
out["sha256_sorted_records_after_pass2"] = hashlib.sha256(b.export(sort=False)[order].tobytes()).hexdigest()
b.purge_edges()
out["sha256_sorted_records_after_purge"] = hashlib.sha256(b.export(sort=False)[order].tobytes()).hexdigest()
out["degree_distribution"] = b.degree_distribution().reshape(-1).tolist()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "full_size_C2_gpu.json"), "w"), indent=1)
print("ok", out["size"], out["sha256_sorted_records_after_purge"])
