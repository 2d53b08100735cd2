"""Experiment (development): do two independent graph builds on two streams overlap on one GPU?  Two builders, each on
half of C2 (scale 0.5), run one after the other and then at the same time from two host threads (ctypes releases the
GIL).  If the concurrent pair is much faster than the sequential pair, splitting a step into sub-pipelines that overlap
latency-bound kernels with bandwidth-bound ones would pay.  usage: python tools/exp_overlap.py [scale] [error]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ray_amd
from ray_amd import synth

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
error = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
cfg = dict(synth.CONFIGS["C2_100Mbp_60x_150bp_k31"])
k = cfg["k"]
genome = int(cfg["genome"] * scale)
n_reads = synth.n_reads_for(genome, cfg["read_len"], cfg["coverage"])
inputs = []
for i in range(2):
    inputs.append(synth.make_reads_device(genome, cfg["read_len"], n_reads, cfg["seed"] + i, error, first=0, device="cuda:0"))
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(2)]


def build(i):
    packed, offsets, lengths = inputs[i]
    b = ray_amd.GraphBuilder(k, device=0, stream=streams[i].cuda_stream)
    b.add_reads_device(packed.data_ptr(), n_reads * ((cfg["read_len"] + 3) // 4), offsets.data_ptr(), lengths.data_ptr(), n_reads, cfg["read_len"])
    b.count_kmers()
    n = b.size_()
    b.close()
    return n


for i in range(2):
    build(i)  # warm-up
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    build(0)
    build(1)
    torch.cuda.synchronize()
    seq = time.perf_counter() - t0
    t0 = time.perf_counter()
    th = [threading.Thread(target=build, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    con = time.perf_counter() - t0
    print("scale %.2f error %.3f: two builds one after the other %.1f ms, at the same time %.1f ms" % (scale, error, seq * 1e3, con * 1e3), flush=True)
