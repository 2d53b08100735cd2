#!/usr/bin/env python3
"""bench.py -- graph-construction throughput on MI355X (BASELINE.json metric).

A step = one pass of the hot path over the workload's reads, already resident in HBM and 2-bit
packed: k-mer counting with Bloom admission, coverage histogram, edge bitmaps (the three slave modes
RAY_SLAVE_MODE_ADD_VERTICES / SEND_DISTRIBUTION / ADD_EDGES).  N=1 runs BASELINE.json configs[1]
(100 Mbp genome, 150 bp reads at 60x, k=31).  N>1 (one rank per GPU under torch.distributed.run) keeps
the per-GPU read count fixed and grows the genome with N (weak scaling); ranks exchange per-owner
buckets with one all-to-all over RCCL/xGMI per step.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PMC_SUMMARY = "profiles/c2_pmc_current.json"   # written by tools/summarize_profile.py together with the hash of the kernel sources it profiled
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="C2_100Mbp_60x_150bp_k31")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink genome and read count (development only)")
    ap.add_argument("--error", type=float, default=None, help="substitution error rate instead of the workload's own (sensitivity runs)")
    ap.add_argument("--emulate-world", type=int, default=1,
                    help="development: run what one of N ranks does on ONE GPU (records exported by destination, as many "
                         "full-coverage records counted as N ranks would send here, entries exported by owner and merged; "
                         "no network).  Not a bench line.")
    ap.add_argument("--cpu-sample-reads", type=int, default=150000, help="reads per CPU baseline instance")
    ap.add_argument("--cpu-cores", type=int, default=0, help="CPU baseline instances; 0 = min(16, host cores): the GPU box's share")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--route", choices=("auto", "records", "entries"), default="auto",
                    help="N > 1: what crosses the links first.  records: super-k-mer records to the rank that aggregates their "
                         "bucket, then aggregated entries to their owners; entries: every rank aggregates its own reads, entries "
                         "to their owners.  auto: entries when one rank's reads cover a k-mer 16 times or more (N = 2 here: "
                         "a third of the bytes on the one link there is), records otherwise (less work per rank, and fewer bytes "
                         "once a rank's own reads hold most k-mers only a few times)")
    ap.add_argument("--parts", type=int, default=4, help="records route: slices of the reads whose record transfers overlap the next slice's kernels")
    ap.add_argument("--sweep", type=lambda x: int(x, 0), default=0, help="rg_debug_set flags (development: forces a sweep / bucket bits)")
    ap.add_argument("--no-error-record", action="store_true", help="skip the extra steps on the workload with 1 %% errors (the error_1pct sub-record)")
    ap.add_argument("--no-after", action="store_true", help="skip the purge and the store hand-back that are measured after the timed steps")
    ap.add_argument("--breakdown", action="store_true", help="print per-call wall times of the last step to stderr")
    return ap.parse_args()


def _cpu_worker(job):
    """one single-rank instance of the oracle (the port of the reference's per-k-mer handlers) on its own reads"""
    h_packed, h_off, h_len, k = job
    from oracle import ray_oracle as ro
    g = ro.Graph(k, nranks=1)
    g.set_reads(h_packed, h_off, h_len)
    t0 = time.perf_counter()
    g.count_kmers()
    g.histogram()
    g.add_edges()
    dt = time.perf_counter() - t0
    occ, stored = g.occurrences(), g.size() // 2
    g.close()
    return occ, stored, dt


def cpu_baseline(packed, offsets, lengths, k, n_sample, cores):
    """The reference's CPU path beside the GPU number: the oracle is a port of Ray's sequential per-k-mer handlers;
    `cores` independent single-rank instances run at once, one per host core, each on its own n_sample reads of the
    same workload.  The aggregate is the sum of the instances' rates: what `mpiexec -n cores Ray` could reach at
    best, since real ranks also pay the message exchange."""
    import multiprocessing as mp
    import numpy as np
    bpr = int(offsets[1].item()) if len(offsets) > 1 else 0
    n = min(n_sample, len(lengths) // max(1, cores))
    jobs = []
    for c in range(cores):
        lo = c * n
        first = int(offsets[lo].item())
        h_packed = packed[first: first + n * bpr + 16].cpu().numpy()
        h_off = (offsets[lo:lo + n].cpu().numpy().astype(np.int64) - first).astype(np.uint64)
        h_len = lengths[lo:lo + n].cpu().numpy().astype(np.uint32)
        jobs.append((h_packed, h_off, h_len, k))
    t0 = time.perf_counter()
    if cores == 1:
        res = [_cpu_worker(jobs[0])]
    else:
        with mp.get_context("spawn").Pool(cores) as pool:   # spawn: the parent holds a GPU context, never fork it
            res = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    occ = sum(r[0] for r in res)
    stored = sum(r[1] for r in res)
    rate = sum(r[0] / r[2] for r in res)
    return {"value": rate, "unit": "k-mer occurrences/s", "cores": cores, "kind": "port",
            "sample": "%d instances x %d reads of the same workload (%d occurrences, %d vertices stored in all), %.1f s per "
                      "instance, %.1f s wall: count + histogram + edges; no message exchange is paid (every instance is a "
                      "whole one-rank graph whose tables fit in cache), so value = sum of the instances' rates is an upper bound "
                      "for mpiexec -n <cores> Ray on these cores"
                      % (cores, n, occ, stored, max(r[2] for r in res), wall),
            "distinct_kmers_per_s": sum(r[1] / r[2] for r in res)}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks here, before anything in this process has
        # touched the GPU (a process that has must never be replaced or forked), and hand on their exit code.  Rank 0's
        # JSON line goes to the same stdout.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    import torch
    import torch.distributed as dist
    import ray_amd
    from ray_amd import synth
    from ray_amd.exchange import exchange_owner_buckets, exchange_records, reduce_histogram

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started %d ranks" % (args.gpus, world))
    backend = os.environ.get("RAYGRAPH_BENCH_BACKEND", "nccl")  # "gloo": rehearse N ranks on one GPU (host-staged)
    if backend == "gloo":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    comm = None   # rg_comm: the exchanges inside the library, RCCL over xGMI (RAYGRAPH_BENCH_EXCHANGE=torch: driven from Python instead)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(dev))
            if os.environ.get("RAYGRAPH_BENCH_EXCHANGE", "library") != "torch":
                from ray_amd.exchange import create_comm
                # every rank must take the same route: if the library's communicator cannot be made on any of them (no
                # librccl.so.1 to open, a refused init), all fall back to the exchanges driven over torch.distributed
                try:
                    comm = create_comm(rank, world, local_rank)
                except Exception as exc:   # noqa: BLE001 -- reported below, the run goes on over the other route
                    comm = None
                    print("rank %d: rg_comm over RCCL unavailable (%s); using the torch.distributed route" % (rank, exc), file=sys.stderr)
                made = torch.tensor([1 if comm is not None else 0], device=dev)
                dist.all_reduce(made, op=dist.ReduceOp.MIN)
                if int(made.item()) == 0 and comm is not None:
                    comm.close()
                    comm = None
        else:
            dist.init_process_group(backend)

    cfg = dict(synth.CONFIGS[args.workload])
    k = cfg["k"]
    emu = args.emulate_world if world == 1 else 1
    nranks = world * emu
    # bytes per k-mer occurrence on the links: records ~ 24 B per 8.4 occurrences plus one entry per distinct k-mer of the
    # whole job; entries: 32 B per distinct k-mer of ONE rank's reads, i.e. 32 / (that rank's k-mer coverage).  Per-rank
    # work measured with --emulate-world (C2 shape, round 3): records 66 ms at N = 8; entries 104 ms at N = 2 (round 2: 185 at N = 8).
    local_kmer_cov = cfg["coverage"] * (cfg["read_len"] - k + 1) / cfg["read_len"] / nranks
    route = args.route if args.route != "auto" else ("entries" if local_kmer_cov >= 16 else "records")
    # weak scaling: the genome grows with N.  Emulated records route: a rank receives full-coverage buckets, like N = 1;
    # emulated entries route: the rank's own reads cover the N-fold genome N times thinner
    genome = max(2000, int(cfg["genome"] * args.scale)) * world * (emu if route == "entries" else 1)
    n_reads = synth.n_reads_for(int(cfg["genome"] * args.scale), cfg["read_len"], cfg["coverage"])
    species = None
    if "species" in cfg:   # C4: the metagenome grows by whole species with N
        sp_len, sp_cov, n_all = synth.metagenome_plan(cfg, args.scale, species_factor=world * (emu if route == "entries" else 1))
        species, genome, n_reads = (sp_len, sp_cov), int(sp_len.sum()), n_all // (world * (emu if route == "entries" else 1))
    error = cfg["error"] if args.error is None else args.error
    t0 = time.perf_counter()
    # same genome on every rank (same seed), different reads per rank.  Single genomes: the counter-based generator
    # (tools/rg_synth.hip) -- rank r makes reads [r * n, (r + 1) * n) of the job; at N = 1 these are exactly the reads the
    # full-size oracle digests of tests/golden/ were computed on.  The metagenome (C4) keeps the torch generator.
    if species is None:
        packed, offsets, lengths = synth.make_reads_device(genome, cfg["read_len"], n_reads, cfg["seed"], error, first=rank * n_reads, device=dev)
    else:
        packed, offsets, lengths = synth.make_reads_torch(genome, cfg["read_len"], n_reads, cfg["seed"], error,
                                                          device=dev, read_seed=cfg["seed"] * 1000 + rank, species=species)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    gen_s = time.perf_counter() - t0
    first_read_id = rank * n_reads
    stream = torch.cuda.current_stream().cuda_stream

    marks = []
    phases = {}   # the reference's step names (TimePrinter, MachineHelper.cpp:605,453,466,583) -> seconds over the timed steps

    def phase(name, t_start):
        phases[name] = phases.get(name, 0.0) + time.perf_counter() - t_start

    def mark(name):
        if args.breakdown:
            torch.cuda.synchronize()
            marks.append((name, time.perf_counter()))

    def one_step(profile=False, keep=False):
        del marks[:]
        mark("start")
        tp = time.perf_counter()
        b = ray_amd.GraphBuilder(k, rank=rank, size=world * emu, device=local_rank, first_read_id=first_read_id, stream=stream, sweep=args.sweep)
        if profile:
            b.set_profiling(True)
        b.add_reads_device(packed.data_ptr(), n_reads * ((cfg["read_len"] + 3) // 4), offsets.data_ptr(),
                           lengths.data_ptr(), n_reads, cfg["read_len"])
        mark("create+add_reads")
        if emu > 1 and route == "entries":
            from ray_amd.graph import RG_ENTRY_BYTES
            b.count_local()
            mark("count_local")
            counts = b.owner_counts()
            send = torch.empty(max(1, sum(counts)) * RG_ENTRY_BYTES, dtype=torch.uint8, device=dev)
            b.export_owner_buckets(send.data_ptr(), max(1, sum(counts)))
            mark("owner buckets")
            # what arrives here: one bucket from each of N ranks holding the same keys -- this rank's own bucket, N times
            mine = send[sum(counts[:rank]) * RG_ENTRY_BYTES:]
            for _ in range(emu):
                b.merge_entries(mine.data_ptr(), counts[rank])
            mark("merge")
            del send, mine
            b.resolve_admission(n_reads)
        elif emu > 1:
            from ray_amd.graph import RG_ENTRY_BYTES
            # records route: this rank's records, all of them counted here in place of the ones N ranks would send
            rsend, n_rec = None, 0
            for part in range(args.parts):
                b.records_local(part, args.parts)
                rcounts, unit = b.record_counts()
                if rsend is None:
                    rsend = torch.empty((int(sum(rcounts) * args.parts * 1.1) + 4096 * args.parts) * unit, dtype=torch.uint8, device=dev)
                b.export_records(rsend.data_ptr() + n_rec * unit, rsend.numel() // unit - n_rec)
                n_rec += sum(rcounts)
            mark("records_local + export")
            b.count_records(rsend.data_ptr(), n_rec)
            del rsend   # (torch keeps the block cached: the exchange buffers of a rank are as good as persistent)
            mark("count_records")
            counts = b.owner_counts()
            send = torch.empty(max(1, sum(counts)) * RG_ENTRY_BYTES, dtype=torch.uint8, device=dev)
            b.export_owner_buckets(send.data_ptr(), max(1, sum(counts)))
            b.release_local()                        # as exchange_owner_buckets does: the aggregation goes before the exchange
            mark("owner buckets")
            b.adopt_entries(send.data_ptr(), sum(counts), keepalive=send)   # as many entries, all distinct, as N ranks would send here; admitted in place
            mark("adopt")
            del send
            b.resolve_admission(n_reads)
        elif world == 1:
            if args.breakdown:
                b.count_local()
                mark("count_local")
                b.resolve_admission()
            else:
                b.count_kmers()                  # RAY_SLAVE_MODE_ADD_VERTICES
        elif comm is not None:
            # records: super-k-mer records to the rank that aggregates their bucket, then the aggregated entries to the rank
            # vertexRank names; entries: every rank aggregates its own reads first.  One call, all inside the library.
            b.count_kmers_exchange(comm, route=route, parts=args.parts, local_reads=n_reads)
        else:
            if route == "records":
                exchange_records(b, device=dev, parts=args.parts)  # super-k-mer records to the rank that aggregates their bucket
            else:
                b.count_local()                       # every rank aggregates its own reads
            mark("records + count")
            exchange_owner_buckets(b, device=dev)     # aggregated entries to the rank vertexRank names
            b.resolve_admission(n_reads)
        mark("count/resolve")
        if profile:          # every call above returns with its stream drained, so wall time per call is the phase's time
            phase("K-mer counting", tp)
        tp = time.perf_counter()
        if comm is not None:
            cov, cnt = b.histogram_exchange(comm)    # RAY_SLAVE_MODE_SEND_DISTRIBUTION, merged on every rank
        else:
            cov, cnt = b.histogram()                 # RAY_SLAVE_MODE_SEND_DISTRIBUTION
            if world > 1:
                cov, cnt = reduce_histogram(cov, cnt, device=dev, size=world)
        mark("histogram")
        if profile:
            phase("Coverage distribution analysis", tp)
        tp = time.perf_counter()
        b.add_edges()                            # RAY_SLAVE_MODE_ADD_EDGES
        mark("add_edges")
        if profile:
            phase("Graph construction", tp)
        st = b.stats()
        sweep = b.kernel_profile() if profile else {}
        if keep:
            return st, sweep, b
        b.close()
        mark("destroy")
        return st, sweep, (cov, cnt)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    kprof = {}
    for _ in range(args.steps):
        st, prof, hist = one_step(profile=True)
        for name, (ms, n) in prof.items():
            a = kprof.setdefault(name, [0.0, 0])
            a[0] += ms
            a[1] += n
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        rdev = dev if backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([st["stored_pairs"], st["occurrences"], st["candidates"]], dtype=torch.int64, device=rdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        stored, occ, cand = [int(x) for x in tot.tolist()]
    else:
        stored, occ, cand = st["stored_pairs"], st["occurrences"], st["candidates"]

    # What follows the three slave modes in a Ray run, measured once after the timed region (not part of `value`):
    # RAY_SLAVE_MODE_PURGE_NULL_EDGES (MachineHelper.cpp:583) and the hand-back of the store to the host GridTable
    # (rg_export: Vertex::dump records to host memory; the host's own inserts are Ray's, not timed here)
    after = {}
    if not args.no_after:
        from ray_amd.exchange import purge_edges_distributed
        _, _, b = one_step(keep=True)
        barrier()
        tp = time.perf_counter()
        if comm is not None:
            b.purge_edges_exchange(comm)
        else:
            purge_edges_distributed(b, device=dev)
        torch.cuda.synchronize()
        after["Null edge purging"] = (time.perf_counter() - tp) * 1e3
        if world == 1 and emu == 1:
            import numpy as np
            from ray_amd.graph import record_dtype
            rec = np.zeros(b.stats()["stored_pairs"], dtype=record_dtype(1 if k <= 31 else 2))   # zero-filled = touched, like the glue's vector<char>
            tp = time.perf_counter()
            n_rec = b.export_into(rec)
            after["store hand-back (rg_export to host memory)"] = (time.perf_counter() - tp) * 1e3
            after["store hand-back records"] = n_rec
            after["store hand-back bytes"] = n_rec * rec.dtype.itemsize
            try:   # the hand-back as a stream (rg_export_stream) with a stand-in for the glue's insert loop that touches every record
                import ctypes as C
                tl = C.CDLL(os.path.join(ROOT, "tools", "librg_synth.so"))
                state = (C.c_uint64 * 3)(0, 0, rec.dtype.itemsize)
                tp = time.perf_counter()
                tl.rgs_touch_records(C.cast(state, C.c_void_p), C.c_void_p(rec.ctypes.data), C.c_uint64(n_rec))
                after["stand-in consumer alone (touches every record, one host thread)"] = (time.perf_counter() - tp) * 1e3
                want_sum = state[0]
                state = (C.c_uint64 * 3)(0, 0, rec.dtype.itemsize)
                b.export_stream(C.cast(tl.rgs_touch_records, C.c_void_p), 0, C.cast(state, C.c_void_p))   # warm: allocates the page-locked staging
                state = (C.c_uint64 * 3)(0, 0, rec.dtype.itemsize)
                tp = time.perf_counter()
                b.export_stream(C.cast(tl.rgs_touch_records, C.c_void_p), 0, C.cast(state, C.c_void_p))
                after["store hand-back as a stream with that consumer (rg_export_stream: copy and consumer overlap)"] = (time.perf_counter() - tp) * 1e3
                assert state[1] == n_rec and state[0] == want_sum, "rg_export_stream handed over other records than rg_export"
            except OSError:
                pass
            try:   # the same into page-locked host memory (what a glue that keeps its buffer registered would see)
                pinned = torch.empty(n_rec * record_dtype(1 if k <= 31 else 2).itemsize, dtype=torch.uint8, pin_memory=True)
                rec2 = pinned.numpy().view(record_dtype(1 if k <= 31 else 2))
                tp = time.perf_counter()
                b.export_into(rec2)
                after["store hand-back into page-locked memory"] = (time.perf_counter() - tp) * 1e3
                del rec2, pinned, rec
            except RuntimeError:
                pass
        b.close()
        barrier()

    # The same workload with 1 % substitution errors, one warm step + two timed ones after everything else (not part of `value`):
    # error-rich reads are what the 8-GPU configurations are made of, and they are 3x further from the roofline.
    error_record = None
    if world == 1 and emu == 1 and species is None and args.error is None and error == 0.0 and not args.no_error_record:
        del packed, offsets, lengths
        torch.cuda.empty_cache()
        ray_amd._lib.load().rg_release_cached_memory()
        packed, offsets, lengths = synth.make_reads_device(genome, cfg["read_len"], n_reads, cfg["seed"], 0.01, first=0, device=dev)
        torch.cuda.synchronize()
        one_step()
        torch.cuda.synchronize()
        tp = time.perf_counter()
        for _ in range(2):
            st_e, _, _ = one_step()
        torch.cuda.synchronize()
        ms_e = (time.perf_counter() - tp) / 2 * 1e3
        bpo = cfg["read_len"] / (4.0 * (cfg["read_len"] - k + 1)) + 8 + 8 * (1 if k <= 31 else 2) + 8
        error_record = {"substitution_error_rate": 0.01, "ms_per_step": ms_e, "steps": 2, "kmer_occurrences": st_e["occurrences"],
                        "distinct_candidates": st_e["candidates"], "vertices_stored": st_e["stored_pairs"],
                        "occurrences_per_s": st_e["occurrences"] / (ms_e / 1e3), "distinct_kmers_per_s": st_e["stored_pairs"] / (ms_e / 1e3),
                        "frac_step": bpo * st_e["occurrences"] / (ms_e / 1e3) / 1e9 / HBM_PEAK_GBS}
        # the CPU baseline below samples the workload's own (error-free) reads again
        del packed, offsets, lengths
        torch.cuda.empty_cache()
        packed, offsets, lengths = synth.make_reads_device(genome, cfg["read_len"], min(n_reads, 16 * args.cpu_sample_reads), cfg["seed"], error, first=0, device=dev)
        torch.cuda.synchronize()

    if rank == 0 and args.breakdown:
        for (n0, t0_), (n1, t1_) in zip(marks, marks[1:]):
            print("  %-18s %8.2f ms" % (n1, (t1_ - t0_) * 1e3), file=sys.stderr)
    if rank == 0:
        per_step = elapsed / args.steps
        # Dominant kernel = the one with the largest HIP-event time per step (events bracket every launch on the
        # launch stream).  Its algorithmic bytes over its time, against the HBM peak.
        L = cfg["read_len"]
        n_per_read = L - k + 1
        W = 1 if k <= 31 else 2
        Kb = 8 * W
        occ0, cand0, recs0 = st["occurrences"], st["candidates"], st["superkmer_records"]
        rec_bytes = 20 if k <= 31 else 8 * (2 * W + 1)  # rg_record_counts reports the same
        ent_bytes = 16 if W == 1 else 32                # an admitted entry between the log and the table (rg_table.h)
        scatter_levels = st["partition_levels"] - (1 if st.get("folded_level_bits") else 0)   # levels that are a k_scatter pass of their own
        log_bytes = 16 if st.get("compact_log") == 1 else 32   # an entry of the sweep's log (16: key + meta, the admission's item carries the stamp)
        alg_bytes = {
            "k_count": occ0 * (L / (4.0 * n_per_read) + Kb + 8),            # read bits + key compare + count RMW
            "k_superkmers": occ0 * L / (4.0 * n_per_read) + recs0 * rec_bytes,  # read bits in, records out
            "k_superkmers_lanes": occ0 * L / (4.0 * n_per_read) + recs0 * (rec_bytes + (1 if st.get("folded_level_bits") else 0)),  # same bytes, a lane per read (+ a digit byte per record when it does the first level)
            "k_scatter": scatter_levels * 2 * recs0 * rec_bytes + max(0, scatter_levels - 1) * recs0,  # records in + out per level, + the next level's digit byte
            "k_level_hist": max(0, st["partition_levels"] - 1) * recs0 * rec_bytes,
            "k_level_hist_side": max(0, st["partition_levels"] - 1) * recs0,     # one digit byte per record and level
            "k_aggregate": recs0 * rec_bytes + st["log_entries"] * (log_bytes + (17 if st.get("items_from_sweep") else 0)),  # records in, entries out (+ the admission's 16-byte item and digit byte per entry when it makes them)
            "k_merge": st["log_entries"] * 32 + cand0 * 32,                  # entries in, slots written
            "k_bloom_items": (st["log_entries"] if "k_insert_admitted" in kprof else st["table_slots"]) * 32 + cand0 * 64,  # scan in, 8 items of 8 B per key out
            "k_bloom_hist": (st["log_entries"] if "k_insert_admitted" in kprof else st["table_slots"]) * 32,   # key scan, nothing written
            "k_bloom_scatter_items": (st["log_entries"] if "k_insert_admitted" in kprof else st["table_slots"]) * 32 + cand0 * 64,  # key scan in, 8 items of 8 B per key out, already in level-0 order
            "k_insert_admitted": st["log_entries"] * 32 + st["stored_pairs"] * 32,   # log in, admitted slots written
            "k_scatter64": 2 * cand0 * 64,                                    # items in + out, per level (x launches below)
            "k_bloom_region": cand0 * 64,                                     # items in (contested stamps are extra)
            "k_bloom_touch": cand0 * (32 + 8 * 8),                           # slot + 8 stamp words
            "k_admit": cand0 * (32 + 8 + 8),                                 # slot + >= 1 stamp word + coverage write
            "bloom_stamp_fill": st["bloom_bits"] * 8,
            "k_histogram": st["table_slots"] * 32,
            "k_entry_hist": st["log_entries"] * 32,                                                  # log scan
            "k_admit_count": st["log_entries"] * 33,                                                 # log scan + one mark byte per entry, nothing written
            "k_scatter_entries": st["log_entries"] * (log_bytes + 1) + st["stored_pairs"] * ent_bytes,                   # first level: log in, admitted entries out (further levels added below)
            "k_build_ranges": st["stored_pairs"] * ent_bytes + st["table_slots"] * 32,                      # entries in, every slot of the table written once
            "k_xor_hist": (st["log_entries"] if "k_build_ranges" in kprof or "k_insert_admitted" in kprof else st["table_slots"]) * 32,
            "k_xor_keys": (st["log_entries"] if "k_build_ranges" in kprof or "k_insert_admitted" in kprof else st["table_slots"]) * 32 + cand0 * 17,  # key scan in, one 16-byte item + digit byte out
            "k_scatter16": cand0 * (16 + 12),                                                        # last level: items in, packed word + slot index out (middle levels added below)
            "k_xor_region": 8 * cand0 * 8 + cand0,                                                   # every packed word once per hash, 8 win bits per key out
            "k_xor_fp": cand0 * (8 + 4 + 1),
        }
        if "k_build_ranges" in kprof:
            del alg_bytes["k_insert_admitted"]   # only the few entries that left their range go through it: no byte model
        if kprof.get("k_scatter_entries", (0, 0))[1] > args.steps:
            alg_bytes["k_scatter_entries"] += (kprof["k_scatter_entries"][1] / args.steps - 1) * 2 * st["stored_pairs"] * ent_bytes
        if kprof.get("k_scatter16", (0, 0))[1] > args.steps:
            alg_bytes["k_scatter16"] += (kprof["k_scatter16"][1] / args.steps - 1) * cand0 * 33
        if "k_scatter64" in kprof:
            alg_bytes["k_scatter64"] *= kprof["k_scatter64"][1] / args.steps
        per_kernel = {name: {"ms_per_step": ms / args.steps, "launches_per_step": n / args.steps,
                             "algorithmic_GBps": (alg_bytes[name] / (ms / args.steps / 1e3) / 1e9) if name in alg_bytes and ms else None}
                      for name, (ms, n) in kprof.items()}
        dom = max(kprof, key=lambda nm: kprof[nm][0]) if kprof else None
        kernel_s = (kprof[dom][0] / 1e3) / args.steps if dom else float("nan")
        if dom and dom not in alg_bytes:   # a kernel without a byte model (retries): the roofline line is about the largest one that has
            dom = max((nm for nm in kprof if nm in alg_bytes), key=lambda nm: kprof[nm][0], default=None)
        achieved = per_kernel[dom]["algorithmic_GBps"] if dom else None
        sweep_ms = kprof[dom][0] if dom else 0.0
        sweep_launches = kprof[dom][1] if dom else 0
        bytes_per_occ = (alg_bytes[dom] / occ0) if dom and occ0 else None
        pass1_bytes_per_occ = L / (4.0 * n_per_read) + 8 + Kb + 8         # stream + Bloom + table
        # HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes of this same command
        # (tools/profile_c2.sh; FETCH_SIZE and WRITE_SIZE in separate runs).  FETCH_SIZE is doubled as
        # MI355X_MICROARCH.md prescribes for gfx950 (checked here on k_scatter / k_level_hist / k_init_table,
        # whose byte counts are known exactly).
        # The summary is quoted only when it was made from the very sources this library is built from (sha256 over
        # ray_amd/csrc, tools/csrc_hash.py) and shows the launches this run makes; otherwise traffic is null.
        traffic = traffic_step = None
        traffic_note = "no PMC summary for this workload"
        try:
            if args.workload.startswith("C2") and args.scale == 1.0 and world == 1 and dom and error == 0.0:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                from csrc_hash import csrc_sha256
                with open(os.path.join(ROOT, PMC_SUMMARY)) as fh:
                    pmc = json.load(fh)
                pm = pmc.get(dom)
                launches_now = kprof[dom][1] / args.steps
                if pmc.get("_csrc_sha256") != csrc_sha256():
                    traffic_note = "the stored PMC summary was made from other kernel sources: not quoted (regenerate with tools/profile_c2.sh)"
                elif not pm or not pm.get("launches") or abs(pm["launches"] - launches_now) > 2:   # (the profiled step's record sample adds up to two small launches)
                    traffic_note = "the stored PMC summary shows %s launches of %s per step, this run %.1f: not quoted" % (pm.get("launches") if pm else None, dom, launches_now)
                else:
                    traffic = (2.0 * pm.get("FETCH_SIZE_bytes", 0.0) + pm.get("WRITE_SIZE_bytes", 0.0)) / max(1.0, launches_now)
                    traffic_step = sum(2.0 * v.get("FETCH_SIZE_bytes", 0.0) + v.get("WRITE_SIZE_bytes", 0.0)
                                       for kn, v in pmc.items() if isinstance(v, dict)) / max(1, pmc.get("_steps", 1))
                    traffic_note = "bytes per launch, rocprofv3 PMC passes of the same command and the same kernel sources (%s, sha256 %s)" % (PMC_SUMMARY, pmc["_csrc_sha256"][:12])
        except (OSError, ValueError, ImportError):
            traffic = traffic_step = None
        launches_dom = (kprof[dom][1] / args.steps) if dom else 0
        step_gbs = pass1_bytes_per_occ * occ0 / per_step / 1e9
        out = {
            "metric": "distinct k-mers inserted/sec (whole node)",
            "value": stored / per_step,
            "unit": "distinct canonical k-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": per_step * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": args.workload + ("" if args.scale == 1.0 else "@scale=%g" % args.scale) + ("" if args.error is None else "@error=%g" % args.error), "k": k,
                       "read_len": L, "reads_per_gpu": n_reads, "genome_bp": genome, "substitution_error_rate": error,
                       "parallelism": "owner-partitioned x%d" % world + ("" if nranks == 1 else ", %s route" % route) + ("" if emu == 1 else " (rank 0 of %d emulated on one GPU, no network)" % emu) + ("" if world == 1 else (", exchange inside the library (rg_comm, RCCL)" if comm is not None else ", exchange over torch.distributed")), "bloom": "default 16*k*reads cap 2^33"},
            "occurrences_per_s": occ / per_step,
            "kmer_occurrences": occ, "distinct_candidates": cand, "vertices_stored": stored,
            # frac = frac_step: SURVEY 8(d)'s bytes per occurrence (stream + Bloom + table: what the path must move whatever
            # the algorithm) x the occurrences of a step / the whole step's time.  achieved / frac_kernel: the dominant
            # kernel's own algorithmic bytes per launch / its average launch duration (HIP events on the launch stream).
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": step_gbs / HBM_PEAK_GBS, "frac_step": step_gbs / HBM_PEAK_GBS,
                         "frac_kernel": (achieved / HBM_PEAK_GBS) if achieved else None,
                         "achieved_step": step_gbs, "step_bytes_per_occurrence": pass1_bytes_per_occ,
                         "traffic": traffic, "traffic_step": traffic_step,
                         "traffic_unit": traffic_note,
                         "algorithmic_bytes_per_launch": (alg_bytes[dom] / launches_dom) if dom in alg_bytes and launches_dom else None,
                         "kernel": dom, "bytes_per_occurrence": bytes_per_occ,
                         "kernel_ms_per_step": kernel_s * 1e3 if sweep_ms else None,
                         "launches_per_step": sweep_launches / args.steps},
            "kernels": per_kernel,
            "pass1_algorithmic": {"bytes_per_occurrence": pass1_bytes_per_occ,
                                  "GBps_over_whole_step": pass1_bytes_per_occ * occ0 / per_step / 1e9,
                                  "frac_of_hbm_peak": pass1_bytes_per_occ * occ0 / per_step / 1e9 / HBM_PEAK_GBS},
            "phases_ms_per_step": {name: sec / args.steps * 1e3 for name, sec in phases.items()},
            "host_syncs_per_step": st.get("host_syncs"),   # times the host waited for the stream (sizes read back for the next grid / allocation)
            "kernels_ms_per_step": sum(ms for ms, _ in kprof.values()) / args.steps,
            "after_the_timed_step_ms": after,
            "error_1pct": error_record,
            "setup_s": gen_s,
        }
        if not args.no_cpu_baseline and world == 1:
            cores = args.cpu_cores if args.cpu_cores > 0 else max(1, min(16, os.cpu_count() or 1))
            out["cpu_baseline"] = cpu_baseline(packed, offsets, lengths, k, args.cpu_sample_reads, cores)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
