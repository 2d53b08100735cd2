"""Loader of the in-tree HIP library (ray_amd/libraygraph.so).

There is no fallback: if the library is missing or does not load, importing the product path
raises.  The library itself refuses to work without a HIP device (rg_create -> RG_ERR_HIP).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libraygraph.so")

RG_OK = 0
RG_ERR_ARG, RG_ERR_HIP, RG_ERR_NOMEM, RG_ERR_STATE, RG_ERR_CAPACITY, RG_ERR_HASH, RG_ERR_COMM = -1, -2, -3, -4, -5, -6, -7
RG_COMM_ID_BYTES = 256
RG_ROUTE_RECORDS, RG_ROUTE_ENTRIES = 1, 2
EXPORT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64)   # rg_export_fn
ALLTOALL_U64_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint32)
ALLTOALLV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32)


class RgTransport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("alltoall_u64", ALLTOALL_U64_FN), ("alltoallv", ALLTOALLV_FN)]


class RgCommStats(C.Structure):
    _fields_ = [("bytes_sent", C.c_uint64), ("bytes_received", C.c_uint64), ("collectives", C.c_uint64), ("wait_seconds", C.c_double)]
HASH_FN = C.CFUNCTYPE(C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p)   # rg_hash_fn
RG_BLOOM_DEFAULT = (1 << 64) - 1
RG_ENTRY_BYTES = 32
UINT64_MAX = (1 << 64) - 1
RG_NOALLOC_MAX_READS = 262144


class RgConfig(C.Structure):
    _fields_ = [("k", C.c_int32), ("kmer_words", C.c_int32), ("rank", C.c_int32), ("size", C.c_int32),
                ("device", C.c_int32), ("reserved0", C.c_int32), ("bloom_bits", C.c_uint64),
                ("table_slots", C.c_uint64), ("first_read_id", C.c_uint64), ("stream", C.c_void_p)]


class RgStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("reads", "occurrences", "candidates", "stored_pairs", "bloom_bits",
                                         "bloom_insertions", "bloom_set_bits", "table_slots", "sweep_partitioned", "superkmer_records", "partition_levels", "log_entries", "bloom_partitioned", "host_syncs", "folded_level_bits", "items_from_sweep", "compact_log")]


# every symbol include/raygraph.h declares
SYMBOLS = ["rg_create", "rg_destroy", "rg_last_error", "rg_add_reads", "rg_add_reads_device", "rg_count_kmers",
           "rg_count_local", "rg_records_local", "rg_records_local_part", "rg_record_counts", "rg_export_records", "rg_count_records", "rg_owner_counts", "rg_export_owner_buckets", "rg_merge_entries",
           "rg_resolve_admission", "rg_histogram", "rg_add_edges", "rg_purge_edges", "rg_purge_query_counts",
           "rg_purge_export_queries", "rg_purge_answer", "rg_purge_apply", "rg_size", "rg_export", "rg_export_stream",
           "rg_find", "rg_add_reads_text", "rg_count_sequences", "rg_load_sequences", "rg_get_reads", "rg_pack_reads_noalloc", "rg_release_local", "rg_adopt_entries", "rg_check_hashes", "rg_hash_kmers", "rg_read_markers", "rg_markers_query_counts", "rg_markers_query_counts_part", "rg_markers_export_queries", "rg_markers_answer", "rg_markers_apply",
           "rg_import_vertices", "rg_write_kmers_txt", "rg_write_genome_graph", "rg_write_coverage_checkpoint", "rg_write_graph_partition", "rg_write_coverage_analysis", "rg_read_genome_graph", "rg_degree_distribution", "rg_get_stats", "rg_set_profiling", "rg_debug_set", "rg_get_kernel_profile", "rg_release_cached_memory", "rg_coverage_distribution", "rg_pack_read",
           "rg_comm_unique_id", "rg_comm_create_rccl", "rg_comm_create_transport", "rg_comm_destroy", "rg_comm_last_error", "rg_comm_get_stats",
           "rg_exchange_records", "rg_exchange_owner_buckets", "rg_count_kmers_exchange", "rg_histogram_exchange", "rg_purge_edges_exchange",
           "rg_read_markers_exchange", "rg_allgather_u64",
           "rg_version"]

_lib = None


def load():
    """dlopen libraygraph.so and bind the C ABI; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: build it with `make -C ray_amd/csrc` (or __graft_entry__.build()); "
                          "ray_amd has no CPU fallback" % LIB_PATH)
    # torch ships its own libamdhip64 (same SONAME as /opt/rocm's).  A process must hold ONE HIP runtime:
    # importing torch first makes the dynamic loader resolve our NEEDED libamdhip64.so.7 to the copy torch
    # already loaded; the other order loads two runtimes and the second one finds no GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
    u64p, u32p, u8p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)
    i32p = C.POINTER(C.c_int32)
    sig = {
        "rg_create": ([C.POINTER(RgConfig), C.POINTER(vp)], C.c_int),
        "rg_destroy": ([vp], C.c_int),
        "rg_last_error": ([vp], C.c_char_p),
        "rg_add_reads": ([vp, vp, u64, vp, vp, u64], C.c_int),
        "rg_add_reads_device": ([vp, vp, u64, vp, vp, u64, u32], C.c_int),
        "rg_count_kmers": ([vp], C.c_int),
        "rg_count_local": ([vp], C.c_int),
        "rg_owner_counts": ([vp, u64p], C.c_int),
        "rg_records_local": ([vp], C.c_int),
        "rg_records_local_part": ([vp, C.c_uint32, C.c_uint32], C.c_int),
        "rg_record_counts": ([vp, u64p, u32p], C.c_int),
        "rg_export_records": ([vp, vp, u64], C.c_int),
        "rg_count_records": ([vp, vp, u64], C.c_int),
        "rg_export_owner_buckets": ([vp, vp, u64], C.c_int),
        "rg_merge_entries": ([vp, vp, u64], C.c_int),
        "rg_resolve_admission": ([vp, u64], C.c_int),
        "rg_histogram": ([vp, u32p, u64p, u64, u64p], C.c_int),
        "rg_add_edges": ([vp], C.c_int),
        "rg_purge_edges": ([vp], C.c_int),
        "rg_purge_query_counts": ([vp, u64p], C.c_int),
        "rg_purge_export_queries": ([vp, vp, u64], C.c_int),
        "rg_purge_answer": ([vp, vp, u64, vp], C.c_int),
        "rg_purge_apply": ([vp, vp, u64], C.c_int),
        "rg_size": ([vp, u64p], C.c_int),
        "rg_export": ([vp, vp, u64, u64p], C.c_int),
        "rg_export_stream": ([vp, u64, C.c_void_p, vp], C.c_int),
        "rg_find": ([vp, u64p, u64, u32p, u8p], C.c_int),
        "rg_degree_distribution": ([vp, u64p], C.c_int),
        "rg_add_reads_text": ([vp, vp, u64, vp, vp, u64], C.c_int),
        "rg_count_sequences": ([C.c_char_p, u64p], C.c_int),
        "rg_load_sequences": ([vp, C.c_char_p, u64, u64, u64p], C.c_int),
        "rg_get_reads": ([vp, vp, u64, vp, vp, u64, u64p, u64p], C.c_int),
        "rg_pack_reads_noalloc": ([vp, vp, u64, vp, vp, u64, vp, u64, vp, vp], C.c_int),
        "rg_read_markers": ([vp, vp, vp], C.c_int),
        "rg_release_local": ([vp], C.c_int),
        "rg_adopt_entries": ([vp, vp, u64], C.c_int),
        "rg_check_hashes": ([vp, HASH_FN, HASH_FN, vp, C.c_uint32], C.c_int),
        "rg_hash_kmers": ([vp, u64p, u64, u64p, u64p], C.c_int),
        "rg_markers_query_counts": ([vp, u64p], C.c_int),
        "rg_markers_query_counts_part": ([vp, C.c_uint32, C.c_uint32, u64p], C.c_int),
        "rg_markers_export_queries": ([vp, vp, u64], C.c_int),
        "rg_markers_answer": ([vp, vp, u64, vp], C.c_int),
        "rg_markers_apply": ([vp, vp, u64, vp, vp], C.c_int),
        "rg_import_vertices": ([vp, vp, u64], C.c_int),
        "rg_write_kmers_txt": ([vp, u64, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int], C.c_int),
        "rg_write_genome_graph": ([vp, u64, C.c_int, C.c_int, C.c_char_p], C.c_int),
        "rg_write_coverage_checkpoint": ([u32p, u64p, u64, C.c_char_p], C.c_int),
        "rg_write_graph_partition": ([u64p, C.c_int, C.c_char_p], C.c_int),
        "rg_write_coverage_analysis": ([u32p, u64p, u64, C.c_int, C.c_int32, C.c_int32, C.c_int32, C.c_char_p, C.c_char_p], C.c_int),
        "rg_read_genome_graph": ([C.c_char_p, C.c_int, C.c_int, vp, u64, u64p], C.c_int),
        "rg_get_stats": ([vp, C.POINTER(RgStats)], C.c_int),
        "rg_release_cached_memory": ([], C.c_int),
        "rg_set_profiling": ([vp, C.c_int], C.c_int),
        "rg_debug_set": ([vp, C.c_uint32], C.c_int),
        "rg_get_kernel_profile": ([vp, C.c_char_p, u64], C.c_int),
        "rg_coverage_distribution": ([u32p, u64p, u64, C.c_char_p, i32p, i32p, i32p], C.c_int),
        "rg_pack_read": ([C.c_char_p, C.c_int, u8p], C.c_int),
        "rg_comm_unique_id": ([vp], C.c_int),
        "rg_comm_create_rccl": ([vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)], C.c_int),
        "rg_comm_create_transport": ([C.POINTER(RgTransport), C.c_int, C.c_int, C.c_int, C.POINTER(vp)], C.c_int),
        "rg_comm_destroy": ([vp], C.c_int),
        "rg_comm_last_error": ([vp], C.c_char_p),
        "rg_comm_get_stats": ([vp, C.POINTER(RgCommStats)], C.c_int),
        "rg_exchange_records": ([vp, vp, C.c_uint32], C.c_int),
        "rg_exchange_owner_buckets": ([vp, vp], C.c_int),
        "rg_count_kmers_exchange": ([vp, vp, C.c_int, C.c_uint32, u64], C.c_int),
        "rg_histogram_exchange": ([vp, vp, u32p, u64p, u64, u64p], C.c_int),
        "rg_purge_edges_exchange": ([vp, vp], C.c_int),
        "rg_read_markers_exchange": ([vp, vp, C.c_uint32, vp, vp], C.c_int),
        "rg_allgather_u64": ([vp, vp, u64, u64p], C.c_int),
        "rg_version": ([], C.c_char_p),
    }
    for name in SYMBOLS:
        f = getattr(L, name)  # AttributeError if the library does not export it
        f.argtypes, f.restype = sig[name]
    _lib = L
    return L
