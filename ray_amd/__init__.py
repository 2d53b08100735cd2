"""ray_amd -- MI355X-native de Bruijn graph construction for the Ray assembler's hot path.

Only what the path needs: csrc/ (HIP kernels + the C ABI of include/raygraph.h), the ctypes
binding and the multi-GPU exchange.  Importing the package loads the in-tree HIP library and fails if it is not built.
"""
from . import _lib

_lib.load()

from .graph import (Comm, GraphBuilder, RayGraphError, count_sequences, coverage_distribution, pack_read, pack_reads,  # noqa: E402
                    read_genome_graph, write_coverage_analysis, write_coverage_checkpoint, write_genome_graph,
                    write_graph_partition)

__all__ = ["GraphBuilder", "Comm", "RayGraphError", "coverage_distribution", "pack_read", "pack_reads", "count_sequences",
           "write_coverage_checkpoint", "write_genome_graph", "write_graph_partition", "write_coverage_analysis",
           "read_genome_graph"]
