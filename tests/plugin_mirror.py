"""Test scaffolding: a host-side mirror of Ray's three graph-construction plugins, for this path only.

Same names, argument meaning and completion behaviour as the reference's handlers, so the parity
tests read like tests of Ray's own plugins:

  KmerAcademyBuilder.call_RAY_SLAVE_MODE_ADD_VERTICES   (KmerAcademyBuilder/KmerAcademyBuilder.cpp:43-211)
  CoverageGatherer.call_RAY_SLAVE_MODE_SEND_DISTRIBUTION (CoverageGatherer/CoverageGatherer.cpp:124-190)
  VerticesExtractor.call_RAY_SLAVE_MODE_ADD_EDGES        (VerticesExtractor/VerticesExtractor.cpp:40-274)
  GridTable.size / find                                  (VerticesExtractor/GridTable.cpp:80-134)

The handler bodies call the C ABI (include/raygraph.h); nothing is computed here.  In Ray itself the
same calls go into the C++ handlers.  Each handler runs to completion in one call
(the reference re-enters it once per engine tick) and then reports the completion tag the reference
sends (RAY_MPI_TAG_KMER_ACADEMY_DISTRIBUTED, RAY_MPI_TAG_VERTICES_DISTRIBUTED, RAY_MPI_TAG_COVERAGE_END).
"""
import os

import numpy as np

from ray_amd._lib import RG_BLOOM_DEFAULT
from ray_amd.graph import GraphBuilder, coverage_distribution

CONFIG_RAY_VERSION = "2.3.2-devel"


class Parameters:
    """The slice of Mock/Parameters the path reads: -k, rank/size, -bloom-filter-bits, -o, -write-kmers."""

    def __init__(self, k=21, rank=0, size=1, bloom_filter_bits=None, prefix=None, write_kmers=False,
                 maximum_kmer_length=None, device=0):
        # Parameters.cpp:892-907: even k becomes k-1
        if k % 2 == 0:
            k -= 1
        self._k, self._rank, self._size = k, rank, size
        self._bloom = bloom_filter_bits
        self._prefix = prefix
        self._write_kmers = write_kmers
        self._maxk = maximum_kmer_length if maximum_kmer_length else (32 if k <= 31 else 64)
        self.device = device

    def getWordSize(self):
        return self._k

    def getRank(self):
        return self._rank

    def getSize(self):
        return self._size

    def getPrefix(self):
        return self._prefix

    def writeKmers(self):
        return self._write_kmers

    def getColorSpaceMode(self):
        return False

    def getKmerWords(self):
        """KMER_U64_ARRAY_SIZE for CONFIG_MAXKMERLENGTH (Kmer.h:46-61)"""
        return 1 if self._maxk <= 32 else 2

    def getBloomFilterBits(self):
        """-bloom-filter-bits (MessageProcessor.cpp:892-893); None = the default formula"""
        return RG_BLOOM_DEFAULT if self._bloom is None else int(self._bloom)

    def getCoverageDistributionFile(self):
        return os.path.join(self._prefix, "CoverageDistribution.txt")


class Vertex:
    """what GridTable::find hands back: coverage and the edge byte in the queried k-mer's orientation"""

    def __init__(self, coverage, edges):
        self._coverage, self._edges = int(coverage), int(edges)

    def getCoverage(self, kmer=None):
        return self._coverage

    def getEdges(self, kmer=None):
        return self._edges

    def getIngoingEdges(self, kmer=None, k=None):
        return [c for c in range(4) if self._edges & (1 << c)]

    def getOutgoingEdges(self, kmer=None, k=None):
        return [c for c in range(4) if self._edges & (1 << (4 + c))]


class GridTable:
    """m_subgraph: this rank's partition, resident on the GPU"""

    def __init__(self, parameters, first_read_id=0, table_slots=0, stream=None):
        self.m_parameters = parameters
        self.builder = GraphBuilder(parameters.getWordSize(), rank=parameters.getRank(), size=parameters.getSize(),
                                    device=parameters.device, bloom_bits=parameters.getBloomFilterBits(),
                                    kmer_words=parameters.getKmerWords(), table_slots=table_slots,
                                    first_read_id=first_read_id, stream=stream)

    def size(self):
        return self.builder.size_()

    def find(self, kmer_words):
        cov, e = self.builder.find(np.asarray(kmer_words, dtype=np.uint64).reshape(1, -1))
        return Vertex(cov[0], e[0]) if cov[0] else None

    def export(self):
        return self.builder.export()


class KmerAcademyBuilder:
    def __init__(self, parameters, subgraph, exchange=None):
        self.m_parameters, self.m_subgraph = parameters, subgraph
        self.m_finished = False
        self._exchange = exchange  # callable(engine) for size > 1

    def setReads(self, packed, offsets, lengths):
        self.m_subgraph.builder.add_reads(packed, offsets, lengths)

    def call_RAY_SLAVE_MODE_ADD_VERTICES(self):
        if self.m_finished:
            return None
        b = self.m_subgraph.builder
        if self.m_parameters.getSize() == 1:
            b.count_kmers()
        else:
            b.count_local()
            self._exchange(b)
            b.resolve_admission()
        self.m_finished = True
        return "RAY_MPI_TAG_KMER_ACADEMY_DISTRIBUTED"

    def finished(self):
        return self.m_finished


class CoverageGatherer:
    def __init__(self, parameters, subgraph):
        self.m_parameters, self.m_subgraph = parameters, subgraph
        self.m_distributionOfCoverage = None

    def call_RAY_SLAVE_MODE_SEND_DISTRIBUTION(self):
        cov, cnt = self.m_subgraph.builder.histogram()
        self.m_distributionOfCoverage = (cov, cnt)
        return "RAY_MPI_TAG_COVERAGE_END"

    def writeKmers(self, path, append=False):
        """CoverageGatherer::writeKmers (CoverageGatherer.cpp:47-122); line order is table order"""
        from kmers_txt import write_kmers
        write_kmers(self.m_subgraph.builder, path, self.m_parameters, append=append)


class VerticesExtractor:
    def __init__(self, parameters, subgraph):
        self.m_parameters, self.m_subgraph = parameters, subgraph
        self.m_finished = False

    def getDefaultNumberOfBitsForBloomFilter(self, local_reads):
        """VerticesExtractor.cpp:276-320"""
        bits = 4 * (self.m_parameters.getWordSize() * 2 * 2) * local_reads
        return min(bits, 1024 * 1024 * 1024 * 8)

    def call_RAY_SLAVE_MODE_ADD_EDGES(self):
        if self.m_finished:
            return None
        self.m_subgraph.builder.add_edges()
        self.m_finished = True
        return "RAY_MPI_TAG_VERTICES_DISTRIBUTED"

    def finished(self):
        return self.m_finished


def build_graph(packed, offsets, lengths, k, bloom_filter_bits=None, prefix=None, purge=True, maximum_kmer_length=None,
                write_kmers=False, device=0):
    """Single-rank master-mode sequence of MachineHelper.cpp:1223-1231 for this path:
    ADD_VERTICES -> SEND_DISTRIBUTION -> (master) CoverageDistribution.txt -> ADD_EDGES -> PURGE_NULL_EDGES
    -> WRITE_KMERS.  Returns (subgraph, (coverage, count), (minimum, peak, repeat))."""
    params = Parameters(k=k, bloom_filter_bits=bloom_filter_bits, prefix=prefix, write_kmers=write_kmers,
                        maximum_kmer_length=maximum_kmer_length, device=device)
    subgraph = GridTable(params)
    academy = KmerAcademyBuilder(params, subgraph)
    academy.setReads(packed, offsets, lengths)
    gatherer = CoverageGatherer(params, subgraph)
    extractor = VerticesExtractor(params, subgraph)
    assert academy.call_RAY_SLAVE_MODE_ADD_VERTICES() == "RAY_MPI_TAG_KMER_ACADEMY_DISTRIBUTED"
    assert gatherer.call_RAY_SLAVE_MODE_SEND_DISTRIBUTION() == "RAY_MPI_TAG_COVERAGE_END"
    cov, cnt = gatherer.m_distributionOfCoverage
    path = None
    if prefix:
        os.makedirs(prefix, exist_ok=True)
        path = params.getCoverageDistributionFile()
    values = coverage_distribution(cov, cnt, path) if len(cov) else None
    assert extractor.call_RAY_SLAVE_MODE_ADD_EDGES() == "RAY_MPI_TAG_VERTICES_DISTRIBUTED"
    if purge:
        subgraph.builder.purge_edges()
    if write_kmers and prefix:
        gatherer.writeKmers(os.path.join(prefix, "kmers.txt"))
    return subgraph, (cov, cnt), values
