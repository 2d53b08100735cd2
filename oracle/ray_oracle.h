/*
 * ray_oracle.h -- CPU restatement of sebhtml/Ray's de Bruijn graph construction path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under ray_amd/ (the product) may include, link or
 * call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and only as the checker / the timed CPU baseline.
 *
 * PARITY STATUS (see DESIGN.md "Oracle"):
 *   - the reference cannot be built here: every hot-path source includes headers of
 *     RayPlatform (absent: /root/reference/RayPlatform is a dangling symlink), so there is no
 *     oracle/_ref.  The restatement is pinned by the known answers the reference holds in its
 *     own comments (tests/golden/reference_known_answers.json cites them) and by the values
 *     SURVEY.md section 8c recorded from the reference's own objects.
 *   - hash-derived values (owner rank, Bloom bit indices) are PARITY UNPINNED: the two mixes
 *     uniform_hashing_function_{1,2}_64_64 live in RayPlatform/cryptography/crypto.cpp, which
 *     is not in the container (no version pin exists in the reference either).  They are
 *     restated from the published Thomas Wang 64-bit mixes in ro_mix1/ro_mix2 only.
 *
 * All citations are relative to /root/reference/code/.
 */
#ifndef RAY_ORACLE_H
#define RAY_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- A0: Read storage (SequencesLoader/Read.cpp:31-68 trim, :76-159 constructor, :161-174 getSeq) */
/* packs `seq` (NUL terminated) the way Read::constructor does; returns the stored length.
 * `packed` needs (strlen(seq)+3)/4 bytes. */
int ro_read_pack(const char *seq, int trim_flag, uint8_t *packed);
void ro_read_get_seq(const uint8_t *packed, int length, char *out);
int ro_read_required_bytes(int length); /* Read.cpp:206-220 */

/* ---- A1..A4: Kmer value type.  A k-mer is 2 words; word 1 is 0 when W==1. */
void ro_word_id(const char *ascii, uint64_t out[2]);                        /* Mock/common_functions.cpp:189-207 */
int ro_is_valid_dna(const char *x);                                         /* Mock/common_functions.cpp:41-49 */
void ro_complement_vertex(const uint64_t in[2], int k, uint64_t out[2]);    /* KmerAcademyBuilder/Kmer.cpp:398-420 */
int ro_less(const uint64_t a[2], const uint64_t b[2], int W);               /* Kmer.cpp:38-47 */
uint64_t ro_mix1(uint64_t x); /* RayPlatform crypto.cpp uniform_hashing_function_1_64_64 -- ASSUMED, unpinned */
uint64_t ro_mix2(uint64_t x); /* RayPlatform crypto.cpp uniform_hashing_function_2_64_64 -- ASSUMED, unpinned */
uint64_t ro_hash1(const uint64_t a[2], int W);                              /* Kmer.cpp:284-296 */
uint64_t ro_hash2(const uint64_t a[2], int W);                              /* Kmer.cpp:298-310 */
int ro_vertex_rank(const uint64_t a[2], int k, int W, int size);            /* Kmer.cpp:165-170 */
void ro_id_to_word(const uint64_t a[2], int k, char *out);                  /* Kmer.cpp:312-336 */
int ro_outgoing_edges(const uint64_t a[2], uint8_t edges, int k, int W, uint64_t out[4][2]); /* Kmer.cpp:176-214 */
int ro_ingoing_edges(const uint64_t a[2], uint8_t edges, int k, int W, uint64_t out[4][2]);  /* Kmer.cpp:220-282 */

/* ---- A6: Bloom filter (KmerAcademyBuilder/BloomFilter.cpp) */
void ro_bloom_bit_indices(const uint64_t lower[2], int W, uint64_t m_bits, uint64_t out[8]); /* :124-150 */
uint64_t ro_default_bloom_bits(int k, uint64_t local_reads); /* VerticesExtractor/VerticesExtractor.cpp:276-320 */

/* ---- A10/A11: Vertex edge bitmap (VerticesExtractor/Vertex.cpp) */
uint8_t ro_convert_bitmap(uint8_t map);                                     /* Vertex.cpp:341-361 */
/* edges of `vertex` (any strand) after addOutgoingEdge/addIngoingEdge on a fresh Vertex whose
 * stored key is the lower one; returns the stored byte (Vertex.cpp:110-173). */
uint8_t ro_add_outgoing_edge(uint8_t stored, const uint64_t vertex[2], const uint64_t child[2], int k, int W);
uint8_t ro_add_ingoing_edge(uint8_t stored, const uint64_t vertex[2], const uint64_t parent[2], int k, int W);
uint8_t ro_get_edges(uint8_t stored, const uint64_t vertex[2], int k, int W); /* Vertex.cpp:169-173 */

/* ---- A5,A7..A12: the distributed graph, R ranks, sequential and message-faithful ---- */
typedef struct ro_graph ro_graph;

#define RO_BLOOM_DEFAULT (~(uint64_t)0) /* use 16*k*localReads capped at 2^33 (per rank) */

ro_graph *ro_graph_create(int k, int W, int nranks, uint64_t bloom_bits /* RO_BLOOM_DEFAULT, 0=off, or -bloom-filter-bits */);
void ro_graph_destroy(ro_graph *g);
/* reads: concatenated Read::getRawSequence() buffers; offsets in bytes; lengths in bases.
 * Global read i belongs to rank per SequencesLoader/SequencesLoader.cpp:287-297. */
int ro_graph_set_reads(ro_graph *g, const uint8_t *packed, const uint64_t *offsets, const uint32_t *lengths, uint64_t nreads);
int ro_graph_count_kmers(ro_graph *g); /* KmerAcademyBuilder.cpp:43-211 + MessageProcessor.cpp:692-809, then :984-1046 */
int ro_graph_add_edges(ro_graph *g);   /* VerticesExtractor.cpp:40-274 + MessageProcessor.cpp:858-968 */
int ro_graph_purge_edges(ro_graph *g); /* EdgePurger/EdgePurgerWorker.cpp:28-102 + MessageProcessor.cpp:1224-1256 */
uint64_t ro_graph_size(const ro_graph *g, int rank /* -1 = all ranks */); /* GridTable.cpp:80-82,129-131: 2 per stored pair */
uint64_t ro_graph_bloom_bits(const ro_graph *g, int rank);
uint64_t ro_graph_bloom_set_bits(const ro_graph *g, int rank);   /* BloomFilter.cpp:224-226 (valid until count_kmers ends) */
uint64_t ro_graph_bloom_insertions(const ro_graph *g, int rank); /* BloomFilter.cpp:228-230 */
uint64_t ro_graph_occurrences(const ro_graph *g);                /* k-mer occurrences streamed in pass 1 */
/* Vertex::dump records (Vertex.cpp:438-454): W*8 key bytes, u32 coverage, u8 edges; sorted by
 * (word0, word1).  Returns the number of records; writes at most cap_records. */
uint64_t ro_graph_export(const ro_graph *g, int rank, uint8_t *buf, uint64_t cap_records);
int ro_graph_find(const ro_graph *g, const uint64_t kmer[2], uint32_t *coverage, uint8_t *edges_in_kmer_orientation);
/* CoverageGatherer.cpp:124-190 + MessageProcessor.cpp:1048-1061, merged over ranks, ascending */
uint64_t ro_graph_histogram(const ro_graph *g, uint32_t *cov, uint64_t *cnt, uint64_t cap);
/* MachineHelper.cpp:548-578: 5x5 [parents][children] over both strands of every vertex */
void ro_graph_degree_distribution(const ro_graph *g, uint64_t out[25]);
/* CoverageGatherer.cpp:47-122: body lines of kmers.txt (no header), one per strand per vertex, in
 * sorted-lower-key order; returns bytes needed; writes at most cap bytes. */
uint64_t ro_graph_kmers_txt(const ro_graph *g, char *out, uint64_t cap);

/* ---- A13: CoverageDistribution (CoverageGatherer/CoverageDistribution.cpp:33-112) */
/* writes the file when path != NULL; fills minimum/peak/repeat.  An empty vote map is undefined
 * behaviour in the reference (votes.begin() of an empty map); here it selects position 0. */
int ro_coverage_distribution(const uint32_t *cov, const uint64_t *cnt, uint64_t n, const char *path,
                             int *minimum, int *peak, int *repeat);

/* ---- next rows: on-disk formats, read ingest, SequencesIndexer markers ---- */
#define RO_MAX_READ_LENGTH 65535 /* RAY_MAXIMUM_READ_LENGTH, Mock/constants.h:100 */
int ro_graph_write_genome_graph(const ro_graph *g, int rank, const char *path);      /* MessageProcessor.cpp:658-678, Vertex.cpp:257-273 */
int ro_write_coverage_checkpoint(const uint32_t *cov, const uint64_t *cnt, uint64_t n, const char *path); /* MachineHelper.cpp:243-259 */
int ro_graph_write_partition(const ro_graph *g, const char *path);                   /* EdgePurger.cpp:172-217 */
uint64_t ro_load_sequences(const char *path, int is_fastq, uint8_t *packed, uint64_t packed_cap, uint64_t *offsets,
                           uint32_t *lengths, uint64_t cap_reads, uint64_t *packed_bytes); /* FastqLoader.cpp:84-119, FastaLoader.cpp:26-64 */
int ro_graph_read_markers(const ro_graph *g, int32_t *forward, int32_t *reverse);    /* SequencesIndexer/IndexerWorker.cpp:67-248,328-362 */
int ro_write_coverage_analysis(const uint32_t *cov, const uint64_t *cnt, uint64_t n, int wordSize, int minimumCoverage, int peakCoverage,
                               int repeatCoverage, const char *distribution_file, const char *path); /* MachineHelper.cpp:283-327 */
int ro_graph_read_genome_graph(ro_graph *g, const char *path);                        /* MessageProcessor.cpp:596-655 */
/* the counter-based generator of synthetic reads (ro_synth.c; device twin: tools/rg_synth.hip) */
int ro_synth_generate(uint8_t *packed, uint64_t *offsets, uint32_t *lengths, uint64_t n, uint64_t r0, uint64_t genome, uint32_t L, uint64_t seed,
                      double error);

#ifdef __cplusplus
}
#endif
#endif
