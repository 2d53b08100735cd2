/*
 * raygraph.h -- C ABI of the MI355X (gfx950) de Bruijn graph construction path.
 *
 * This is the drop-in boundary for Ray's graph-construction plugins.  Ray keeps its three
 * CorePlugin classes, their slave-mode symbols and completion tags; only the handler bodies
 * change and call the functions below (INTEGRATION.md shows the glue).  Plain C: opaque handle,
 * int status codes, plain pointers and sizes, no C++/torch types.  One host thread per handle
 * (a Ray rank is a single-threaded event loop, application_core/Machine.cpp:560).
 *
 * Citations (file:line) are relative to /root/reference/code/ and name the reference interface
 * each entry point replaces.
 */
#ifndef RAYGRAPH_H
#define RAYGRAPH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes: 0 = ok, negative = error (glue maps non-zero to Ray's abort route,
 * MachineHelper/MachineHelper.cpp:261-268) */
#define RG_OK 0
#define RG_ERR_ARG (-1)      /* bad argument / configuration */
#define RG_ERR_HIP (-2)      /* a HIP runtime call failed; rg_last_error() has the text */
#define RG_ERR_NOMEM (-3)    /* device or host allocation failed */
#define RG_ERR_STATE (-4)    /* call made in the wrong phase */
#define RG_ERR_CAPACITY (-5) /* caller buffer too small / internal limit hit */
#define RG_ERR_HASH (-6)     /* rg_check_hashes: the library's k-mer hashes are not the caller's */
#define RG_ERR_COMM (-7)     /* a collective between ranks failed (RCCL, or the caller's transport); rg_last_error() has the text */

/* bloom_bits: size the filter like VerticesExtractor::getDefaultNumberOfBitsForBloomFilter
 * (VerticesExtractor/VerticesExtractor.cpp:276-320): 16*k*localReads capped at 2^33 */
#define RG_BLOOM_DEFAULT (~(uint64_t)0)

typedef struct rg_handle rg_handle;

/* What Parameters hands the plugins (Mock/Parameters.cpp: getWordSize, getSize, getRank;
 * MessageProcessor/MessageProcessor.cpp:890-893 for -bloom-filter-bits). */
typedef struct rg_config {
	int32_t k;              /* -k, odd; 1 <= k <= 32*kmer_words-1 (Ray itself enforces >= 15, Parameters.cpp:892-907) */
	int32_t kmer_words;     /* KMER_U64_ARRAY_SIZE: 1 (MAXKMERLENGTH=32) or 2 (=64); 0 = 1 if k<=31 else 2 (Kmer.h:46-61) */
	int32_t rank;           /* MPI rank == GPU owner index */
	int32_t size;           /* number of ranks */
	int32_t device;         /* HIP device ordinal */
	int32_t reserved0;      /* must be 0 */
	uint64_t bloom_bits;    /* RG_BLOOM_DEFAULT, 0 = Bloom filter off, else -bloom-filter-bits */
	uint64_t table_slots;   /* initial open-addressing capacity hint (like -hash-table-buckets); 0 = auto */
	uint64_t first_read_id; /* global id of this rank's first read (SequencesLoader.cpp:287-297) */
	void *stream;           /* hipStream_t to launch on; NULL = the library creates its own */
} rg_config;
/* Ordering: every entry point queues its work on the handle's stream and, unless its comment says otherwise, returns after
 * that stream has drained -- results are complete on return.  The other direction is the caller's: a device buffer handed
 * in (reads, received records / entries, queries, answers) must be complete before the call, i.e. whatever other stream
 * or collective wrote or cleared it must have finished (the library's own stream is non-blocking and waits for nobody). */

/* ---- life cycle ----------------------------------------------------------------------- */
/* replaces the constructors KmerAcademyBuilder.cpp:217-246, VerticesExtractor.cpp:322-341,
 * GridTable.cpp:34-66 */
int rg_create(const rg_config *cfg, rg_handle **out);
int rg_destroy(rg_handle *h);
/* text of the last error on this handle ("" if none); h may be NULL for rg_create failures */
const char *rg_last_error(const rg_handle *h);

/* ---- the hashes, checked against Ray's own ----------------------------------------------------------- */
/* Owners (vertexRank = hash_function_1 % size, Kmer.cpp:165-170) and Bloom bits (hash_function_2 ^ magic,
 * BloomFilter.cpp:124-200) rest on RayPlatform's uniform_hashing_function_1_64_64 / _2_64_64
 * (cryptography/crypto.cpp), a file that is absent from the reference tree this library was written against: the
 * kernels hold the published Thomas Wang mixes those functions are known as.  The rest of Ray computes owners with the
 * real ones, so the glue proves agreement once, right after rg_create, and aborts on RG_ERR_HASH:
 *   rg_check_hashes  hashes n_samples k-mers (fixed corners + a pseudo-random stream; 16 <= n <= 2^20) ON THE DEVICE,
 *                    with the code the kernels use, and compares with the caller's functions.  hash1 / hash2 get the
 *                    words of a k-mer as Kmer::getU64(0..kmer_words-1) returns them and return
 *                    Kmer::hash_function_1() / _2() (Kmer.cpp:284-310); `user` is passed through.
 *   rg_hash_kmers    the device's two hashes for the caller's own k-mers (host arrays; kmer_words words per k-mer),
 *                    for a glue that prefers to compare by itself. */
typedef uint64_t (*rg_hash_fn)(const uint64_t *kmer_words, void *user);
int rg_check_hashes(rg_handle *h, rg_hash_fn hash1, rg_hash_fn hash2, void *user, uint32_t n_samples);
int rg_hash_kmers(rg_handle *h, const uint64_t *kmers, uint64_t n, uint64_t *hash1_out, uint64_t *hash2_out);

/* ---- input: ArrayOfReads (SequencesLoader/Read.cpp:176-204 getRawSequence()/length()) ------- */
/* packed: the reads' 2-bit buffers back to back (base i of a read = bits 2(i%4).. of byte i/4,
 * Read.cpp:135-141); byte_offsets[i] = start of read i in `packed`; lengths[i] in bases
 * (<= 65535).  The library copies; the caller keeps ownership.  May be called repeatedly
 * (appends, read ids continue).  Host pointers. */
int rg_add_reads(rg_handle *h, const uint8_t *packed, uint64_t packed_bytes,
                 const uint64_t *byte_offsets, const uint32_t *lengths, uint64_t n_reads);
/* same, but the three arrays already live in this device's memory and are used in place
 * (they must stay valid until rg_destroy; `packed` must be 8-byte aligned and have 16 readable
 * bytes after packed_bytes). */
int rg_add_reads_device(rg_handle *h, const void *d_packed, uint64_t packed_bytes,
                        const void *d_byte_offsets, const void *d_lengths, uint64_t n_reads,
                        uint32_t max_length);

/* ---- pass 1: RAY_SLAVE_MODE_ADD_VERTICES ------------------------------------------------- */
/* size == 1: the whole of KmerAcademyBuilder::call_RAY_SLAVE_MODE_ADD_VERTICES
 * (KmerAcademyBuilder.cpp:43-211) + call_RAY_MPI_TAG_VERTICES_DATA (MessageProcessor.cpp:692-809)
 * + Bloom sizing (MessageProcessor.cpp:886-899).  Equivalent to rg_count_local +
 * rg_resolve_admission. */
int rg_count_kmers(rg_handle *h);

/* Record exchange (size > 1), the preferred multi-GPU route: what travels between ranks is the reads cut
 * into super-k-mer records (2.9 B per k-mer occurrence at k=31 instead of the 8 B key the reference sends per
 * occurrence, KmerAcademyBuilder.cpp:187-196), *before* aggregation, so that every rank aggregates complete
 * buckets at the full coverage whatever the number of ranks:
 *   rg_records_local    reads -> records, counted per destination rank (destination = bucket of the record's
 *                       minimizer % size; an internal choice, invisible in the results)
 *   rg_record_counts    counts[r] = records for rank r; *record_bytes = size of one record (20 bytes for k <= 31, 24 for k = 32, 40 above; opaque to the caller)
 *   rg_export_records   the records grouped by destination in rank order, into the caller's device buffer
 *   -- all-to-all(v) of the records --
 *   rg_count_records    aggregates the received records (device pointer; must stay valid during the call)
 * and then, exactly as after rg_count_local: rg_owner_counts / rg_export_owner_buckets (entries to the ranks
 * vertexRank names) -- all-to-all(v) -- rg_merge_entries, rg_resolve_admission.
 * rg_records_local_part(part, nparts) does the first step for reads [n*part/nparts, n*(part+1)/nparts) of every
 * batch only: counts + export + all-to-all per part, so that the transfer of one part overlaps the kernels of the
 * next (the reference overlaps the same way, message by message: KmerAcademyBuilder.cpp:187-196 sends while
 * it keeps reading); rg_count_records then takes all received records at once, in any order of parts. */
int rg_records_local(rg_handle *h);
int rg_records_local_part(rg_handle *h, uint32_t part, uint32_t nparts);
int rg_record_counts(rg_handle *h, uint64_t *counts, uint32_t *record_bytes);
int rg_export_records(rg_handle *h, void *d_out, uint64_t capacity_records);
int rg_count_records(rg_handle *h, const void *d_records, uint64_t n_records);

/* size >= 1, decomposed so that the host can run the owner exchange (an all-to-all) between:
 *  1. rg_count_local        sender half (KmerAcademyBuilder.cpp:143-205): encode, canonical pick,
 *                           aggregate this rank's k-mers per canonical key
 *  2. rg_owner_counts       entries per owner rank, owner = hash1(lower) % size (Kmer.cpp:165-170)
 *  3. rg_export_owner_buckets  32-byte entries grouped by owner into a device buffer
 *  4. (host) all-to-all of the buckets -- replaces RAY_MPI_TAG_VERTICES_DATA messages
 *  5. rg_merge_entries      receiver half: merge incoming entries into this rank's partition.  Hand over everything
 *                           the rank received in ONE call when you can: entries of the record route are distinct keys,
 *                           and a partition that arrives whole is kept as a log, admitted first and only its admitted
 *                           vertices inserted (as on a single rank); several calls merge into the table as before
 *  6. rg_resolve_admission  Bloom admission + coverage (MessageProcessor.cpp:741-804) */
int rg_count_local(rg_handle *h);
int rg_owner_counts(rg_handle *h, uint64_t *counts /* [size] */);
/* d_out: device buffer of sum(counts)*RG_ENTRY_BYTES bytes; bucket r starts at entry
 * sum(counts[0..r)) */
#define RG_ENTRY_BYTES 32
int rg_export_owner_buckets(rg_handle *h, void *d_out, uint64_t capacity_entries);
int rg_merge_entries(rg_handle *h, const void *d_entries, uint64_t n_entries);
/* Two calls that keep big partitions (tens of GB of entries per rank) from sitting in memory three times; same results,
 * no counterpart in the reference (its messages are 4 KB and its table grows in place, GridTable.cpp:111-134):
 *   rg_release_local   after rg_export_owner_buckets: frees this rank's aggregation before the exchange instead of at the
 *                      first rg_merge_entries (the buckets hold everything it knew)
 *   rg_adopt_entries   rg_merge_entries for a partition that arrives whole, without the copy: admission runs in the
 *                      caller's device buffer, which must stay valid and untouched until rg_resolve_admission has
 *                      returned (its contents are undefined afterwards).  Where rg_merge_entries would not keep the entries as a
 *                      log (a second delivery, entries of the entry route) it behaves exactly like rg_merge_entries. */
int rg_release_local(rg_handle *h);
int rg_adopt_entries(rg_handle *h, void *d_entries, uint64_t n_entries);
/* local_reads_for_bloom: this rank's read count for RG_BLOOM_DEFAULT sizing
 * (VerticesExtractor.cpp:298-300); pass UINT64_MAX to use the reads added to this handle */
int rg_resolve_admission(rg_handle *h, uint64_t local_reads_for_bloom);

/* ---- RAY_SLAVE_MODE_SEND_DISTRIBUTION ----------------------------------------------------- */
/* local part of CoverageGatherer::call_RAY_SLAVE_MODE_SEND_DISTRIBUTION
 * (CoverageGatherer/CoverageGatherer.cpp:124-190): (coverage, count) pairs ascending, count = 2
 * per stored pair (GridTableIterator.cpp:41-54).  *n = pairs available; fills min(*n, cap). */
int rg_histogram(rg_handle *h, uint32_t *coverage, uint64_t *count, uint64_t cap, uint64_t *n);

/* ---- pass 2: RAY_SLAVE_MODE_ADD_EDGES ------------------------------------------------------ */
/* VerticesExtractor::call_RAY_SLAVE_MODE_ADD_EDGES (VerticesExtractor.cpp:40-274) +
 * call_RAY_MPI_TAG_OUT_EDGES_DATA / IN_EDGES_DATA (MessageProcessor.cpp:858-968).  Edge bits
 * travel with the pass-1 entries (they are order-free ORs), so this call only publishes them. */
int rg_add_edges(rg_handle *h);

/* ---- next row: RAY_SLAVE_MODE_PURGE_NULL_EDGES (EdgePurger/EdgePurgerWorker.cpp:28-102) ------- */
/* size == 1: clears edge bits whose neighbour is not stored, in one call */
int rg_purge_edges(rg_handle *h);
/* size > 1: the same with one query exchange.  Replaces the per-edge RAY_MPI_TAG_REQUEST_VERTEX_COVERAGE
 * round trips of EdgePurgerWorker.cpp:36-60,73-96 (served by MessageProcessor.cpp:1224-1256):
 *   rg_purge_query_counts    counts[r] = neighbour keys this rank must ask owner r about
 *   rg_purge_export_queries  the keys, grouped by owner in rank order, kmer_words*8 bytes each (opaque)
 *   -- all-to-all(v) of the keys; every rank answers what it received: --
 *   rg_purge_answer          answers[i] = 1 if key i is stored here (coverage != 0), else 0
 *   -- all-to-all(v) of the answers back, same order as the exported keys --
 *   rg_purge_apply           deletes the edges whose neighbour's coverage is 0 (EdgePurgerWorker.cpp:54-56,91-93)
 * All pointers are device pointers except counts.  Works for size == 1 too (everything stays local). */
int rg_purge_query_counts(rg_handle *h, uint64_t *counts);
int rg_purge_export_queries(rg_handle *h, void *d_queries, uint64_t cap_queries);
int rg_purge_answer(rg_handle *h, const void *d_queries, uint64_t n, uint8_t *d_answers);
int rg_purge_apply(rg_handle *h, const uint8_t *d_answers, uint64_t n);

/* ---- multi-GPU exchange: one rank per GPU, the all-to-all(v) between them ------------------------------------------
 * Replaces RayPlatform's per-message MPI point-to-point traffic for this path -- RAY_MPI_TAG_VERTICES_DATA (sender
 * KmerAcademyBuilder.cpp:187-196, receiver MessageProcessor.cpp:692-809), OUT_/IN_EDGES_DATA (VerticesExtractor.cpp:175-251),
 * REQUEST_VERTEX_COVERAGE (EdgePurgerWorker.cpp:38-47, IndexerWorker.cpp:76-96, served by MessageProcessor.cpp:1224-1256),
 * COVERAGE_DATA (CoverageGatherer.cpp:164-181 -> MessageProcessor.cpp:1048-1061) -- by a few all-to-all(v) per pass.  The
 * rg_*_exchange entry points below drive the step-wise entry points above (counts, export, answer, apply ...) and move what
 * they export through a communicator:
 *   rg_comm_create_rccl       RCCL over xGMI (librccl.so.1 is opened at run time; nothing links against it).  Rank 0 calls
 *                             rg_comm_unique_id and sends the RG_COMM_ID_BYTES it returns to the other ranks -- one Ray
 *                             message, a pipe, a file --; then EVERY rank calls rg_comm_create_rccl(id, rank, size, device)
 *                             (a collective: it returns when all ranks have joined).  One rank per GPU.
 *   rg_comm_create_transport  the caller's own collectives, e.g. MPI_Alltoall / MPI_Alltoallv of the GPU-aware MPI Ray
 *                             already runs on: alltoall_u64 moves n words per rank between HOST arrays (send[r*n ..] goes to
 *                             rank r, recv[r*n ..] comes from it); alltoallv moves send_counts[r] items of unit_bytes to rank
 *                             r out of d_send (grouped in rank order) and recv_counts[r] items from rank r into d_recv, both
 *                             DEVICE memory, and returns when d_recv is complete.  Both return 0 on success.
 * A communicator belongs to one rank of one job, outlives the handles it is used with (create it once per Ray run) and may
 * be NULL when size == 1 (everything then stays on the device).  One host thread per communicator.  Errors of a
 * collective are RG_ERR_COMM with the text in rg_last_error(h). */
typedef struct rg_comm rg_comm;
#define RG_COMM_ID_BYTES 256
typedef struct rg_transport {
	void *user;
	int (*alltoall_u64)(void *user, const uint64_t *send, uint64_t *recv, uint32_t n_per_rank);
	int (*alltoallv)(void *user, const void *d_send, const uint64_t *send_counts, void *d_recv, const uint64_t *recv_counts,
	                 uint32_t unit_bytes);
} rg_transport;
int rg_comm_unique_id(void *id_out /* RG_COMM_ID_BYTES */);
int rg_comm_create_rccl(const void *id, int rank, int size, int device, rg_comm **out);
int rg_comm_create_transport(const rg_transport *t, int rank, int size, int device, rg_comm **out);
int rg_comm_destroy(rg_comm *c);
/* text of the communicator's last error; c may be NULL for failures of rg_comm_unique_id / rg_comm_create_* */
const char *rg_comm_last_error(const rg_comm *c);
/* payload bytes this rank sent to / received from OTHER ranks, payload collectives issued, seconds the host waited for them */
typedef struct rg_comm_stats {
	uint64_t bytes_sent, bytes_received, collectives;
	double wait_seconds;
} rg_comm_stats;
int rg_comm_get_stats(rg_comm *c, rg_comm_stats *out);

/* RAY_SLAVE_MODE_ADD_VERTICES across ranks (KmerAcademyBuilder.cpp:43-211 + MessageProcessor.cpp:692-809):
 *   rg_exchange_records        record route, first exchange: rg_records_local_part / rg_record_counts / rg_export_records per
 *                              slice of the reads, all-to-all(v) of the records (the transfer of slice p runs while the
 *                              kernels of slice p + 1 produce theirs; parts = 0 -> 1), then rg_count_records on what arrived
 *   rg_exchange_owner_buckets  both routes: rg_owner_counts / rg_export_owner_buckets / rg_release_local, all-to-all(v) of the
 *                              32-byte entries to the ranks vertexRank names, ONE rg_adopt_entries of everything received
 *                              (admitted in the receive buffer, which the library owns).  Call rg_resolve_admission next.
 *   rg_count_kmers_exchange    the whole pass: RG_ROUTE_RECORDS = rg_exchange_records, RG_ROUTE_ENTRIES = rg_count_local; then
 *                              rg_exchange_owner_buckets and rg_resolve_admission(local_reads_for_bloom).  Which route:
 *                              entries when one rank's reads cover a k-mer ~16 times or more (2 ranks), records otherwise.
 * RAY_SLAVE_MODE_SEND_DISTRIBUTION: rg_histogram_exchange = rg_histogram of every rank merged as rank 0 merges
 *   RAY_MPI_TAG_COVERAGE_DATA (MessageProcessor.cpp:1048-1061), on every rank (same calling convention as rg_histogram).
 * RAY_SLAVE_MODE_PURGE_NULL_EDGES: rg_purge_edges_exchange = rg_purge_query_counts .. rg_purge_apply with the two all-to-all(v).
 * SequencesIndexer: rg_read_markers_exchange = rg_markers_query_counts_part .. rg_markers_apply per slice of the reads
 *   (parts = 0 -> 1) with the two all-to-all(v); d_forward / d_reverse as rg_read_markers.
 * rg_allgather_u64: one word from every rank into out[size] (GridTable sizes for GraphPartition.txt, EdgePurger.cpp:172-217). */
#define RG_ROUTE_RECORDS 1
#define RG_ROUTE_ENTRIES 2
int rg_exchange_records(rg_handle *h, rg_comm *c, uint32_t parts);
int rg_exchange_owner_buckets(rg_handle *h, rg_comm *c);
int rg_count_kmers_exchange(rg_handle *h, rg_comm *c, int route, uint32_t parts, uint64_t local_reads_for_bloom);
int rg_histogram_exchange(rg_handle *h, rg_comm *c, uint32_t *coverage, uint64_t *count, uint64_t cap, uint64_t *n);
int rg_purge_edges_exchange(rg_handle *h, rg_comm *c);
int rg_read_markers_exchange(rg_handle *h, rg_comm *c, uint32_t parts, int32_t *d_forward, int32_t *d_reverse);
int rg_allgather_u64(rg_handle *h, rg_comm *c, uint64_t value, uint64_t *out);

/* ---- the store: GridTable / Vertex ---------------------------------------------------------- */
/* GridTable::size (GridTable.cpp:80-82): 2 per stored pair */
int rg_size(rg_handle *h, uint64_t *size);
/* Vertex::dump records (VerticesExtractor/Vertex.cpp:438-454): kmer_words*8 bytes lower key,
 * 4 bytes coverage, 1 byte edge bitmap = 13 or 21 bytes each, in table order.  host_buf holds
 * cap_records records; *n_records = records available. */
int rg_export(rg_handle *h, void *host_buf, uint64_t cap_records, uint64_t *n_records);
/* The same hand-back as a stream, for the glue's insert loop (GridTable::insert + Vertex::constructor + Vertex::load per
 * record: GridTable.cpp:111-134, Vertex.cpp:41-51,422-436): `consume` is called on the calling thread with chunk after chunk
 * of records (same layout and order as rg_export; valid during the call only) in page-locked memory of the library's, and
 * while it works through one chunk the copy of the next is running: wall time = max(copy, consumer), not their sum, and no
 * host buffer of the whole store is needed.  chunk_records = 0: 4 M records per chunk.  A non-zero return of `consume`
 * stops the export (RG_ERR_STATE). */
typedef int (*rg_export_fn)(void *user, const void *records, uint64_t n_records);
int rg_export_stream(rg_handle *h, uint64_t chunk_records, rg_export_fn consume, void *user);
/* bulk GridTable::find (GridTable.cpp:84-109) + Vertex::getCoverage/getEdges (Vertex.cpp:98-100,
 * 169-173): kmers = n * kmer_words words, either strand; coverage 0 = absent; edges in the
 * orientation of the queried k-mer.  Host pointers. */
int rg_find(rg_handle *h, const uint64_t *kmers, uint64_t n, uint32_t *coverage, uint8_t *edges);
/* ---- next rows ----------------------------------------------------------------- */
/* Read ingest.  rg_add_reads_text: n raw sequences (any bytes) as they stand in a FASTA/FASTQ file,
 * sequence i = text[starts[i], starts[i]+raw_lengths[i]); the device applies Read::constructor with
 * Read::trim (SequencesLoader/Read.cpp:31-159: acgt upper-cased, leading/trailing non-ACGT dropped,
 * inner ones become A) and packs 2 bits per base.  Host pointers; raw_lengths[i] < 65535.
 * rg_count_sequences / rg_load_sequences: FastaLoader::load (FastaLoader.cpp:26-64) and
 * FastqLoader::loadWithPeriod (FastqLoader.cpp:84-119) for plain .fasta/.fa/.fna/.fas and .fastq/.fq
 * files; sequences [first, last) of the file, as SequencesLoader.cpp:287-297 partitions them.
 * rg_get_reads copies the packed reads back (rg_add_reads layout): *n_reads / *packed_bytes are what
 * exists; filled only when everything fits the capacities. */
int rg_add_reads_text(rg_handle *h, const char *text, uint64_t text_bytes, const uint64_t *starts,
                      const uint32_t *raw_lengths, uint64_t n_reads);
int rg_count_sequences(const char *path, uint64_t *n);
int rg_load_sequences(rg_handle *h, const char *path, uint64_t first, uint64_t last, uint64_t *loaded);
int rg_get_reads(rg_handle *h, uint8_t *packed, uint64_t packed_cap, uint64_t *byte_offsets, uint32_t *lengths,
                 uint64_t cap_reads, uint64_t *n_reads, uint64_t *packed_bytes);
/* Read ingest that allocates nothing.  No counterpart in the reference: it is for callers that cannot allocate
 * device memory -- code under HIP graph capture, tight serving loops that ingest one batch of reads after another,
 * harnesses that forbid hipMalloc.  Same trimming and packing as rg_add_reads_text (same kernels), but every array
 * is the caller's and lives in device memory: sequence i = d_text[d_starts[i], d_starts[i] + d_raw_lengths[i])
 * (d_raw_lengths[i] < 65535); out: d_lengths[i] = stored length, d_byte_offsets[i] = start of read i in d_packed,
 * d_byte_offsets[n_reads] = packed bytes produced (the array has n_reads + 1 entries).  Bytes beyond packed_cap are
 * not written: d_byte_offsets[n_reads] > packed_cap tells the caller that d_packed is incomplete.  The outputs have
 * the layout rg_add_reads_device takes.  The per-read scratch (16 B per sequence) is a fixed array inside the
 * library, RG_NOALLOC_MAX_READS sequences per call (more: RG_ERR_CAPACITY); one array per GPU, shared by every
 * call on that GPU, so calls that overlap on one GPU (two streams, two threads) race.  The work is queued on the
 * handle's stream and not waited for; no hipMalloc, hipFree or other allocating call is made. */
#define RG_NOALLOC_MAX_READS 262144
int rg_pack_reads_noalloc(rg_handle *h, const char *d_text, uint64_t text_bytes, const uint64_t *d_starts,
                          const uint32_t *d_raw_lengths, uint64_t n_reads, uint8_t *d_packed, uint64_t packed_cap,
                          uint64_t *d_byte_offsets, uint32_t *d_lengths);
/* SequencesIndexer marker selection (SequencesIndexer/IndexerWorker.cpp:67-248,328-362; the build without
 * CONFIG_USE_COVERAGE_DISTRIBUTION): for every read, in the order added, the forward marker = first
 * k-mer position whose coverage reaches the read's threshold, the reverse marker = length-k-(last such
 * position); -1 = none (the reference then leaves the Read's offsets at 0 and attaches nothing).
 * Replaces one RAY_MPI_TAG_REQUEST_VERTEX_COVERAGE round trip per k-mer (:76-96).  Device pointers,
 * one int32 per read.  rg_read_markers: size == 1, one call.  size > 1 (works for size == 1 too), with one key exchange,
 * like the purge:
 *   rg_markers_query_counts    counts[r] = k-mers of this rank's reads whose owner is rank r (vertexRank, Kmer.cpp:165-170)
 *   rg_markers_export_queries  the canonical keys grouped by owner in rank order, kmer_words*8 bytes each (opaque)
 *   -- all-to-all(v) of the keys; every rank answers what it received: --
 *   rg_markers_answer          coverage[i] = coverage of key i here, 0 if it is not stored (MessageProcessor.cpp:1224-1256)
 *   -- all-to-all(v) of the answers (4 bytes each) back, same order as the exported keys --
 *   rg_markers_apply           the marker rule of IndexerWorker.cpp:97-248 on the returned coverages
 * rg_markers_query_counts_part(part, nparts) does this for reads [n*part/nparts, n*(part+1)/nparts) of every batch only (the
 * slices of rg_records_local_part), so that the buffers of a round are bounded: per k-mer of the slice the caller holds a key
 * and an answer, the library one 8-byte index; export / apply work on the slice the last counts call named, and apply writes
 * the markers of that slice's reads only (d_forward / d_reverse are always the arrays of ALL reads of the handle).  Memory is
 * per k-mer that exists: one very long read among short ones costs what its own k-mers cost. */
int rg_read_markers(rg_handle *h, int32_t *d_forward, int32_t *d_reverse);
int rg_markers_query_counts(rg_handle *h, uint64_t *counts);
int rg_markers_query_counts_part(rg_handle *h, uint32_t part, uint32_t nparts, uint64_t *counts);
int rg_markers_export_queries(rg_handle *h, void *d_queries, uint64_t cap_queries);
int rg_markers_answer(rg_handle *h, const void *d_queries, uint64_t n, uint32_t *d_coverage);
int rg_markers_apply(rg_handle *h, const uint32_t *d_coverage, uint64_t n, int32_t *d_forward, int32_t *d_reverse);
/* On-disk formats.  records = rg_export output (Vertex::dump layout).
 * rg_write_genome_graph: Rank<r>.Checkpoint.GenomeGraph.ray (MessageProcessor.cpp:658-678, read back by
 * :596-655; record = Vertex::write, Vertex.cpp:257-273), vertices in the order of `records`.
 * rg_write_coverage_checkpoint: Rank0.Checkpoint.CoverageDistribution.ray (MachineHelper.cpp:243-259).
 * rg_write_graph_partition: <prefix>GraphPartition.txt (EdgePurger/EdgePurger.cpp:172-217), sizes[r] = rg_size of rank r. */
/* rg_write_kmers_txt: kmers.txt (CoverageGatherer/CoverageGatherer.cpp:47-122; header :202-228, written by the
 * master rank only); both strands of every record, in the order of `records`. */
int rg_write_kmers_txt(const void *records, uint64_t n_pairs, int k, int kmer_words, const char *path, int write_header,
                       const char *ray_version, int append);
/* rg_import_vertices: the reverse of rg_export on a fresh handle without reads -- what reading the GenomeGraph
 * checkpoint does to the GridTable (MessageProcessor.cpp:596-655, Vertex::load Vertex.cpp:422-436): n_pairs records in
 * Vertex::dump layout become the vertex store, coverage and edges final (rg_purge_*, rg_find, rg_export, ... work). */
int rg_import_vertices(rg_handle *h, const void *records, uint64_t n_pairs);
int rg_write_genome_graph(const void *records, uint64_t n_pairs, int k, int kmer_words, const char *path);
int rg_write_coverage_checkpoint(const uint32_t *coverage, const uint64_t *count, uint64_t n, const char *path);
int rg_write_graph_partition(const uint64_t *sizes, int nranks, const char *path);
/* rg_write_coverage_analysis: <prefix>CoverageDistributionAnalysis.txt as call_RAY_MASTER_MODE_SEND_COVERAGE_VALUES writes it
 * (MachineHelper/MachineHelper.cpp:283-327) from the merged histogram and the three values of rg_coverage_distribution;
 * distribution_file = the path of CoverageDistribution.txt as it is printed on the last line (:325).
 * rg_read_genome_graph: the reader of Rank<r>.Checkpoint.GenomeGraph.ray (MessageProcessor.cpp:596-655): entries of either
 * strand with their parents and children as whole k-mers -> Vertex::dump records, one per pair (coverage from the lower
 * key's entry, edge bits from both); the reader's assertions on the parents / children counts and the size (:633-641,650)
 * are RG_ERR_ARG here.  records == NULL: *n_pairs only (the header's count / 2).  Feed the records to rg_import_vertices. */
int rg_write_coverage_analysis(const uint32_t *coverage, const uint64_t *count, uint64_t n, int k, int32_t minimum, int32_t peak,
                               int32_t repeat, const char *distribution_file, const char *path);
int rg_read_genome_graph(const char *path, int k, int kmer_words, void *records, uint64_t cap_pairs, uint64_t *n_pairs);

/* call_RAY_SLAVE_MODE_WRITE_KMERS degree table (MachineHelper/MachineHelper.cpp:548-578):
 * out[parents*5+children] over both strands of every stored vertex */
int rg_degree_distribution(rg_handle *h, uint64_t out[25]);

/* ---- statistics (the "Rank r ..." log lines, MessageProcessor.cpp:984-1037) ------------------ */
typedef struct rg_stats {
	uint64_t reads;            /* reads added */
	uint64_t occurrences;      /* k-mer occurrences streamed in pass 1 */
	uint64_t candidates;       /* distinct canonical k-mers seen by this rank's partition */
	uint64_t stored_pairs;     /* retained after admission */
	uint64_t bloom_bits;       /* m_bloomBits */
	uint64_t bloom_insertions; /* BloomFilter::getNumberOfInsertions */
	uint64_t bloom_set_bits;   /* BloomFilter::getNumberOfSetBits */
	uint64_t table_slots;      /* current open-addressing capacity */
	uint64_t sweep_partitioned; /* 1 if pass 1 ran as the partitioned sweep, 0 if as the direct sweep */
	uint64_t superkmer_records; /* partitioned sweep: super-k-mer records written */
	uint64_t partition_levels;  /* partitioned sweep: scatter levels run (1..3) */
	uint64_t log_entries;       /* partitioned sweep: aggregated entries handed to the table merge */
	uint64_t bloom_partitioned; /* admission resolved by: 0 dense first-touch stamps, 1 Bloom-region partitioning of (bit, key) items, 2 key partitions (filters of 2^B bits) */
	uint64_t host_syncs;        /* times the host has waited for the handle's stream so far (sizes read back for the next allocation or grid) */
	uint64_t folded_level_bits; /* partitioned sweep: bits of the first partition level done by the record kernel itself (0 = none); partition_levels counts it */
	uint64_t items_from_sweep;  /* 1: the Bloom admission started from items the aggregation made while it wrote the log (no pass over the log to make them) */
	uint64_t compact_log;       /* one-word keys, items made by the aggregation: 1 = the log was written in 16-byte entries (key, count and edge bits; the item carries the
	                             * arrival stamp) and read as such by the one-pass admission, 2 = written so and expanded to full entries for another route, 0 = full entries */
} rg_stats;
int rg_get_stats(rg_handle *h, rg_stats *out);

/* ---- memory ------------------------------------------------------------------------------ */
/* The library keeps freed device blocks (tables, Bloom stamps) in a per-device pool and reuses them
 * for later handles instead of returning them to the driver (RayPlatform's MyAllocator /
 * __Malloc bookkeeping plays the same role on the host, KmerAcademyBuilder/BloomFilter.cpp:96).
 * This call gives every cached block back. */
int rg_release_cached_memory(void);

/* ---- test and experiment switches -------------------------------------------------------------- */
/* Forces internal routes that the library otherwise chooses by itself; results never depend on them.  No counterpart
 * in the reference; a plugin leaves it alone.  Must be called before the counting starts.  flags:
 *   bits 0-7   sweep: 0 = automatic, 1 = force the partitioned sweep, 2 = force the direct sweep
 *   bit  2     the aggregation never makes the Bloom admission's items on the way (one rank, filters of 2^28 bits or more)
 *   bit  3     fold the first partition level into the record kernel on small inputs too (k = 31)
 *   bit  4     force the Bloom resolution by (bit, key) items per Bloom region
 *   bit  5     force the dense first-touch stamps
 *   bit  6     start with a one-chunk log (tests of the overflow path)
 *   bit  7     never fold the first partition level into the record kernel
 *   bits 8-15  log2 of the bucket count of the partitioned sweep (0 = automatic)
 *   bit  16    cut super-k-mer records with the wavefront-per-read kernel even where the lane-per-read one applies
 *   bit  17    produce the Bloom items straight into their first partition level even from a sparse key array
 *   bit  18    size the log from a sample of the buckets even on small inputs
 *   bit  19    rg_merge_entries never keeps received entries as a log
 *   bit  20    the partition levels' histograms read the records instead of the digit bytes the previous level left
 *   bit  21    aggregate with a workgroup per bucket of a two-level partition (k_aggregate_wg) instead of a wavefront per
 *              bucket of a three-level one
 *   bit  22/23 always / never build the vertex table range by range from partitioned entries
 *   bit  24    resolve a power-of-two Bloom filter by key partitions whatever its fill
 *   bit  25    pick the bucket count from a sample of the records even on small inputs
 *   bit  26    aggregate with the 512-slot k-mer tables that diverse inputs get
 *   bit  27    keep super-k-mer records in their 24-byte form where the 20-byte one applies (k <= 31), and log entries in their
 *              32-byte form where the 16-byte one applies (one-word keys, items made by the aggregation)
 *   bit  28    key partitions: size the first level's ranges from a histogram pass instead of in advance
 *   bit  29    key partitions: the dense region kernel even on a sparse filter
 *   bit  30    cut the records of k = 31 with the generic lane-per-read kernel instead of the one unrolled for its window
 *   bit  31    count the record sample's distinct k-mers with a nested sweep instead of one set-insertion launch */
int rg_debug_set(rg_handle *h, uint32_t flags);

/* ---- profiling (the reference's MACRO_COLLECT_PROFILING_INFORMATION probes,
 * KmerAcademyBuilder.cpp:50,73,116,...) ------------------------------------------------------- */
/* when enabled, the launches of every hot kernel are bracketed by HIP events on the launch stream */
int rg_set_profiling(rg_handle *h, int enable);
/* one text line per kernel: "<name> <total milliseconds> <launches>\n" (synchronises the stream) */
int rg_get_kernel_profile(rg_handle *h, char *buf, uint64_t cap);

/* ---- host-side master-mode helpers (no GPU needed) ------------------------------------------- */
/* CoverageDistribution::CoverageDistribution (CoverageGatherer/CoverageDistribution.cpp:33-112):
 * writes CoverageDistribution.txt to `path` (NULL = no file) and returns minimum / peak / repeat
 * coverage.  n == 0 -> RG_ERR_ARG ("no k-mers found", MachineHelper.cpp:261-268). */
int rg_coverage_distribution(const uint32_t *coverage, const uint64_t *count, uint64_t n,
                             const char *path, int32_t *minimum, int32_t *peak, int32_t *repeat);
/* Read::constructor packing incl. trimming (SequencesLoader/Read.cpp:31-159): returns the stored
 * length; packed needs (strlen(sequence)+3)/4 bytes */
int rg_pack_read(const char *sequence, int trim, uint8_t *packed);

/* library version string */
const char *rg_version(void);

#ifdef __cplusplus
}
#endif
#endif
