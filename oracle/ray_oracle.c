/*
 * ray_oracle.c -- CPU restatement of sebhtml/Ray's graph-construction path (plain C99).
 *
 * TEST INFRASTRUCTURE ONLY -- see ray_oracle.h for the rules and the parity status
 * (hash mixes: PARITY UNPINNED; everything else pinned by the reference's in-comment
 * known answers and SURVEY.md section 8c).
 *
 * The code follows the reference's handlers line by line (same O(k) loops, same number of
 * reverse-complement calls, 500-word message buffers) so that it can also serve as the timed
 * CPU baseline ("port").  It is sequential: R ranks are simulated with the linearisation
 * "source rank 0's whole stream, then rank 1's, ..." (hard parts), which
 * for R=1 is exactly what the reference's stop-and-wait protocol produces.
 *
 * Citations are relative to /root/reference/code/.
 */
#include "ray_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define RO_MESSAGE_WORDS 500 /* NetworkTest/NetworkTest.h:98-99: 4000 bytes / sizeof(uint64_t) */
#define RO_MAX_READ 65535    /* Mock/constants.h:100 RAY_MAXIMUM_READ_LENGTH */

/* ------------------------------------------------------------------ codes */

/* Mock/constants.h:79-82: A=0 C=1 G=2 T=3; common_functions.cpp:151-164: anything else is A */
static uint8_t char_to_code(char a) {
	switch (a) {
	case 'A': return 0;
	case 'T': return 3;
	case 'C': return 1;
	case 'G': return 2;
	default: return 0;
	}
}

/* Kmer.cpp:338-364 (nucleotide mode) */
static char code_to_char(uint8_t a) {
	switch (a) {
	case 0: return 'A';
	case 3: return 'T';
	case 1: return 'C';
	case 2: return 'G';
	}
	return 'A';
}

/* ------------------------------------------------------------------ A0 Read */

int ro_read_required_bytes(int length) {
	int requiredBits = 2 * length;
	int modulo = requiredBits % 8;
	if (modulo != 0) requiredBits += 8 - modulo;
	return requiredBits / 8;
}

/* Read.cpp:31-68 */
static char *read_trim(char *buffer, const char *sequence) {
	int theLen = (int)strlen(sequence);
	strcpy(buffer, sequence);
	for (int i = 0; i < theLen; i++) {
		if (buffer[i] == 'a') buffer[i] = 'A';
		else if (buffer[i] == 't') buffer[i] = 'T';
		else if (buffer[i] == 'c') buffer[i] = 'C';
		else if (buffer[i] == 'g') buffer[i] = 'G';
	}
	int first = 0;
	while (buffer[first] != 'A' && buffer[first] != 'T' && buffer[first] != 'C' && buffer[first] != 'G' && first < theLen)
		first++;
	char *corrected = buffer + first;
	int last = 0;
	int len = (int)strlen(corrected);
	for (int i = 0; i < len; i++) {
		if (corrected[i] == 'A' || corrected[i] == 'T' || corrected[i] == 'C' || corrected[i] == 'G') last = i;
	}
	last++;
	corrected[last] = '\0';
	return corrected;
}

/* Read.cpp:76-159 */
int ro_read_pack(const char *sequence, int trim_flag, uint8_t *packed) {
	static char buffer[RO_MAX_READ + 2];
	if (trim_flag && strlen(sequence) < RO_MAX_READ) sequence = read_trim(buffer, sequence);
	int length = (int)strlen(sequence);
	int requiredBytes = ro_read_required_bytes(length);
	for (int i = 0; i < requiredBytes; i++) packed[i] = 0;
	for (int position = 0; position < length; position++) {
		char nucleotide = sequence[position];
		if (nucleotide != 'A' && nucleotide != 'T' && nucleotide != 'C' && nucleotide != 'G') nucleotide = 'A';
		uint8_t code = char_to_code(nucleotide);
		int positionInWorkingBuffer = position / 4;
		int codePositionInWord = position % 4;
		code = (uint8_t)(code << (codePositionInWord * 2));
		packed[positionInWorkingBuffer] |= code;
	}
	return length;
}

/* Read.cpp:161-174 */
void ro_read_get_seq(const uint8_t *packed, int length, char *out) {
	for (int position = 0; position < length; position++) {
		uint8_t word = packed[position / 4];
		int codePositionInWord = position % 4;
		uint8_t code = (uint8_t)(word << (6 - codePositionInWord * 2));
		code = (uint8_t)(code >> 6);
		out[position] = code_to_char(code);
	}
	out[length] = '\0';
}

/* ------------------------------------------------------------------ A1..A4 Kmer */

int ro_is_valid_dna(const char *x) {
	int len = (int)strlen(x);
	for (int i = 0; i < len; i++) {
		char a = x[i];
		if (!(a == 'A' || a == 'T' || a == 'C' || a == 'G')) return 0;
	}
	return 1;
}

void ro_word_id(const char *a, uint64_t out[2]) {
	out[0] = out[1] = 0;
	int theLen = (int)strlen(a);
	for (int j = 0; j < theLen; j++) {
		uint64_t k = char_to_code(a[j]);
		int bitPosition = 2 * j;
		int chunk = bitPosition / 64;
		int bitPositionInChunk = bitPosition % 64;
		out[chunk] |= (k << bitPositionInChunk);
	}
}

void ro_complement_vertex(const uint64_t in[2], int wordSize, uint64_t out[2]) {
	uint64_t o[2] = {0, 0};
	int bitPositionInOutput = 0;
	uint64_t mask = 3;
	for (int positionInMer = wordSize - 1; positionInMer >= 0; positionInMer--) {
		int u64_id = positionInMer / 32;
		int bitPositionInChunk = (2 * positionInMer) % 64;
		uint64_t chunk = in[u64_id];
		uint64_t j = (chunk << (62 - bitPositionInChunk)) >> 62;
		j = ~j & mask;
		int outputChunk = bitPositionInOutput / 64;
		o[outputChunk] |= (j << (bitPositionInOutput % 64));
		bitPositionInOutput += 2;
	}
	out[0] = o[0];
	out[1] = o[1];
}

int ro_less(const uint64_t a[2], const uint64_t b[2], int W) {
	for (int i = 0; i < W; i++) {
		if (a[i] < b[i]) return 1;
		else if (a[i] > b[i]) return 0;
	}
	return 0;
}

static int kmer_equal(const uint64_t a[2], const uint64_t b[2], int W) {
	for (int i = 0; i < W; i++)
		if (a[i] != b[i]) return 0;
	return 1;
}

/*
 * ASSUMED (parity unpinned): RayPlatform/cryptography/crypto.cpp is absent.  These are the
 * published Thomas Wang mixes ("hash64shift" and the 64-bit body of "hash6432shift") that
 * SURVEY.md section 8c recalls for uniform_hashing_function_1_64_64 / _2_64_64.
 */
uint64_t ro_mix1(uint64_t key) {
	key = (~key) + (key << 21);
	key = key ^ (key >> 24);
	key = (key + (key << 3)) + (key << 8);
	key = key ^ (key >> 14);
	key = (key + (key << 2)) + (key << 4);
	key = key ^ (key >> 28);
	key = key + (key << 31);
	return key;
}

uint64_t ro_mix2(uint64_t key) {
	key = (~key) + (key << 18);
	key = key ^ (key >> 31);
	key = key * 21;
	key = key ^ (key >> 11);
	key = key + (key << 6);
	key = key ^ (key >> 22);
	return key;
}

uint64_t ro_hash1(const uint64_t a[2], int W) {
	if (W == 1) return ro_mix1(a[0]);
	uint64_t key = a[0];
	for (int i = 0; i < W; i++) key ^= ro_mix1(a[i]);
	return key;
}

uint64_t ro_hash2(const uint64_t a[2], int W) {
	if (W == 1) return ro_mix2(a[0]);
	uint64_t key = a[0];
	for (int i = 0; i < W; i++) key ^= ro_mix2(a[i]);
	return key;
}

int ro_vertex_rank(const uint64_t a[2], int k, int W, int size) {
	uint64_t b[2];
	ro_complement_vertex(a, k, b);
	if (ro_less(a, b, W)) {
		b[0] = a[0];
		b[1] = a[1];
	}
	return (int)(ro_hash1(b, W) % (uint64_t)size);
}

void ro_id_to_word(const uint64_t a[2], int kmerLength, char *buffer) {
	for (int p = 0; p < kmerLength; p++) {
		int bitPosition = 2 * p;
		int chunkId = p / 32;
		int bitPositionInChunk = bitPosition % 64;
		uint64_t chunk = a[chunkId];
		uint64_t j = (chunk << (62 - bitPositionInChunk)) >> 62;
		buffer[p] = code_to_char((uint8_t)j);
	}
	buffer[kmerLength] = '\0';
}

/* Kmer.cpp:145-153 */
static uint8_t second_segment_last_code(const uint64_t a[2], int w) {
	int bitPosition = 2 * w;
	int chunkId = bitPosition / 64;
	int bitPositionInChunk = bitPosition % 64;
	uint64_t chunk = a[chunkId];
	chunk = (chunk << (64 - bitPositionInChunk)) >> 62;
	return (uint8_t)chunk;
}

/* Kmer.cpp:155-163 */
static uint8_t first_segment_first_code(const uint64_t a[2]) {
	uint64_t v = a[0];
	v = v << 62;
	v = v >> 62;
	return (uint8_t)v;
}

int ro_outgoing_edges(const uint64_t a[2], uint8_t edges, int k, int W, uint64_t out[4][2]) {
	uint64_t t[2] = {a[0], a[1]};
	for (int i = 0; i < W; i++) {
		uint64_t word = t[i] >> 2;
		if (i != W - 1) {
			uint64_t next = t[i + 1];
			next = (next << 62);
			word = word | next;
		}
		t[i] = word;
	}
	int positionToUpdate = 2 * k;
	int chunkIdToUpdate = positionToUpdate / 64;
	positionToUpdate = positionToUpdate % 64;
	int n = 0;
	for (int i = 0; i < 4; i++) {
		int j = (int)((((uint64_t)edges) << (64 - 5 - i)) >> 63);
		if (j == 1) {
			uint64_t nk[2] = {t[0], t[1]};
			uint64_t filter = (uint64_t)i;
			filter = filter << (positionToUpdate - 2);
			nk[chunkIdToUpdate] |= filter;
			out[n][0] = nk[0];
			out[n][1] = nk[1];
			n++;
		}
	}
	return n;
}

int ro_ingoing_edges(const uint64_t a[2], uint8_t edges, int k, int W, uint64_t out[4][2]) {
	uint64_t t[2] = {a[0], a[1]};
	int posToClear = 2 * k;
	for (int i = 0; i < W; i++) {
		uint64_t element = a[i];
		element = element << 2;
		if (i != 0) {
			uint64_t last = a[i - 1];
			last = (last >> 62);
			element = element | last;
		}
		if (i * 64 <= posToClear && posToClear < i * 64 + 64) {
			int position = posToClear % 64;
			uint64_t filter = 3;
			filter = filter << position;
			filter = ~filter;
			element = element & filter;
		}
		t[i] = element;
	}
	int n = 0;
	for (int i = 0; i < 4; i++) {
		int j = (int)((((uint64_t)edges) << (63 - i)) >> 63);
		if (j == 1) {
			out[n][0] = t[0] | (uint64_t)i;
			out[n][1] = t[1];
			n++;
		}
	}
	return n;
}

/* ------------------------------------------------------------------ A6 Bloom */

static const uint64_t RO_BLOOM_MAGIC[8] = {/* BloomFilter.cpp:72-79 */
	0xe70b369c4c19f0f9ULL, 0xbeb7f38b993441a2ULL, 0x3149cd9246ca7995ULL, 0x2ef5b2c17d479ee8ULL,
	0xfb8daceab90fe233ULL, 0x20fa74e37d497859ULL, 0x9007d0caef749698ULL, 0x4e4100a5605ef967ULL};

typedef struct {
	uint64_t bits;
	uint64_t *bitmap;
	uint64_t set_bits;
	uint64_t insertions;
} ro_bloom;

static void bloom_constructor(ro_bloom *b, uint64_t numberOfBits) {
	b->set_bits = 0;
	b->insertions = 0;
	b->bits = numberOfBits;
	uint64_t requiredBytes = numberOfBits / 8;
	uint64_t required8Bytes = requiredBytes / 8;
	if (numberOfBits % 64 != 0) required8Bytes++;
	b->bitmap = (uint64_t *)calloc(required8Bytes ? required8Bytes : 1, sizeof(uint64_t));
}

static int bloom_has_value(const ro_bloom *b, const uint64_t lower[2], int W) {
	uint64_t origin = ro_hash2(lower, W);
	for (int i = 0; i < 8; i++) {
		uint64_t hashValue = origin ^ RO_BLOOM_MAGIC[i];
		uint64_t bit = hashValue % b->bits;
		uint64_t chunk = bit / 64;
		int bitInChunk = (int)(bit % 64);
		int bitValue = (int)((b->bitmap[chunk] << (63 - bitInChunk)) >> 63);
		if (bitValue == 0) return 0;
	}
	return 1;
}

static void bloom_insert_value(ro_bloom *b, const uint64_t lower[2], int W) {
	uint64_t origin = ro_hash2(lower, W);
	for (int i = 0; i < 8; i++) {
		uint64_t hashValue = origin ^ RO_BLOOM_MAGIC[i];
		uint64_t bit = hashValue % b->bits;
		uint64_t chunk = bit / 64;
		uint64_t bitInChunk = bit % 64;
		int oldBitValue = (int)((b->bitmap[chunk] << (63 - bitInChunk)) >> 63);
		if (oldBitValue == 1) continue;
		uint64_t filter = 1;
		filter <<= bitInChunk;
		b->bitmap[chunk] |= filter;
		b->set_bits++;
	}
	b->insertions++;
}

void ro_bloom_bit_indices(const uint64_t lower[2], int W, uint64_t m_bits, uint64_t out[8]) {
	uint64_t origin = ro_hash2(lower, W);
	for (int i = 0; i < 8; i++) out[i] = (origin ^ RO_BLOOM_MAGIC[i]) % m_bits;
}

uint64_t ro_default_bloom_bits(int k, uint64_t local_reads) {
	uint64_t numberOfErrorsPerRead = 4;
	uint64_t erroneousKmersPerError = (uint64_t)k * 2 * 2;
	uint64_t bits = numberOfErrorsPerRead * erroneousKmersPerError * local_reads;
	uint64_t maximumBits = 1024;
	maximumBits *= 1024 * 1024 * 8;
	if (bits > maximumBits) bits = maximumBits;
	return bits;
}

/* ------------------------------------------------------------------ A10/A11 Vertex bitmap */

/* Vertex.cpp:363-415 */
static uint8_t swap_bits(uint8_t map, int bit1, int bit2) {
	int bit1Value = (int)(((uint64_t)map << (63 - bit1)) >> 63);
	int bit2Value = (int)(((uint64_t)map << (63 - bit2)) >> 63);
	if (bit1Value == bit2Value) return map;
	if (bit1Value == 1) {
		map |= (uint8_t)(1 << bit2);
		map &= (uint8_t)~(1 << bit1);
	} else if (bit2Value == 1) {
		map |= (uint8_t)(1 << bit1);
		map &= (uint8_t)~(1 << bit2);
	}
	return map;
}

uint8_t ro_convert_bitmap(uint8_t bitMap) {
	uint8_t newBits4_7 = (uint8_t)(bitMap << 4);
	uint8_t newBits0_3 = (uint8_t)(bitMap >> 4);
	uint8_t baseMap = newBits4_7 | newBits0_3;
	baseMap = swap_bits(baseMap, 7, 4);
	baseMap = swap_bits(baseMap, 6, 5);
	baseMap = swap_bits(baseMap, 3, 0);
	baseMap = swap_bits(baseMap, 2, 1);
	return baseMap;
}

/* the stored key of the Vertex that holds `vertex` is the lower of the pair (GridTable.cpp:120-126) */
static void lower_key_of(const uint64_t vertex[2], int k, int W, uint64_t lower[2]) {
	ro_complement_vertex(vertex, k, lower);
	if (ro_less(vertex, lower, W)) {
		lower[0] = vertex[0];
		lower[1] = vertex[1];
	}
}

/* Vertex.cpp:169-173 with an explicit lower key */
static uint8_t vertex_get_edges(uint8_t stored, const uint64_t lowerKey[2], const uint64_t a[2], int W) {
	if (kmer_equal(a, lowerKey, W)) return stored;
	return ro_convert_bitmap(stored);
}

/* Vertex.cpp:158-163 */
static uint8_t vertex_set_edges(const uint64_t lowerKey[2], const uint64_t a[2], uint8_t edges, int W) {
	if (kmer_equal(a, lowerKey, W)) return edges;
	return ro_convert_bitmap(edges);
}

uint8_t ro_get_edges(uint8_t stored, const uint64_t vertex[2], int k, int W) {
	uint64_t lower[2];
	lower_key_of(vertex, k, W, lower);
	return vertex_get_edges(stored, lower, vertex, W);
}

uint8_t ro_add_outgoing_edge(uint8_t stored, const uint64_t vertex[2], const uint64_t child[2], int k, int W) {
	uint64_t lower[2];
	lower_key_of(vertex, k, W, lower);
	uint8_t s2Last = second_segment_last_code(child, k); /* Vertex.cpp:131-144 */
	uint64_t newBits = (uint64_t)1 << (4 + s2Last);
	return vertex_set_edges(lower, vertex, (uint8_t)(vertex_get_edges(stored, lower, vertex, W) | newBits), W);
}

uint8_t ro_add_ingoing_edge(uint8_t stored, const uint64_t vertex[2], const uint64_t parent[2], int k, int W) {
	uint64_t lower[2];
	lower_key_of(vertex, k, W, lower);
	uint8_t s1First = first_segment_first_code(parent); /* Vertex.cpp:110-116 */
	uint8_t newBits = (uint8_t)(1 << s1First);
	return vertex_set_edges(lower, vertex, (uint8_t)(vertex_get_edges(stored, lower, vertex, W) | newBits), W);
}

/* ------------------------------------------------------------------ the store */

/*
 * Stand-in for MyHashTable<Kmer,Vertex> (RayPlatform, absent).  Its layout is unobservable except
 * for iteration order, so any map works; every consumer below sorts before comparing.
 */
typedef struct {
	uint64_t key[2]; /* m_lowerKey */
	uint32_t coverage; /* m_coverage_lower */
	uint8_t edges; /* m_edges_lower */
	uint8_t used;
} ro_vertex;

typedef struct {
	ro_vertex *slots;
	uint64_t capacity; /* power of two */
	uint64_t count;    /* stored pairs */
	int inserted;      /* GridTable::inserted() */
	uint64_t size;     /* GridTable::m_size, 2 per pair */
} ro_table;

static uint64_t table_hash(const uint64_t key[2]) {
	uint64_t h = key[0] * 0x9E3779B97F4A7C15ULL ^ (key[1] + 0x7F4A7C15ULL) * 0xC2B2AE3D27D4EB4FULL;
	h ^= h >> 29;
	h *= 0xBF58476D1CE4E5B9ULL;
	h ^= h >> 32;
	return h;
}

static void table_init(ro_table *t) {
	t->capacity = 1024;
	t->slots = (ro_vertex *)calloc(t->capacity, sizeof(ro_vertex));
	t->count = 0;
	t->inserted = 0;
	t->size = 0;
}

static ro_vertex *table_raw_find(const ro_table *t, const uint64_t key[2], int W, int *found) {
	uint64_t mask = t->capacity - 1;
	uint64_t i = table_hash(key) & mask;
	for (;;) {
		ro_vertex *v = &t->slots[i];
		if (!v->used) {
			*found = 0;
			return v;
		}
		if (kmer_equal(v->key, key, W)) {
			*found = 1;
			return v;
		}
		i = (i + 1) & mask;
	}
}

static void table_grow(ro_table *t, int W) {
	ro_vertex *old = t->slots;
	uint64_t oldcap = t->capacity;
	t->capacity *= 2;
	t->slots = (ro_vertex *)calloc(t->capacity, sizeof(ro_vertex));
	for (uint64_t i = 0; i < oldcap; i++) {
		if (old[i].used) {
			int f;
			ro_vertex *v = table_raw_find(t, old[i].key, W, &f);
			*v = old[i];
		}
	}
	free(old);
}

/* GridTable::find (GridTable.cpp:84-109) */
static ro_vertex *grid_find(const ro_table *t, const uint64_t key[2], int k, int W) {
	uint64_t lowerKey[2];
	ro_complement_vertex(key, k, lowerKey);
	if (ro_less(key, lowerKey, W)) {
		lowerKey[0] = key[0];
		lowerKey[1] = key[1];
	}
	int found;
	ro_vertex *v = table_raw_find(t, lowerKey, W, &found);
	return found ? v : NULL;
}

/* GridTable::insert (GridTable.cpp:111-134) */
static ro_vertex *grid_insert(ro_table *t, const uint64_t key[2], int k, int W) {
	uint64_t lowerKey[2];
	ro_complement_vertex(key, k, lowerKey);
	if (ro_less(key, lowerKey, W)) {
		lowerKey[0] = key[0];
		lowerKey[1] = key[1];
	}
	if ((t->count + 1) * 10 > t->capacity * 6) table_grow(t, W);
	int found;
	ro_vertex *v = table_raw_find(t, lowerKey, W, &found);
	t->inserted = !found;
	if (!found) {
		v->used = 1;
		v->key[0] = lowerKey[0];
		v->key[1] = W > 1 ? lowerKey[1] : 0;
		t->count++;
		t->size += 2;
	}
	return v;
}

/* Vertex::setCoverage (Vertex.cpp:80-92) */
static void vertex_set_coverage(ro_vertex *v, const uint64_t a[2], uint32_t coverage, int W) {
	if (kmer_equal(a, v->key, W)) {
		uint32_t max = 0;
		max = max - 1;
		if (v->coverage == max) return;
		v->coverage = coverage;
	}
}

/* ------------------------------------------------------------------ the graph */

struct ro_graph {
	int k, W, nranks;
	uint64_t bloom_cfg;
	const uint8_t *packed;
	const uint64_t *offsets;
	const uint32_t *lengths;
	uint64_t nreads;
	ro_table *tables; /* one GridTable per rank */
	ro_bloom *blooms;
	uint64_t *bloom_bits; /* m_bloomBits per rank */
	uint64_t occurrences;
};

ro_graph *ro_graph_create(int k, int W, int nranks, uint64_t bloom_bits) {
	if (k < 1 || k > 32 * W - 1 || (k % 2) == 0 || nranks < 1 || (W != 1 && W != 2)) return NULL;
	ro_graph *g = (ro_graph *)calloc(1, sizeof(ro_graph));
	g->k = k;
	g->W = W;
	g->nranks = nranks;
	g->bloom_cfg = bloom_bits;
	g->tables = (ro_table *)calloc((size_t)nranks, sizeof(ro_table));
	g->blooms = (ro_bloom *)calloc((size_t)nranks, sizeof(ro_bloom));
	g->bloom_bits = (uint64_t *)calloc((size_t)nranks, sizeof(uint64_t));
	for (int r = 0; r < nranks; r++) table_init(&g->tables[r]);
	return g;
}

void ro_graph_destroy(ro_graph *g) {
	if (!g) return;
	for (int r = 0; r < g->nranks; r++) {
		free(g->tables[r].slots);
		free(g->blooms[r].bitmap);
	}
	free(g->tables);
	free(g->blooms);
	free(g->bloom_bits);
	free(g);
}

int ro_graph_set_reads(ro_graph *g, const uint8_t *packed, const uint64_t *offsets, const uint32_t *lengths, uint64_t nreads) {
	g->packed = packed;
	g->offsets = offsets;
	g->lengths = lengths;
	g->nreads = nreads;
	return 0;
}

/* SequencesLoader/SequencesLoader.cpp:287-297 */
static void rank_partition(const ro_graph *g, int rank, uint64_t *first, uint64_t *count) {
	uint64_t sequencesPerRank = g->nreads / (uint64_t)g->nranks;
	uint64_t start = (uint64_t)rank * sequencesPerRank;
	uint64_t end = start + sequencesPerRank; /* exclusive */
	if (rank == g->nranks - 1) end = g->nreads;
	*first = start;
	*count = end - start;
}

/* MessageProcessor.cpp:692-809, one message of `count` words arriving at `rank` */
static void handle_vertices_data(ro_graph *g, int rank, const uint64_t *incoming, int count) {
	int W = g->W, k = g->k;
	ro_table *subgraph = &g->tables[rank];
	ro_bloom *bloom = &g->blooms[rank];
	uint64_t bloomBits = g->bloom_bits[rank];
	for (int i = 0; i < count; i += W) {
		uint64_t kmerObject[2] = {incoming[i], W > 1 ? incoming[i + 1] : 0};
		uint64_t lowerKmer[2] = {kmerObject[0], kmerObject[1]};
		uint64_t reverseComplement[2];
		ro_complement_vertex(kmerObject, k, reverseComplement);
		if (ro_less(reverseComplement, lowerKmer, W)) {
			lowerKmer[0] = reverseComplement[0];
			lowerKmer[1] = reverseComplement[1];
		}
		if (bloomBits > 0 && !bloom_has_value(bloom, lowerKmer, W)) {
			bloom_insert_value(bloom, lowerKmer, W);
			continue;
		}
		ro_vertex *tmp = grid_insert(subgraph, kmerObject, k, W);
		if (subgraph->inserted) {
			tmp->coverage = 0; /* Vertex::constructor */
			tmp->edges = 0;
			uint32_t startingValue = 0;
			if (bloomBits > 0) startingValue++;
			vertex_set_coverage(tmp, kmerObject, startingValue, W);
		}
		uint32_t oldCoverage = tmp->coverage;
		uint32_t newCoverage = oldCoverage + 1;
		if (newCoverage > oldCoverage) vertex_set_coverage(tmp, kmerObject, newCoverage, W);
	}
}

/* a BufferedData stand-in: one 500-word buffer per destination (KmerAcademyBuilder.cpp:238) */
typedef struct {
	uint64_t *words; /* nranks * RO_MESSAGE_WORDS */
	int *counts;
	int nranks;
} ro_buffers;

static void buffers_init(ro_buffers *b, int nranks) {
	b->nranks = nranks;
	b->words = (uint64_t *)malloc(sizeof(uint64_t) * RO_MESSAGE_WORDS * (size_t)nranks);
	b->counts = (int *)calloc((size_t)nranks, sizeof(int));
}

static void buffers_free(ro_buffers *b) {
	free(b->words);
	free(b->counts);
}

int ro_graph_count_kmers(ro_graph *g) {
	int k = g->k, W = g->W;
	/* MessageProcessor.cpp:886-899: each rank sizes its own filter from its own read count */
	for (int r = 0; r < g->nranks; r++) {
		uint64_t first, count;
		rank_partition(g, r, &first, &count);
		uint64_t bits = g->bloom_cfg;
		if (bits == RO_BLOOM_DEFAULT) bits = ro_default_bloom_bits(k, count);
		g->bloom_bits[r] = bits;
		if (bits > 0) bloom_constructor(&g->blooms[r], bits);
	}
	ro_buffers buf;
	buffers_init(&buf, g->nranks);
	char *readSequence = (char *)malloc(RO_MAX_READ + 2);
	char memory[128];
	g->occurrences = 0;
	for (int src = 0; src < g->nranks; src++) {
		uint64_t first, count;
		rank_partition(g, src, &first, &count);
		for (uint64_t id = first; id < first + count; id++) {
			ro_read_get_seq(g->packed + g->offsets[id], (int)g->lengths[id], readSequence);
			int len = (int)strlen(readSequence);
			if (len < k) continue; /* KmerAcademyBuilder.cpp:130-134 */
			int maximumPosition = len - k + 1;
			for (int position = 0; position < maximumPosition; position++) {
				memcpy(memory, readSequence + position, (size_t)k);
				memory[k] = '\0';
				if (!ro_is_valid_dna(memory)) continue;
				uint64_t kmerToSend[2];
				ro_word_id(memory, kmerToSend);
				uint64_t reverseKmer[2];
				ro_complement_vertex(kmerToSend, k, reverseKmer);
				if (ro_less(reverseKmer, kmerToSend, W)) {
					kmerToSend[0] = reverseKmer[0];
					kmerToSend[1] = reverseKmer[1];
				}
				int rankToFlush = ro_vertex_rank(kmerToSend, k, W, g->nranks);
				uint64_t *dst = buf.words + (size_t)rankToFlush * RO_MESSAGE_WORDS;
				for (int i = 0; i < W; i++) dst[buf.counts[rankToFlush]++] = kmerToSend[i];
				g->occurrences++;
				/* BufferedData::flush: send once another entry would not fit */
				if (buf.counts[rankToFlush] + W > RO_MESSAGE_WORDS) {
					handle_vertices_data(g, rankToFlush, dst, buf.counts[rankToFlush]);
					buf.counts[rankToFlush] = 0;
				}
			}
		}
		/* flushAll (KmerAcademyBuilder.cpp:255-261) before KMER_ACADEMY_DISTRIBUTED */
		for (int d = 0; d < g->nranks; d++) {
			if (buf.counts[d] > 0) {
				handle_vertices_data(g, d, buf.words + (size_t)d * RO_MESSAGE_WORDS, buf.counts[d]);
				buf.counts[d] = 0;
			}
		}
	}
	free(readSequence);
	buffers_free(&buf);
	/* MessageProcessor.cpp:984-1046: the filters are destroyed before the histogram; the
	 * statistics stay readable here for tests (bitmap freed, counters kept). */
	for (int r = 0; r < g->nranks; r++) {
		free(g->blooms[r].bitmap);
		g->blooms[r].bitmap = NULL;
	}
	return 0;
}

/* MessageProcessor.cpp:858-884 */
static void handle_out_edges(ro_graph *g, int rank, const uint64_t *incoming, int length) {
	int W = g->W, k = g->k;
	int period = 2 * W;
	for (int i = 0; i < length; i += period) {
		uint64_t prefix[2] = {incoming[i], W > 1 ? incoming[i + 1] : 0};
		uint64_t suffix[2] = {incoming[i + W], W > 1 ? incoming[i + W + 1] : 0};
		ro_vertex *node = grid_find(&g->tables[rank], prefix, k, W);
		if (node == NULL) continue;
		uint8_t s2Last = second_segment_last_code(suffix, k);
		uint64_t newBits = (uint64_t)1 << (4 + s2Last);
		node->edges = vertex_set_edges(node->key, prefix, (uint8_t)(vertex_get_edges(node->edges, node->key, prefix, W) | newBits), W);
	}
}

/* MessageProcessor.cpp:905-968 */
static void handle_in_edges(ro_graph *g, int rank, const uint64_t *incoming, int length) {
	int W = g->W, k = g->k;
	int period = 2 * W;
	for (int i = 0; i < length; i += period) {
		uint64_t prefix[2] = {incoming[i], W > 1 ? incoming[i + 1] : 0};
		uint64_t suffix[2] = {incoming[i + W], W > 1 ? incoming[i + W + 1] : 0};
		ro_vertex *node = grid_find(&g->tables[rank], suffix, k, W);
		if (node == NULL) continue;
		uint8_t s1First = first_segment_first_code(prefix);
		uint8_t newBits = (uint8_t)(1 << s1First);
		node->edges = vertex_set_edges(node->key, suffix, (uint8_t)(vertex_get_edges(node->edges, node->key, suffix, W) | newBits), W);
	}
}

static void push_edge(ro_graph *g, ro_buffers *b, int out, int dest, const uint64_t first[2], const uint64_t second[2]) {
	int W = g->W;
	uint64_t *dst = b->words + (size_t)dest * RO_MESSAGE_WORDS;
	for (int i = 0; i < W; i++) dst[b->counts[dest]++] = first[i];
	for (int i = 0; i < W; i++) dst[b->counts[dest]++] = second[i];
	if (b->counts[dest] + 2 * W > RO_MESSAGE_WORDS) {
		if (out) handle_out_edges(g, dest, dst, b->counts[dest]);
		else handle_in_edges(g, dest, dst, b->counts[dest]);
		b->counts[dest] = 0;
	}
}

int ro_graph_add_edges(ro_graph *g) {
	int k = g->k, W = g->W;
	ro_buffers outb, inb;
	buffers_init(&outb, g->nranks);
	buffers_init(&inb, g->nranks);
	char *readSequence = (char *)malloc(RO_MAX_READ + 2);
	char memory[1000];
	for (int src = 0; src < g->nranks; src++) {
		uint64_t first, count;
		rank_partition(g, src, &first, &count);
		for (uint64_t id = first; id < first + count; id++) {
			int hasPreviousVertex = 0; /* reset at every read end, VerticesExtractor.cpp:267-271 */
			uint64_t previousVertex[2] = {0, 0}, previousVertexRC[2] = {0, 0};
			ro_read_get_seq(g->packed + g->offsets[id], (int)g->lengths[id], readSequence);
			int len = (int)strlen(readSequence);
			if (len < k) continue;
			int maximumPosition = len - k + 1;
			for (int p = 0; p < maximumPosition; p++) {
				memcpy(memory, readSequence + p, (size_t)k);
				memory[k] = '\0';
				if (ro_is_valid_dna(memory)) {
					uint64_t currentForwardKmer[2];
					ro_word_id(memory, currentForwardKmer);
					if (hasPreviousVertex) {
						int outgoingRank = ro_vertex_rank(previousVertex, k, W, g->nranks);
						push_edge(g, &outb, 1, outgoingRank, previousVertex, currentForwardKmer);
						int ingoingRank = ro_vertex_rank(currentForwardKmer, k, W, g->nranks);
						push_edge(g, &inb, 0, ingoingRank, previousVertex, currentForwardKmer);
					}
					uint64_t currentReverseKmer[2];
					ro_complement_vertex(currentForwardKmer, k, currentReverseKmer);
					if (hasPreviousVertex) {
						int outgoingRank = ro_vertex_rank(currentReverseKmer, k, W, g->nranks);
						push_edge(g, &outb, 1, outgoingRank, currentReverseKmer, previousVertexRC);
						int ingoingRank = ro_vertex_rank(previousVertexRC, k, W, g->nranks);
						push_edge(g, &inb, 0, ingoingRank, currentReverseKmer, previousVertexRC);
					}
					hasPreviousVertex = 1;
					previousVertex[0] = currentForwardKmer[0];
					previousVertex[1] = currentForwardKmer[1];
					previousVertexRC[0] = currentReverseKmer[0];
					previousVertexRC[1] = currentReverseKmer[1];
				} else {
					hasPreviousVertex = 0;
				}
			}
		}
		/* flushAll, VerticesExtractor.cpp:351-367 */
		for (int d = 0; d < g->nranks; d++) {
			if (outb.counts[d] > 0) {
				handle_out_edges(g, d, outb.words + (size_t)d * RO_MESSAGE_WORDS, outb.counts[d]);
				outb.counts[d] = 0;
			}
			if (inb.counts[d] > 0) {
				handle_in_edges(g, d, inb.words + (size_t)d * RO_MESSAGE_WORDS, inb.counts[d]);
				inb.counts[d] = 0;
			}
		}
	}
	free(readSequence);
	buffers_free(&outb);
	buffers_free(&inb);
	return 0;
}

/* MessageProcessor.cpp:1224-1256 at the owner of `vertex` */
static uint32_t request_vertex_coverage(const ro_graph *g, const uint64_t vertex[2]) {
	int owner = ro_vertex_rank(vertex, g->k, g->W, g->nranks);
	ro_vertex *node = grid_find(&g->tables[owner], vertex, g->k, g->W);
	uint32_t coverage = 0;
	if (node != NULL) coverage = node->coverage;
	return coverage;
}

/* EdgePurgerWorker.cpp:28-102 for one (Vertex, currentKmer) pair handed out by GridTableIterator */
static void purge_worker(ro_graph *g, ro_vertex *v, const uint64_t currentKmer[2]) {
	int k = g->k, W = g->W;
	uint64_t edges[4][2];
	int n = ro_ingoing_edges(currentKmer, vertex_get_edges(v->edges, v->key, currentKmer, W), k, W, edges);
	for (int i = 0; i < n; i++) {
		if (request_vertex_coverage(g, edges[i]) == 0) {
			/* Vertex::deleteIngoingEdge, Vertex.cpp:118-125 */
			uint8_t s1First = first_segment_first_code(edges[i]);
			uint8_t newBits = (uint8_t)(1 << s1First);
			newBits = (uint8_t)~newBits;
			v->edges = vertex_set_edges(v->key, currentKmer, (uint8_t)(vertex_get_edges(v->edges, v->key, currentKmer, W) & newBits), W);
		}
	}
	n = ro_outgoing_edges(currentKmer, vertex_get_edges(v->edges, v->key, currentKmer, W), k, W, edges);
	for (int i = 0; i < n; i++) {
		if (request_vertex_coverage(g, edges[i]) == 0) {
			/* Vertex::deleteOutgoingEdge, Vertex.cpp:146-152 */
			uint8_t s2Last = second_segment_last_code(edges[i], k);
			uint64_t newBits = (uint64_t)1 << (4 + s2Last);
			newBits = ~newBits;
			v->edges = vertex_set_edges(v->key, currentKmer, (uint8_t)(vertex_get_edges(v->edges, v->key, currentKmer, W) & newBits), W);
		}
	}
}

int ro_graph_purge_edges(ro_graph *g) {
	for (int r = 0; r < g->nranks; r++) {
		ro_table *t = &g->tables[r];
		for (uint64_t i = 0; i < t->capacity; i++) {
			ro_vertex *v = &t->slots[i];
			if (!v->used) continue;
			/* GridTableIterator.cpp:41-54: the stored key first, then its reverse complement */
			purge_worker(g, v, v->key);
			uint64_t rc[2];
			ro_complement_vertex(v->key, g->k, rc);
			purge_worker(g, v, rc);
		}
	}
	return 0;
}

uint64_t ro_graph_size(const ro_graph *g, int rank) {
	if (rank >= 0) return g->tables[rank].size;
	uint64_t s = 0;
	for (int r = 0; r < g->nranks; r++) s += g->tables[r].size;
	return s;
}

uint64_t ro_graph_bloom_bits(const ro_graph *g, int rank) { return g->bloom_bits[rank]; }
uint64_t ro_graph_bloom_set_bits(const ro_graph *g, int rank) { return g->blooms[rank].set_bits; }
uint64_t ro_graph_bloom_insertions(const ro_graph *g, int rank) { return g->blooms[rank].insertions; }
uint64_t ro_graph_occurrences(const ro_graph *g) { return g->occurrences; }

static int cmp_vertex_ptr(const void *a, const void *b) {
	const ro_vertex *x = *(const ro_vertex *const *)a, *y = *(const ro_vertex *const *)b;
	if (x->key[0] != y->key[0]) return x->key[0] < y->key[0] ? -1 : 1;
	if (x->key[1] != y->key[1]) return x->key[1] < y->key[1] ? -1 : 1;
	return 0;
}

/* all stored vertices of `rank` (or of all ranks when rank<0), sorted by key */
static ro_vertex **sorted_vertices(const ro_graph *g, int rank, uint64_t *n_out) {
	uint64_t n = 0;
	for (int r = 0; r < g->nranks; r++)
		if (rank < 0 || r == rank) n += g->tables[r].count;
	ro_vertex **list = (ro_vertex **)malloc(sizeof(ro_vertex *) * (size_t)(n ? n : 1));
	uint64_t j = 0;
	for (int r = 0; r < g->nranks; r++) {
		if (!(rank < 0 || r == rank)) continue;
		const ro_table *t = &g->tables[r];
		for (uint64_t i = 0; i < t->capacity; i++)
			if (t->slots[i].used) list[j++] = &t->slots[i];
	}
	qsort(list, (size_t)n, sizeof(ro_vertex *), cmp_vertex_ptr);
	*n_out = n;
	return list;
}

uint64_t ro_graph_export(const ro_graph *g, int rank, uint8_t *buf, uint64_t cap_records) {
	uint64_t n;
	ro_vertex **list = sorted_vertices(g, rank, &n);
	size_t rec = (size_t)g->W * 8 + 4 + 1;
	for (uint64_t i = 0; i < n && i < cap_records; i++) {
		uint8_t *p = buf + i * rec;
		/* Vertex::dump, Vertex.cpp:438-454 */
		memcpy(p, list[i]->key, (size_t)g->W * 8);
		memcpy(p + g->W * 8, &list[i]->coverage, 4);
		p[g->W * 8 + 4] = list[i]->edges;
	}
	free(list);
	return n;
}

int ro_graph_find(const ro_graph *g, const uint64_t kmer[2], uint32_t *coverage, uint8_t *edges) {
	int owner = ro_vertex_rank(kmer, g->k, g->W, g->nranks);
	ro_vertex *node = grid_find(&g->tables[owner], kmer, g->k, g->W);
	if (node == NULL) {
		if (coverage) *coverage = 0;
		if (edges) *edges = 0;
		return 0;
	}
	if (coverage) *coverage = node->coverage;
	if (edges) *edges = vertex_get_edges(node->edges, node->key, kmer, g->W);
	return 1;
}

static int cmp_u32(const void *a, const void *b) {
	uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
	return x < y ? -1 : (x > y ? 1 : 0);
}

uint64_t ro_graph_histogram(const ro_graph *g, uint32_t *cov, uint64_t *cnt, uint64_t cap) {
	uint64_t total = 0;
	for (int r = 0; r < g->nranks; r++) total += g->tables[r].count;
	uint32_t *all = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(total ? total : 1));
	uint64_t j = 0;
	for (int r = 0; r < g->nranks; r++) {
		const ro_table *t = &g->tables[r];
		for (uint64_t i = 0; i < t->capacity; i++)
			if (t->slots[i].used) all[j++] = t->slots[i].coverage;
	}
	qsort(all, (size_t)total, sizeof(uint32_t), cmp_u32);
	uint64_t n = 0;
	for (uint64_t i = 0; i < total;) {
		uint64_t e = i;
		while (e < total && all[e] == all[i]) e++;
		if (n < cap) {
			cov[n] = all[i];
			cnt[n] = 2 * (e - i); /* the iterator visits the key and its reverse complement */
		}
		n++;
		i = e;
	}
	free(all);
	return n;
}

void ro_graph_degree_distribution(const ro_graph *g, uint64_t out[25]) {
	memset(out, 0, sizeof(uint64_t) * 25);
	uint64_t tmp[4][2];
	for (int r = 0; r < g->nranks; r++) {
		const ro_table *t = &g->tables[r];
		for (uint64_t i = 0; i < t->capacity; i++) {
			const ro_vertex *v = &t->slots[i];
			if (!v->used) continue;
			uint64_t keys[2][2];
			keys[0][0] = v->key[0];
			keys[0][1] = v->key[1];
			ro_complement_vertex(v->key, g->k, keys[1]);
			for (int s = 0; s < 2; s++) {
				uint8_t e = vertex_get_edges(v->edges, v->key, keys[s], g->W);
				int parents = ro_ingoing_edges(keys[s], e, g->k, g->W, tmp);
				int children = ro_outgoing_edges(keys[s], e, g->k, g->W, tmp);
				out[parents * 5 + children]++;
			}
		}
	}
}

uint64_t ro_graph_kmers_txt(const ro_graph *g, char *out, uint64_t cap) {
	uint64_t n;
	ro_vertex **list = sorted_vertices(g, -1, &n);
	uint64_t pos = 0;
	char line[512], word[130];
	uint64_t nb[4][2];
	for (uint64_t i = 0; i < n; i++) {
		uint64_t keys[2][2];
		keys[0][0] = list[i]->key[0];
		keys[0][1] = list[i]->key[1];
		ro_complement_vertex(list[i]->key, g->k, keys[1]);
		for (int s = 0; s < 2; s++) {
			uint8_t e = vertex_get_edges(list[i]->edges, list[i]->key, keys[s], g->W);
			ro_id_to_word(keys[s], g->k, word);
			int len = snprintf(line, sizeof line, "%s;%u;", word, list[i]->coverage);
			int m = ro_ingoing_edges(keys[s], e, g->k, g->W, nb);
			for (int j = 0; j < m; j++) {
				ro_id_to_word(nb[j], g->k, word);
				if (j != 0) line[len++] = ' ';
				line[len++] = word[0];
			}
			line[len++] = ';';
			m = ro_outgoing_edges(keys[s], e, g->k, g->W, nb);
			for (int j = 0; j < m; j++) {
				ro_id_to_word(nb[j], g->k, word);
				if (j != 0) line[len++] = ' ';
				line[len++] = word[g->k - 1];
			}
			line[len++] = '\n';
			if (pos + (uint64_t)len <= cap) memcpy(out + pos, line, (size_t)len);
			pos += (uint64_t)len;
		}
	}
	free(list);
	return pos;
}

/* ------------------------------------------------------------------ A13 CoverageDistribution */

int ro_coverage_distribution(const uint32_t *x, const uint64_t *y, uint64_t n64, const char *path,
                             int *minimum, int *peak, int *repeat) {
	int n = (int)n64;
	if (path != NULL) {
		FILE *f = fopen(path, "w");
		if (!f) return -1;
		fprintf(f, "# KmerCoverage\tFrequency\n");
		fprintf(f, "# Any frequency is a even number because of odd k-mer length\n");
		for (int i = 0; i < n; i++) fprintf(f, "%u %llu\n", x[i], (unsigned long long)y[i]);
		fclose(f);
	}
	if (n == 0) return -2; /* the caller aborts before this point, MachineHelper.cpp:261-268 */

	int windowSize = 10;
	uint32_t minimumX = 1;
	uint64_t minimumY = 2 * 4096;
	uint64_t minimumY2 = 55000;
	uint32_t maximumX = 65535 - 1;
	uint32_t safeThreshold = 256;

	int *votes = (int *)calloc((size_t)n, sizeof(int)); /* map<int,int>: 0 == key absent */
	for (int i = 0; i < n; i++) {
		int largestPosition = i;
		for (int j = 0; j < windowSize; j++) {
			int position = i + j;
			if (position >= n) break;
			if (y[position] > y[largestPosition]) largestPosition = position;
		}
		if (x[largestPosition] > maximumX) continue;
		if (x[largestPosition] < minimumX) continue;
		if (x[largestPosition] >= safeThreshold && y[largestPosition] < minimumY2) continue;
		if (y[largestPosition] > minimumY) votes[largestPosition]++;
	}
	int largestPosition = -1;
	for (int i = 0; i < n; i++) {
		if (votes[i] == 0) continue;
		if (largestPosition < 0) { /* votes.begin()->first */
			largestPosition = i;
		}
		if (votes[i] > votes[largestPosition] || y[i] > y[largestPosition]) largestPosition = i;
	}
	if (largestPosition < 0) largestPosition = 0; /* undefined in the reference; see header */
	free(votes);

	int minimumPosition = largestPosition;
	int i = largestPosition;
	while (i >= 0) {
		if (y[i] <= y[minimumPosition]) minimumPosition = i;
		i--;
	}
	int m_minimumCoverage = (int)x[minimumPosition];
	int m_peakCoverage = (int)x[largestPosition];
	int m_repeatCoverage = 2 * m_peakCoverage;
	int diff = m_peakCoverage - m_minimumCoverage;
	int candidate = m_peakCoverage + diff;
	if (candidate < m_repeatCoverage) m_repeatCoverage = candidate;
	*minimum = m_minimumCoverage;
	*peak = m_peakCoverage;
	*repeat = m_repeatCoverage;
	return 0;
}

/* ------------------------------------------------------------------ next rows: on-disk formats */

/* GenomeGraph checkpoint of one rank: MessageProcessor.cpp:658-678 (writer), :596-655 (reader),
 * record = Vertex::write (Vertex.cpp:257-273) with Kmer::write (Kmer.cpp:366-371).
 * The reference walks its hash table (order not reproducible); here vertices go in key order, each
 * as the iterator hands them out: the stored key, then its reverse complement (GridTableIterator.cpp:41-54). */
int ro_graph_write_genome_graph(const ro_graph *g, int rank, const char *path) {
	FILE *f = fopen(path, "wb");
	if (!f) return -1;
	uint64_t n;
	ro_vertex **list = sorted_vertices(g, rank, &n);
	uint64_t theSize = g->tables[rank].size; /* LargeCount: 2 per stored pair */
	fwrite(&theSize, sizeof(uint64_t), 1, f);
	uint64_t nb[4][2];
	for (uint64_t i = 0; i < n; i++) {
		uint64_t keys[2][2];
		keys[0][0] = list[i]->key[0];
		keys[0][1] = list[i]->key[1];
		ro_complement_vertex(list[i]->key, g->k, keys[1]);
		for (int s = 0; s < 2; s++) {
			int coverage = (int)list[i]->coverage;
			fwrite(keys[s], sizeof(uint64_t), (size_t)g->W, f);
			fwrite(&coverage, sizeof(int), 1, f);
			uint8_t e = vertex_get_edges(list[i]->edges, list[i]->key, keys[s], g->W);
			int numberOfParents = ro_ingoing_edges(keys[s], e, g->k, g->W, nb);
			fwrite(&numberOfParents, sizeof(int), 1, f);
			for (int j = 0; j < numberOfParents; j++) fwrite(nb[j], sizeof(uint64_t), (size_t)g->W, f);
			int numberOfChildren = ro_outgoing_edges(keys[s], e, g->k, g->W, nb);
			fwrite(&numberOfChildren, sizeof(int), 1, f);
			for (int j = 0; j < numberOfChildren; j++) fwrite(nb[j], sizeof(uint64_t), (size_t)g->W, f);
		}
	}
	free(list);
	fclose(f);
	return 0;
}

/* CoverageDistribution checkpoint: MachineHelper.cpp:243-259 (writer), :226-241 (reader) */
int ro_write_coverage_checkpoint(const uint32_t *cov, const uint64_t *cnt, uint64_t n, const char *path) {
	FILE *f = fopen(path, "wb");
	if (!f) return -1;
	int theSize = (int)n;
	fwrite(&theSize, sizeof(int), 1, f);
	for (uint64_t i = 0; i < n; i++) {
		uint32_t coverage = cov[i];
		uint64_t count = cnt[i];
		fwrite(&coverage, sizeof(uint32_t), 1, f);
		fwrite(&count, sizeof(uint64_t), 1, f);
	}
	fclose(f);
	return 0;
}

/* EdgePurger::writeGraphPartition (EdgePurger.cpp:172-217); C++ stream formatting of a double = %g */
int ro_graph_write_partition(const ro_graph *g, const char *path) {
	FILE *f = fopen(path, "w");
	if (!f) return -1;
	uint64_t total = 0;
	for (int i = 0; i < g->nranks; i++) total += g->tables[i].size;
	uint64_t best = total / (uint64_t)g->nranks;
	fprintf(f, "#Rank\tNumberOfKmers\tIdealNumberOfKmers\tDifference\tRelativeDifference\n");
	fprintf(f, "#TotalKmers: %llu\n", (unsigned long long)total);
	fprintf(f, "#Ranks: %d\n", g->nranks);
	fprintf(f, "#IdealNumberOfKmers: %llu\n", (unsigned long long)best);
	for (int i = 0; i < g->nranks; i++) {
		uint64_t actual = g->tables[i].size;
		int difference = 0;
		if (actual < best) {
			difference = (int)(best - actual);
			difference = -difference;
		} else {
			difference = (int)(actual - best);
		}
		double relativeDifference = difference * 100;
		if (best != 0) relativeDifference /= (double)best;
		fprintf(f, "%d\t%llu\t%llu\t%d\t%g%%\n", i, (unsigned long long)actual, (unsigned long long)best, difference, relativeDifference);
	}
	fclose(f);
	return 0;
}

/* CoverageDistributionAnalysis.txt: call_RAY_MASTER_MODE_SEND_COVERAGE_VALUES (MachineHelper.cpp:283-327), statement by
 * statement; distribution_file is what Parameters::getCoverageDistributionFile() returns (prefix + CoverageDistribution.txt).
 * A C++ stream prints a double with 6 significant digits in general notation = %g. */
int ro_write_coverage_analysis(const uint32_t *cov, const uint64_t *cnt, uint64_t n, int wordSize, int minimumCoverage, int peakCoverage,
                               int repeatCoverage, const char *distribution_file, const char *path) {
	uint64_t numberOfVertices = 0;
	uint64_t verticesWith1Coverage = 0;
	uint32_t lowestCoverage = 9999;
	uint64_t genomeKmers = 0;
	for (uint64_t i = 0; i < n; i++) { /* map<CoverageDepth,LargeCount>: ascending coverage */
		uint32_t coverageValue = cov[i];
		uint64_t vertices = cnt[i];
		if (coverageValue < lowestCoverage) {
			verticesWith1Coverage = vertices;
			lowestCoverage = coverageValue;
		}
		if ((int64_t)coverageValue >= (int64_t)minimumCoverage) genomeKmers += vertices;
		numberOfVertices += vertices;
	}
	double percentageSeenOnce = (0.0 + (double)verticesWith1Coverage) / (double)numberOfVertices * 100.00;
	FILE *f = fopen(path, "w");
	if (!f) return -1;
	fprintf(f, "k-mer length:\t%d\n", wordSize);
	fprintf(f, "Number of k-mers in the distributed de Bruijn graph: %llu\n", (unsigned long long)numberOfVertices);
	fprintf(f, "Lowest coverage observed:\t%u\n", lowestCoverage);
	fprintf(f, "MinimumCoverage:\t%d\n", minimumCoverage);
	fprintf(f, "PeakCoverage:\t%d\n", peakCoverage);
	fprintf(f, "RepeatCoverage:\t%d\n", repeatCoverage);
	fprintf(f, "Number of k-mers with at least MinimumCoverage:\t%llu k-mers\n", (unsigned long long)genomeKmers);
	fprintf(f, "Percentage of vertices with coverage %u:\t%g %%\n", lowestCoverage, percentageSeenOnce);
	fprintf(f, "DistributionFile: %s\n", distribution_file);
	fclose(f);
	return 0;
}

/* The reader of the GenomeGraph checkpoint: call_RAY_MPI_TAG_START_INDEXING_SEQUENCES (MessageProcessor.cpp:596-655) into a
 * fresh graph of one rank: per entry Kmer::read, int coverage, GridTable::insert, constructor + setCoverage when it was
 * inserted (setCoverage only acts on the lower key, Vertex.cpp:80-92), then the parents through addIngoingEdge and the
 * children through addOutgoingEdge; the CONFIG_ASSERT checks of :633-641,650 are returned as errors (-2 parents, -3
 * children, -4 size).  The graph must have been created with the file's k and be empty. */
int ro_graph_read_genome_graph(ro_graph *g, const char *path) {
	FILE *f = fopen(path, "rb");
	if (!f) return -1;
	int W = g->W, k = g->k;
	ro_table *subgraph = &g->tables[0];
	uint64_t n = 0;
	if (fread(&n, sizeof(uint64_t), 1, f) != 1) {
		fclose(f);
		return -1;
	}
	int status = 0;
	for (uint64_t i = 0; i < n && status == 0; i++) {
		uint64_t kmer[2] = {0, 0};
		int coverage = 0;
		if (fread(kmer, sizeof(uint64_t), (size_t)W, f) != (size_t)W || fread(&coverage, sizeof(int), 1, f) != 1) {
			status = -1;
			break;
		}
		ro_vertex *tmp = grid_insert(subgraph, kmer, k, W);
		if (subgraph->inserted) {
			tmp->coverage = 0; /* Vertex::constructor */
			tmp->edges = 0;
			vertex_set_coverage(tmp, kmer, (uint32_t)coverage, W);
		}
		int parents = 0;
		if (fread(&parents, sizeof(int), 1, f) != 1) status = -1;
		for (int j = 0; j < parents && status == 0; j++) {
			uint64_t parent[2] = {0, 0};
			if (fread(parent, sizeof(uint64_t), (size_t)W, f) != (size_t)W) status = -1;
			else tmp->edges = ro_add_ingoing_edge(tmp->edges, kmer, parent, k, W);
		}
		int children = 0;
		if (status == 0 && fread(&children, sizeof(int), 1, f) != 1) status = -1;
		for (int j = 0; j < children && status == 0; j++) {
			uint64_t child[2] = {0, 0};
			if (fread(child, sizeof(uint64_t), (size_t)W, f) != (size_t)W) status = -1;
			else tmp->edges = ro_add_outgoing_edge(tmp->edges, kmer, child, k, W);
		}
		if (status == 0) {
			uint64_t nb[4][2];
			uint8_t e = vertex_get_edges(tmp->edges, tmp->key, kmer, W);
			if (ro_ingoing_edges(kmer, e, k, W, nb) != parents) status = -2;
			else if (ro_outgoing_edges(kmer, e, k, W, nb) != children) status = -3;
		}
	}
	fclose(f);
	if (status == 0 && subgraph->size != n) status = -4;
	return status;
}

/* ------------------------------------------------------------------ next row: read ingest */

/* FastqLoader::loadWithPeriod (FastqLoader.cpp:84-119): line 1 of every 4 is the sequence; FastaLoader::load
 * (FastaLoader.cpp:26-64): whitespace-separated tokens, a token starting with '>' opens a record and the
 * rest of its line is skipped, the other tokens are concatenated; every sequence goes through
 * Read::constructor(sequence, trim=true).  is_fastq: 1 = FASTQ, 0 = FASTA.
 * Returns the number of reads; packed/offsets/lengths as ro_graph_set_reads takes them (NULL = count only). */
uint64_t ro_load_sequences(const char *path, int is_fastq, uint8_t *packed, uint64_t packed_cap, uint64_t *offsets,
                           uint32_t *lengths, uint64_t cap_reads, uint64_t *packed_bytes) {
	FILE *f = fopen(path, "r");
	if (!f) return 0;
	static char line[RO_MAX_READ_LENGTH + 8];
	static char seqbuf[RO_MAX_READ_LENGTH + 8];
	static uint8_t work[RO_MAX_READ_LENGTH];
	uint64_t n = 0, used = 0;
#define RO_EMIT(S)                                                              \
	do {                                                                        \
		int len_ = ro_read_pack((S), 1, work);                                  \
		uint64_t bytes_ = (uint64_t)ro_read_required_bytes(len_);               \
		if (packed && n < cap_reads && used + bytes_ <= packed_cap) {           \
			memcpy(packed + used, work, (size_t)bytes_);                        \
			offsets[n] = used;                                                  \
			lengths[n] = (uint32_t)len_;                                        \
		}                                                                       \
		used += bytes_;                                                         \
		n++;                                                                    \
	} while (0)
	if (is_fastq) {
		int rotatingVariable = 0;
		while (fgets(line, RO_MAX_READ_LENGTH, f) != NULL) {
			if (rotatingVariable == 1) RO_EMIT(line);
			rotatingVariable++;
			if (rotatingVariable == 4) rotatingVariable = 0;
		}
	} else {
		char token[RO_MAX_READ_LENGTH + 8];
		int have_id = 0;
		size_t seqlen = 0;
		seqbuf[0] = 0;
		while (fscanf(f, "%65535s", token) == 1) {
			if (token[0] == '>') {
				int c;
				while ((c = fgetc(f)) != EOF && c != '\n') {
				}
				if (have_id) RO_EMIT(seqbuf);
				have_id = 1;
				seqlen = 0;
				seqbuf[0] = 0;
			} else {
				size_t t = strlen(token);
				if (seqlen + t < RO_MAX_READ_LENGTH) {
					memcpy(seqbuf + seqlen, token, t + 1);
					seqlen += t;
				}
			}
		}
		RO_EMIT(seqbuf); /* the last record is pushed unconditionally (FastaLoader.cpp:56-61) */
	}
#undef RO_EMIT
	fclose(f);
	if (packed_bytes) *packed_bytes = used;
	return n;
}

/* ------------------------------------------------------------------ next row: SequencesIndexer markers */

/* IndexerWorker::getThreshold (IndexerWorker.cpp:328-362) */
static int indexer_threshold(const int *data, int n) {
	int multiplicator = 2;
	int bestScore = 0;
	int bestEntry = -1;
	for (int value = 10; value >= 1; value--) {
		int bin1 = value;
		int bin2 = value * multiplicator;
		int bin1Count = 0;
		int bin2Count = 0;
		for (int i = 0; i < n; i++) {
			int point = data[i];
			if (point <= bin1) bin1Count++;
			if (point >= bin2) bin2Count++;
		}
		int score = bin1Count + bin2Count;
		if (bestEntry == -1 || score > bestScore) {
			bestEntry = value;
			bestScore = score;
		}
	}
	return multiplicator * bestEntry;
}

/* IndexerWorker::work (IndexerWorker.cpp:67-248), build without CONFIG_USE_COVERAGE_DISTRIBUTION (the
 * define is commented out, SeedExtender.cpp:40): the coverage of every k-mer of the read is fetched from
 * its owner (MessageProcessor.cpp:1224-1256, 0 = absent), the forward marker is the first position whose
 * coverage reaches the threshold, the reverse marker the last one, stored as length-k-position.
 * forward[i] / reverse[i] = -1 when no position qualifies (the reference then leaves the Read's offsets 0
 * and attaches nothing). */
int ro_graph_read_markers(const ro_graph *g, int32_t *forward, int32_t *reverse) {
	int k = g->k;
	int *data = (int *)malloc(sizeof(int) * 65536);
	char *seq = (char *)malloc(65536 + 1);
	char word[130];
	for (uint64_t r = 0; r < g->nreads; r++) {
		int L = (int)g->lengths[r];
		int n = 0;
		ro_read_get_seq(g->packed + g->offsets[r], L, seq);
		for (int position = 0; !(position > L - k); position++) {
			uint64_t vertex[2];
			memcpy(word, seq + position, (size_t)k);
			word[k] = 0;
			ro_word_id(word, vertex);
			data[n++] = (int)request_vertex_coverage(g, vertex);
		}
		int threshold = indexer_threshold(data, n);
		int selectedPosition = -1;
		for (int i = 0; i < n; i++)
			if (data[i] >= threshold) {
				selectedPosition = i;
				break;
			}
		forward[r] = selectedPosition;
		selectedPosition = -1;
		for (int i = n - 1; i >= 0; i--)
			if (data[i] >= threshold) {
				selectedPosition = i;
				break;
			}
		reverse[r] = selectedPosition == -1 ? -1 : L - k - selectedPosition;
	}
	free(data);
	free(seq);
	return 0;
}
