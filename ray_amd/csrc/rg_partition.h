// rg_partition.h -- the partitioned sweep: super-k-mer records -> two-level bucket partition -> LDS
// aggregation -> entry log.
//
// Why: a scattered device-scope atomic costs the same ~17.5 G/s chip-wide whatever it touches
// (tools/microbench.hip), so a sweep that pays one global atomic per k-mer occurrence tops out
// there.  LDS atomics run ~100x faster.  To let a workgroup aggregate in LDS, all occurrences of a
// canonical k-mer must meet in one place first.  Moving k-mers themselves would cost 16 B per
// occurrence per pass; instead consecutive k-mers of a read that share a bucket travel together as
// one *super-k-mer record* (their bases + one header), ~2.5 B per occurrence:
//
//   k_superkmers   reads -> records.  bucket(k-mer) = hash of its canonical minimizer (the m-mer,
//                  m = min(k,15), with the smallest mixed value among the k-mer's m-mers on both
//                  strands), which is strand independent, so every occurrence of a canonical k-mer
//                  gets the same bucket.
//   k_scatter      two passes (coarse 8 bits, fine 8 bits): tile-local counting sort in LDS, one
//                  cursor claim per (tile, bucket), coalesced runs out.
//   k_aggregate    one workgroup per fine bucket: expand the records, merge (min stamp, count, edge
//                  bits) per canonical k-mer in an LDS hash table, append the distinct entries to a log.
//   k_merge        (raygraph.hip) builds the vertex table from the log: ~1 entry per distinct k-mer
//                  instead of one atomic per occurrence.
//
// Correctness never depends on the bucketing: every stage only moves or merges (key, stamp, count,
// edge bits) contributions and the final merge is the same commutative atomic upsert as the direct
// sweep, so a k-mer split over two buckets, or an LDS table that fills up, costs time, not parity.
#pragma once
#include "rg_device.h"

namespace rg {

static constexpr int MMER_MAX = 15;          // minimizer length (<= 15 so that it fits 30 bits)
static constexpr uint64_t REC_INVALID = ~0ull;
static constexpr uint32_t REC_CHUNK = 1024;  // records a wavefront claims at a time from the output cursor
static constexpr int NB_COARSE = 256;  // first-level digit; further levels split by up to 8 more bits each
static constexpr int MAX_BUCKET_BITS = 24;

// A record: 2*W words of bases (2 bits each, same order as the packed read) then a header word:
//   bits  0..15 position of the first k-mer in its read     bits 24..30 offset of the minimizer in the record
//   bits 16..21 number of k-mers - 1                         bits 31..63 global read id (33 bits)
//   bit  22     a k-mer precedes the first one in the read (then base 0 of the record is its first base)
//   bit  23     a k-mer follows the last one in the read (then the base after the last k-mer is present)
template <int W>
struct Rec {
	uint64_t w[2 * W + 1];
};

// The same record in 20 bytes, the form it takes in memory when k <= 31: a record of w = k - m + 1 <= 17 k-mers spans at most
// 1 + k + 16 + 1 <= 49 bases (98 bits), its minimizer starts before base 64 and the header's two top bits of those fields are
// zero, which leaves 160 - 98 = 62 bits for the other 62 of the header: position, number of k-mers - 1 (5 bits), the two
// neighbour bits, minimizer offset (6 bits), read id.  Every pass over the records is bound by their bytes: a sixth less
// of them.  Kernels work on the unpacked form; only loads and stores see this one.
//   bits 0..97 bases          bits 98..159: header bits 0..20, 22..29, 31..63
struct Rec20 {
	uint32_t w[5];
};
template <int W, bool C>
struct RecMemT {
	using type = Rec<W>;
};
template <>
struct RecMemT<1, true> {
	using type = Rec20;
};
template <int W, bool C>
using RecMem = typename RecMemT<W, C>::type;
__host__ __device__ inline bool rec_compact_for(int W, int k) { return W == 1 && k <= 31; }

__device__ __forceinline__ Rec<1> rec_unpack(const Rec20 &c) {
	Rec<1> r;
	r.w[0] = (uint64_t)c.w[0] | ((uint64_t)c.w[1] << 32);
	r.w[1] = (uint64_t)c.w[2] | ((uint64_t)(c.w[3] & 3u) << 32);
	const uint64_t h62 = (uint64_t)(c.w[3] >> 2) | ((uint64_t)c.w[4] << 30);
	const uint64_t H = (h62 & 0x1FFFFFull) | (((h62 >> 21) & 0xFFull) << 22) | ((h62 >> 29) << 31);
	r.w[2] = h62 == 0x3FFFFFFFFFFFFFFFull ? REC_INVALID : H;
	return r;
}
__device__ __forceinline__ Rec20 rec_pack(const Rec<1> &r, Rec20 *) {
	const uint64_t H = r.w[2];
	const uint64_t h62 = (H & 0x1FFFFFull) | (((H >> 22) & 0xFFull) << 21) | ((H >> 31) << 29);
	Rec20 c;
	c.w[0] = (uint32_t)r.w[0];
	c.w[1] = (uint32_t)(r.w[0] >> 32);
	c.w[2] = (uint32_t)r.w[1];
	c.w[3] = ((uint32_t)(r.w[1] >> 32) & 3u) | ((uint32_t)h62 << 2);
	c.w[4] = (uint32_t)(h62 >> 30);
	return c;
}
template <int W>
__device__ __forceinline__ const Rec<W> &rec_unpack(const Rec<W> &r) {
	return r;
}
template <int W>
__device__ __forceinline__ const Rec<W> &rec_pack(const Rec<W> &r, Rec<W> *) {
	return r;
}
__device__ __forceinline__ bool rec_valid(const Rec20 &c) { return (c.w[3] >> 2) != 0x3FFFFFFFu || c.w[4] != 0xFFFFFFFFu; }
template <int W>
__device__ __forceinline__ bool rec_valid(const Rec<W> &r) {
	return r.w[2 * W] != REC_INVALID;
}
__device__ __forceinline__ void rec_mark_invalid(Rec20 *c) {
	c->w[3] = 0xFFFFFFFFu;
	c->w[4] = 0xFFFFFFFFu;
}
template <int W>
__device__ __forceinline__ void rec_mark_invalid(Rec<W> *r) {
	r->w[2 * W] = REC_INVALID;
}

__host__ __device__ inline uint32_t fmix32(uint32_t h) {  // murmur3 finalizer: a bijection on 32 bits
	h ^= h >> 16;
	h *= 0x85ebca6bu;
	h ^= h >> 13;
	h *= 0xc2b2ae35u;
	h ^= h >> 16;
	return h;
}
// bucket of a canonical minimizer; a different mix than the one that orders minimizers, whose small
// values are exactly the ones that get picked
// 32 bits: the partition levels take the top bits (up to MAX_BUCKET_BITS of them), what is left below splits a bucket whose
// k-mers overflow the aggregation table by whole records (k_aggregate<SLOW>)
static constexpr int BUCKET_TOP = 32;
__host__ __device__ inline uint32_t bucket_number(uint32_t canon) { return fmix32(canon * 0x2545F491u + 0x68E31DA4u); }

__host__ __device__ inline int lmax_for(int W, int k) {
	int l = 64 * W - k - 1;
	return l > 64 ? 64 : l;
}

__device__ __forceinline__ uint32_t canon_mmer(uint64_t f, int m) {
	uint64_t r = (~rev2(f)) >> (64 - 2 * m);
	return (uint32_t)(f < r ? f : r);
}

// Two-word keys: the compiler turns the chain of selects over a record's four words into one load at a computed address,
// which leaves the words in a private array -- in LDS while there is room for 64 bytes per thread, in scratch memory
// otherwise.  An empty asm statement per word makes each a register value that only selects can choose between.
__device__ __forceinline__ uint64_t in_register(uint64_t v) {
	uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
	asm volatile("" : "+v"(lo), "+v"(hi));
	return ((uint64_t)hi << 32) | lo;
}

// base t of a record
template <int W>
__device__ __forceinline__ uint32_t rec_base(const uint64_t (&b)[2 * W], uint32_t t) {
	uint32_t j = t >> 5;
	if (W == 2) {
		const uint64_t b0 = in_register(b[0]), b1 = in_register(b[1]), b2 = in_register(b[2]), b3 = in_register(b[3]);
		const uint64_t lo = (j & 1u) ? b1 : b0, hi = (j & 1u) ? b3 : b2;
		const uint64_t word = (j & 2u) ? hi : lo;
		return (uint32_t)(word >> ((t & 31u) * 2u)) & 3u;
	}
	uint64_t word = b[0];
	if (j == 1) word = b[1];
	return (uint32_t)(word >> ((t & 31u) * 2u)) & 3u;
}

// 64 bits of a record's base string starting at base t (t < 64*W)
template <int W>
__device__ __forceinline__ uint64_t rec_ext(const uint64_t (&b)[2 * W], uint32_t t) {
	uint32_t j = t >> 5, s = (t & 31u) * 2u;
	uint64_t lo = b[0], hi = b[1];
	if (W == 2) {
		const uint64_t b0 = in_register(b[0]), b1 = in_register(b[1]), b2 = in_register(b[2]), b3 = in_register(b[3]);
		lo = (j & 2u) ? ((j & 1u) ? b3 : b2) : ((j & 1u) ? b1 : b0);
		hi = (j & 2u) ? ((j & 1u) ? 0ull : b3) : ((j & 1u) ? b2 : b1);
	} else if (j == 1) {
		lo = b[1];
		hi = 0;
	}
	return s ? ((lo >> s) | (hi << (64 - s))) : lo;
}

template <int W>
__device__ __forceinline__ uint32_t rec_bucket(const Rec<W> &r, int m) {
	uint64_t b[2 * W];
#pragma unroll
	for (int i = 0; i < 2 * W; ++i) b[i] = r.w[i];
	uint32_t moff = (uint32_t)(r.w[2 * W] >> 24) & 127u;
	uint64_t f = rec_ext<W>(b, moff) & ((1ull << (2 * m)) - 1);
	return bucket_number(canon_mmer(f, m));
}

template <int W>
__device__ __forceinline__ uint32_t rec_bucket(const Rec20 &c, int m) {  // W == 1
	const uint64_t b[2] = {(uint64_t)c.w[0] | ((uint64_t)c.w[1] << 32), (uint64_t)c.w[2] | ((uint64_t)(c.w[3] & 3u) << 32)};
	const uint32_t moff = (c.w[3] >> 25) & 63u;  // header bits 24..29 = bits 23..28 of the 62, which start at bit 2 of w[3]
	const uint64_t f = rec_ext<1>(b, moff) & ((1ull << (2 * m)) - 1);
	return bucket_number(canon_mmer(f, m));
}

enum PartCounter { PC_REC_CURSOR = 16, PC_REC_OVF, PC_LOG_CURSOR, PC_LOG_OVF, PC_LDS_SPILL, PC_AGG_RETRY, PC_NONUNIQUE, PC_LOG_KEYS, PC_SAMPLE, PC_SAMPLE_DISTINCT, PC_SAMPLE_FAIL, PC_CUT_ANCHORED = 30, PC_CUT_EXPANDED = 31,
                   PC_LOG_MULTI = 10 /* = C_LOG_MULTI */, PC_ITEM_OVF = 13 };

// The sample of the records that sizes the partition: the records whose bucket number has its top SAMPLE_BITS bits clear,
// i.e. the first 2^(b - SAMPLE_BITS) buckets of a partition into 2^b, whole (the bucket number is a hash: 1 / 4096 of the
// input, whatever it is).  The kernels that see every record anyway copy them aside (sample != nullptr); the host counts the
// distinct k-mers in them with a small sweep of its own and picks the number of buckets from that.
static constexpr int SAMPLE_BITS = 12;
template <typename RM>
__device__ __forceinline__ void sample_record(const RM &rec, uint32_t bucket, RM *__restrict__ sample, uint64_t sample_cap,
                                              unsigned long long *__restrict__ counters) {
	if (sample && (bucket >> (BUCKET_TOP - SAMPLE_BITS)) == 0) {
		const unsigned long long pos = atomicAdd(&counters[PC_SAMPLE], 1ull);
		if (pos < sample_cap) sample[pos] = rec;
	}
}

// ------------------------------------------------------------------------------------------------
// reads -> super-k-mer records.  A wavefront takes 64 consecutive reads at a time (their lengths and
// offsets arrive in one coalesced load) and walks them one by one, SK_STEP k-mer positions per step,
// every lane owning two positions.  Per step:
//   1. the words of the read that the step needs are staged in LDS (one lane per 32-bit word),
//   2. every m-mer is made canonical and mixed once: x[j] = (32-bit bijective mix) << 32 | j.  The order is
//      total on canonical m-mers (equal mix = equal m-mer), so the minimizer of a k-mer, hence its bucket,
//      depends on the k-mer alone and not on the strand or the read it was seen in: every k-mer lives in
//      exactly one bucket (k_aggregate / k_merge rely on it for their unique-entry fast path),
//   3. the minimum of each window of w = k-m+1 entries comes from a sparse table built with
//      floor(log2 w) doubling passes over LDS instead of w reads per position,
//   4. a record starts wherever the minimizer *occurrence* changes (so a record never holds more than
//      w k-mers; the host checks w <= lmax_for()); the starts are compacted through LDS so that record i
//      of the step is assembled by lane i and consecutive lanes store consecutive records.
// The kernel is bound by VALU issue, not by bytes: everything here is about instructions per position.
static constexpr uint32_t SK_STEP = 128;
static constexpr uint32_t SK_X = 192;     // m-mers per step: SK_STEP + w - 1 <= 128 + 49
static constexpr uint32_t SK_XPAD = 256;  // + sentinels that the doubling passes may read (j + 32 <= 223)
static constexpr uint32_t SK_WORDS = 24;  // staged 32-bit words: 2 * (1 + 128 + 63 + 1) bits + misalignment, + 2 read ahead

// bijective; 0 (poly-A) must not map to the minimum.  One multiplication: the order is decided by the top bits of the
// product, and those depend on every bit of c (a second round of mixing would only help the low bits, and 32-bit
// multiplications are quarter rate: the super-k-mer kernels are bound by instruction issue)
// the first active lane's value as a wave-uniform (scalar) one: what is indexed from it is addressed with scalar arithmetic
// (a __shfl from lane 0 leaves it in a vector register: 64-bit multiply-adds per lane for every record address)
__device__ __forceinline__ unsigned long long wave_first(unsigned long long v) {
	return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
	       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}
__device__ __forceinline__ uint32_t order_mix(uint32_t c) { return (c ^ 0x5bd1e995u) * 0x9E3779B1u; }

template <int W, bool C = false>
__global__ __launch_bounds__(256) void k_superkmers(const uint64_t *__restrict__ P, const uint64_t *__restrict__ offsets,
                                                    const uint32_t *__restrict__ lengths, uint64_t nreads, uint64_t read_id_base,
                                                    int k, int m, RecMem<W, C> *__restrict__ recs, uint64_t rec_cap,
                                                    unsigned long long *__restrict__ counters,
                                                    unsigned long long *__restrict__ coarse_hist, uint32_t modulo, uint32_t hist_shift,
                                                    RecMem<W, C> *__restrict__ sample, uint64_t sample_cap) {
	using RM = RecMem<W, C>;
	// coarse_hist[256]: records per first-level digit (bucket >> hist_shift, at most 8 bits), or, with modulo != 0, per
	// destination rank bucket % modulo (record exchange between ranks, modulo <= 256)
	constexpr int NB = 2 * W;
	__shared__ unsigned long long s_x[4][2][SK_XPAD];
	__shared__ uint32_t s_cn[4][SK_X];
	__shared__ uint32_t s_wd[4][SK_WORDS];
	__shared__ uint32_t s_list[4][SK_STEP + 4];
	__shared__ uint32_t s_hist[NB_COARSE];
	const uint32_t *__restrict__ P32 = reinterpret_cast<const uint32_t *>(P);
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	uint32_t *const wd = s_wd[wave];
	uint32_t *const cn = s_cn[wave];
	uint32_t *const list = s_list[wave];
	s_hist[threadIdx.x] = 0;
	s_x[wave][0][SK_X + lane] = ~0ull;
	s_x[wave][1][SK_X + lane] = ~0ull;
	__syncthreads();
	const uint32_t w = (uint32_t)(k - m + 1);
	const uint32_t maskm = (1u << (2 * m)) - 1;
	const uint32_t rshift = 32u - 2u * (uint32_t)m;
	const unsigned long long lt_mask = (1ull << lane) - 1;
	const uint64_t nwaves = (uint64_t)gridDim.x * 4;
	const uint64_t ngroups = (nreads + 63) / 64;
	uint64_t chunk_base = 0;
	uint32_t chunk_used = REC_CHUNK;
	bool have_chunk = false, dead = false;
	unsigned long long occ = 0;

	for (uint64_t g = (uint64_t)blockIdx.x * 4 + wave; g < ngroups; g += nwaves) {
		const uint64_t rb = g * 64;
		uint32_t myL = 0, myoff_lo = 0, myoff_hi = 0;
		if (rb + lane < nreads) {
			myL = lengths[rb + lane];
			const uint64_t o = offsets[rb + lane] * 8;  // bit offset of the read
			myoff_lo = (uint32_t)o;
			myoff_hi = (uint32_t)(o >> 32);
		}
		const uint32_t nin = (nreads - rb < 64) ? (uint32_t)(nreads - rb) : 64u;
		for (uint32_t i = 0; i < nin; ++i) {
			const uint32_t L = (uint32_t)__builtin_amdgcn_readlane((int)myL, (int)i);
			if (L < (uint32_t)k) continue;
			const uint64_t bit0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)myoff_hi, (int)i) << 32) |
			                      (uint32_t)__builtin_amdgcn_readlane((int)myoff_lo, (int)i);
			const uint32_t n = L - (uint32_t)k + 1;
			const uint64_t rid = (read_id_base + rb + i) << 31;
			occ += n;
			for (uint32_t p0 = 0; p0 < n; p0 += SK_STEP) {
				const uint32_t cnt = (n - p0 < SK_STEP) ? (n - p0) : SK_STEP;
				const uint32_t fb = p0 ? p0 - 1 : 0;  // first base the step can need (left neighbour of k-mer p0)
				const uint64_t bitf = bit0 + 2ull * fb;
				const uint64_t w0 = bitf >> 5;
				const uint32_t sh = (uint32_t)bitf & 31u;  // base j >= fb starts at bit sh + 2 (j - fb) of the staged words
				const uint32_t lastb = (p0 + cnt + (uint32_t)k - 1 < L - 1) ? (p0 + cnt + (uint32_t)k - 1) : (L - 1);
				const uint32_t nw = (sh + 2u * (lastb - fb) + 2u + 31u) >> 5;
				if (lane < SK_WORDS) wd[lane] = (lane < nw) ? P32[w0 + lane] : 0u;
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

				// canonical m-mers p0 .. p0+need-1, mixed
				const uint32_t need = cnt + w - 1;
				const uint32_t mb = sh + 2u * (p0 - fb);  // bit of m-mer 0 of the step
				unsigned long long x[3];
#pragma unroll
				for (uint32_t e = 0; e < 3; ++e) {
					x[e] = ~0ull;
					if (64u * e < need) {
						const uint32_t j = lane + 64u * e;
						uint32_t c = 0;
						if (j < need) {
							const uint32_t rbit = mb + 2u * j;
							const uint32_t wi = rbit >> 5;
							const uint32_t f = __builtin_amdgcn_alignbit(wd[wi + 1], wd[wi], rbit & 31u) & maskm;
							uint32_t y = __brev(f);
							y = ((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1);
							const uint32_t rc = (~y) >> rshift;
							c = f < rc ? f : rc;
							x[e] = ((unsigned long long)order_mix(c) << 32) | j;
						}
						s_x[wave][0][j] = x[e];
						cn[j] = c;
					}
				}
				// sparse table: after the pass with stride s, entry j is the minimum of x[j .. j+2s)
				uint32_t cur = 0, s = 1;
				for (; 2u * s <= w; s <<= 1) {
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
					__builtin_amdgcn_wave_barrier();
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
					const unsigned long long *src = s_x[wave][cur];
					unsigned long long *dst = s_x[wave][cur ^ 1u];
#pragma unroll
					for (uint32_t e = 0; e < 3; ++e)
						if (64u * e < need) {
							const uint32_t j = lane + 64u * e;
							const unsigned long long o = src[j + s];
							x[e] = o < x[e] ? o : x[e];
							dst[j] = x[e];
						}
					cur ^= 1u;
				}
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
				// s is now the largest power of two <= w: window [p, p+w) = [p, p+s) u [p+w-s, p+w)
				const unsigned long long *tab = s_x[wave][cur];
				const uint32_t tail = w - s;
				uint32_t jr[2];
				bool act[2];
#pragma unroll
				for (uint32_t e = 0; e < 2; ++e) {
					const uint32_t p = lane + 64u * e;
					act[e] = p < cnt;
					const unsigned long long o = tab[p + tail];
					const unsigned long long v = o < x[e] ? o : x[e];
					jr[e] = (uint32_t)v & 255u;  // where the minimizer of k-mer p0+p starts, relative to p0
				}
				uint32_t prev0 = __shfl_up(jr[0], 1);
				uint32_t prev1 = __shfl_up(jr[1], 1);
				const uint32_t last0 = (uint32_t)__builtin_amdgcn_readlane((int)jr[0], 63);
				if (lane == 0) prev1 = last0;
				const bool st0 = act[0] && (lane == 0 || jr[0] != prev0);
				const bool st1 = act[1] && (jr[1] != prev1);
				const unsigned long long b0 = __ballot(st0), b1 = __ballot(st1);
				const uint32_t c0 = (uint32_t)__popcll(b0);
				const uint32_t c = c0 + (uint32_t)__popcll(b1);
				if (chunk_used + c > REC_CHUNK) {  // wave-uniform
					if (have_chunk && !dead)
						for (uint32_t q = chunk_used + lane; q < REC_CHUNK; q += 64) rec_mark_invalid(&recs[chunk_base + q]);
					unsigned long long nbase = 0;
					if (lane == 0) nbase = atomicAdd(&counters[PC_REC_CURSOR], (unsigned long long)REC_CHUNK);
					nbase = wave_first(nbase);
					if (nbase + REC_CHUNK > rec_cap) {
						dead = true;
						if (lane == 0) atomicAdd(&counters[PC_REC_OVF], 1ull);
					}
					chunk_base = nbase;
					chunk_used = 0;
					have_chunk = true;
				}
				if (st0) list[(uint32_t)__popcll(b0 & lt_mask)] = lane | (jr[0] << 8);
				if (st1) list[c0 + (uint32_t)__popcll(b1 & lt_mask)] = (lane + 64u) | (jr[1] << 8);
				if (lane == 0) list[c] = cnt;
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
				if (!dead)
					for (uint32_t i0 = 0; i0 < c; i0 += 64) {
						const uint32_t idx = i0 + lane;
						if (idx < c) {
							const uint32_t ent = list[idx];
							const uint32_t pr = ent & 255u, jm = ent >> 8;
							const uint32_t len = (list[idx + 1] & 255u) - pr;
							const uint32_t p = p0 + pr;
							const uint32_t hasL = p > 0, hasR = (p + len) < n;
							const uint32_t first = p - hasL;  // read position of the record's base 0
							const uint32_t nbases = hasL + len + (uint32_t)k - 1 + hasR;
							const uint32_t rbit = sh + 2u * (first - fb);
							Rec<W> rec;
#pragma unroll
							for (int q = 0; q < NB; ++q) {
								const uint32_t bq = rbit + 64u * q;
								const uint32_t wi = bq >> 5, sf = bq & 31u;
								uint64_t word = 0;
								if (nbases > 32u * q) {
									const uint32_t a0 = wd[wi], a1 = wd[wi + 1], a2 = wd[wi + 2];
									word = (uint64_t)__builtin_amdgcn_alignbit(a1, a0, sf) | ((uint64_t)__builtin_amdgcn_alignbit(a2, a1, sf) << 32);
									// bits past the last base are zeroed: equal super-k-mers are equal records (k_aggregate merges them)
									const uint32_t rem = nbases - 32u * q;
									if (rem < 32u) word &= (1ull << (2u * rem)) - 1;
								}
								rec.w[q] = word;
							}
							const uint32_t moff = p0 + jm - first;
							rec.w[NB] = (uint64_t)p | ((uint64_t)(len - 1) << 16) | ((uint64_t)hasL << 22) | ((uint64_t)hasR << 23) |
							            ((uint64_t)moff << 24) | rid;
							const RM packed = rec_pack(rec, (RM *)nullptr);
							recs[chunk_base + chunk_used + idx] = packed;
							const uint32_t bk = bucket_number(cn[jm]);
							atomicAdd(&s_hist[modulo ? bk % modulo : bk >> hist_shift], 1u);
							sample_record<RM>(packed, bk, sample, sample_cap, counters);
						}
					}
				chunk_used += c;
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
			}
		}
	}
	if (have_chunk && !dead)
		for (uint32_t q = chunk_used + lane; q < REC_CHUNK; q += 64) rec_mark_invalid(&recs[chunk_base + q]);
	if (lane == 0 && occ) atomicAdd(&counters[2 /*C_OCC*/], occ);
	__syncthreads();
	if (s_hist[threadIdx.x]) atomicAdd(&coarse_hist[threadIdx.x], (unsigned long long)s_hist[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// reads -> super-k-mer records, one *lane* per read (short reads: the same records as k_superkmers cuts inside one
// step, found with about a third of the instructions per position).  The 64 lanes of a wavefront walk their 64
// reads in lock step, one m-mer per step and lane:
//   1. the next base comes out of a 16-base register buffer (one dword load per 16 steps, issued 16 steps ahead),
//      the forward and reverse-complement m-mers roll by one base, the canonical one is mixed: x = mix << 32 | j;
//   2. the minimum of the window of w entries that ends at j is found without looking back over the window: the
//      m-mers are cut in blocks of w; P = running minimum of the current block, S[i] = minimum of entries i.. of the
//      previous block (one backwards pass per block), window minimum = min(S[i+1], P).  The block lives in LDS, one
//      8-byte slot per (entry, lane); slot i holds S[i] of the previous block until step i-1 has read it and x[i]
//      of the current block from step i on, so one array serves both;
//   3. a lane whose minimizer occurrence changed queues a 8-byte descriptor of the record that just ended (ballot
//      + rank into a ring in LDS); whenever 64 are queued, the wavefront assembles 64 records at once, lane per
//      record, bases fetched from the read (cache hits), stores to consecutive slots.
// x orders (mix, position) exactly like k_superkmers, so every k-mer still has one bucket whatever kernel cut it.
// WIDE (windows of more than 24 m-mers: k >= 39, C3's k = 63): the window slots are split into a 32-bit mix and a 16-bit
// position (6 bytes per slot and lane instead of 8: w = 49 then takes 75 KB per workgroup, two workgroups per CU, where
// 8-byte slots would leave one), the position of an entry of the current block is implied by its index and only stored for
// the suffix minima; the descriptor gives 6 bits to the record length and the minimizer offset and 11 to the start.
static constexpr uint32_t SKL_RING = 128;
static constexpr int SKL_MAX_W = 32;       // narrow descriptor: 5 bits each for the record length - 1 and the minimizer offset
static constexpr int SKL_MAX_W_WIDE = 64;  // wide descriptor: 6 bits each, start position below 2048

constexpr uint32_t inv32(uint32_t a) {  // inverse of an odd number modulo 2^32 (Newton)
	uint32_t x = a;
	for (int i = 0; i < 5; ++i) x *= 2u - a * x;
	return x;
}
static_assert(inv32(0x9E3779B1u) * 0x9E3779B1u == 1u, "inv32");
__device__ __forceinline__ uint32_t order_unmix(uint32_t h) {  // order_mix is a bijection: this is its inverse
	return (h * inv32(0x9E3779B1u)) ^ 0x5bd1e995u;
}

template <int W, bool WIDE, bool C = false>
__global__ __launch_bounds__(256) void k_superkmers_lanes(const uint64_t *__restrict__ P, const uint64_t *__restrict__ offsets,
                                                          const uint32_t *__restrict__ lengths, uint64_t nreads, uint64_t read_id_base,
                                                          int k, int m, RecMem<W, C> *__restrict__ recs, uint64_t rec_cap,
                                                          unsigned long long *__restrict__ counters,
                                                          unsigned long long *__restrict__ coarse_hist, uint32_t modulo, uint32_t hist_shift,
                                                          RecMem<W, C> *__restrict__ sample, uint64_t sample_cap) {
	using RM = RecMem<W, C>;
	constexpr int NB = 2 * W;
	extern __shared__ unsigned long long s_win_all[];  // [4 wavefronts][w entries][64 lanes]
	__shared__ unsigned long long s_q[4][SKL_RING];
	__shared__ uint32_t s_hist[NB_COARSE];
	const uint32_t *__restrict__ P32 = reinterpret_cast<const uint32_t *>(P);
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint32_t w = (uint32_t)(k - m + 1);
	unsigned long long *const win = s_win_all + (size_t)wave * w * 64 + lane;  // entry i of this lane: win[i * 64]
	uint32_t *const winm = reinterpret_cast<uint32_t *>(s_win_all) + (size_t)wave * w * 64 + lane;  // WIDE: the mixes ...
	uint16_t *const winj = reinterpret_cast<uint16_t *>(reinterpret_cast<uint32_t *>(s_win_all) + (size_t)4 * w * 64) + (size_t)wave * w * 64 + lane;  // ... and the positions
	constexpr uint32_t SB = WIDE ? 11u : 16u;  // descriptor: start | (length - 1) << SB | lane << LANE_SH | (minimizer - start) << MO_SH
	constexpr uint32_t LNB = WIDE ? 6u : 5u;
	constexpr uint32_t LANE_SH = SB + LNB, MO_SH = LANE_SH + 6u;
	unsigned long long *const ring = s_q[wave];
	s_hist[threadIdx.x] = 0;
	__syncthreads();
	const uint32_t maskm = (1u << (2 * m)) - 1;
	const uint32_t top = 2u * (uint32_t)(m - 1);
	const uint64_t nwaves = (uint64_t)gridDim.x * 4;
	const uint64_t ngroups = (nreads + 63) / 64;
	uint64_t chunk_base = 0;
	uint32_t chunk_used = REC_CHUNK;
	bool have_chunk = false, dead = false;
	unsigned long long occ = 0;
	uint32_t qhead = 0, qtail = 0;  // wave-uniform

	for (uint64_t g = (uint64_t)blockIdx.x * 4 + wave; g < ngroups; g += nwaves) {
		const uint64_t rb = g * 64;
		uint32_t L = 0;
		uint64_t bit0 = 0;
		if (rb + lane < nreads) {
			L = lengths[rb + lane];
			bit0 = offsets[rb + lane] * 8;
		}
		if (L < (uint32_t)k) L = 0;
		const uint32_t n = L ? L - (uint32_t)k + 1 : 0;
		occ += n;
		uint32_t maxn = n;
#pragma unroll
		for (int d = 32; d; d >>= 1) {
			const uint32_t o = __shfl_xor(maxn, d);
			maxn = o > maxn ? o : maxn;
		}
		maxn = (uint32_t)__builtin_amdgcn_readfirstlane((int)maxn);
		if (maxn == 0) continue;
		const uint32_t sh = (uint32_t)bit0 & 31u;
		const uint32_t *__restrict__ rp = P32 + (bit0 >> 5);
		const uint32_t lastw = L ? (sh + 2u * L - 1u) >> 5 : 0u;  // last dword that holds a base of this read
		uint32_t lo = L ? rp[0] : 0u;
		uint32_t pend = (L && 1u <= lastw) ? rp[1] : 0u;
		uint32_t widx = 2, f = 0, r = 0;
		uint32_t pre_hi = ~0u, pre_lo = 0, cur_hi = 0, cur_lo = ~0u, dbase = 0, i = 0;
		const uint32_t lane_term = lane << LANE_SH;
		const uint32_t nsteps = maxn + w;  // m-mers 0 .. max(nm): k-mer p is complete at j = p + w - 1, and p = n closes the last record
		const uint32_t nbase = nsteps + (uint32_t)m - 1;  // bases walked: m-mer j is complete with base j + m - 1
		for (uint32_t t0 = 0; t0 < nbase; t0 += 16) {
			// 16 bases per refill; the dword for the next refill is requested now and not touched inside the inner loop
			const uint32_t hi = pend;
			uint32_t buf = __builtin_amdgcn_alignbit(hi, lo, sh);
			lo = hi;
			asm volatile("" : "+v"(buf) : : "memory");  // the wait for the old dword must not end up behind the new request
			pend = (L && widx <= lastw) ? rp[widx] : 0u;
			++widx;
			const uint32_t tend = (t0 + 16u < nbase) ? t0 + 16u : nbase;
			for (uint32_t t = t0; t < tend; ++t) {
				const uint32_t c = buf & 3u;
				buf >>= 2;
				f = (f >> 2) | (c << top);
				r = ((r << 2) | (c ^ 3u)) & maskm;
				if (t + 1 < (uint32_t)m) continue;  // wave-uniform
				const uint32_t j = t + 1 - (uint32_t)m;
				unsigned long long sfx = ~0ull;
				if (i + 1 < w) {  // S[i+1] of the previous block (garbage in the first block: not used there)
					if (WIDE) sfx = ((unsigned long long)winm[(i + 1) * 64] << 32) | winj[(i + 1) * 64];
					else sfx = win[(i + 1) * 64];
				}
				const uint32_t cm = f < r ? f : r;
				const uint32_t mix = order_mix(cm);
				if (WIDE) winm[i * 64] = mix;  // its position is j - (w - 1 - i) seen from the end of the block
				else win[i * 64] = ((unsigned long long)mix << 32) | j;
				// (mix, position) pairs are ordered by mix, the earlier position winning ties: positions only grow, so
				// 32-bit compares of the mixes do, < where the newcomer is the later one, <= where it is the earlier one
				const bool newpre = i == 0 || mix < pre_hi;
				pre_hi = newpre ? mix : pre_hi;
				pre_lo = newpre ? j : pre_lo;
				if (j + 1 >= w) {  // wave-uniform
					// j = w-1 (first block): i+1 = w, the window is the block
					const bool use_sfx = (i + 1 < w) && (uint32_t)(sfx >> 32) <= pre_hi;
					const uint32_t res_hi = use_sfx ? (uint32_t)(sfx >> 32) : pre_hi, res_lo = use_sfx ? (uint32_t)sfx : pre_lo;
					const uint32_t p = j + 1 - w;
					const bool changed = res_lo != cur_lo;
					const bool push = p > 0 && p <= n && (p == n || changed);
					const unsigned long long bal = __ballot(push);
					if (bal) {
						if (push) {
							// dbase + ((p - 1) << 16) = start | (p - start - 1) << 16 | lane << 21 | (minimizer position - start) << 27
							const uint32_t d = dbase + ((p - 1u) << SB);
							const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
							ring[(qtail + rk) & (SKL_RING - 1)] = ((unsigned long long)cur_hi << 32) | d;
						}
						qtail += (uint32_t)__popcll(bal);
					}
					if (p < n && changed) {  // a record starts at p
						dbase = (res_lo << MO_SH) + (lane_term + p * (1u - (1u << SB) - (1u << MO_SH)));
						cur_lo = res_lo;
						cur_hi = res_hi;
					}
				}
				++i;
				if (i == w) {  // block complete: slot t becomes the minimum of entries t .. w-1
					// four entries per round so that their LDS latencies overlap; slot 0 is nobody's S (indices are clamped to it)
					uint32_t run_hi = mix, run_lo = j;
					if (WIDE) {
						if (w >= 2) winj[(w - 1) * 64] = (uint16_t)j;  // S[w-1] is the last entry itself (slot 0 is nobody's S)
						const uint32_t j0 = j + 1u - w;  // position of entry 0 of this block
						for (int q = (int)w - 2; q >= 0; q -= 4) {
							const int q1 = q > 0 ? q - 1 : 0, q2 = q > 1 ? q - 2 : 0, q3 = q > 2 ? q - 3 : 0;
							const uint32_t m0 = winm[q * 64], m1 = winm[q1 * 64], m2 = winm[q2 * 64], m3 = winm[q3 * 64];
							const bool t0 = m0 <= run_hi;
							run_hi = t0 ? m0 : run_hi;
							run_lo = t0 ? j0 + (uint32_t)q : run_lo;
							winm[q * 64] = run_hi;
							winj[q * 64] = (uint16_t)run_lo;
							const bool t1 = m1 <= run_hi;
							run_hi = t1 ? m1 : run_hi;
							run_lo = t1 ? j0 + (uint32_t)q1 : run_lo;
							winm[q1 * 64] = run_hi;
							winj[q1 * 64] = (uint16_t)run_lo;
							const bool t2 = m2 <= run_hi;
							run_hi = t2 ? m2 : run_hi;
							run_lo = t2 ? j0 + (uint32_t)q2 : run_lo;
							winm[q2 * 64] = run_hi;
							winj[q2 * 64] = (uint16_t)run_lo;
							const bool t3 = m3 <= run_hi;
							run_hi = t3 ? m3 : run_hi;
							run_lo = t3 ? j0 + (uint32_t)q3 : run_lo;
							winm[q3 * 64] = run_hi;
							winj[q3 * 64] = (uint16_t)run_lo;
						}
					} else
					for (int q = (int)w - 2; q >= 0; q -= 4) {
						const int q1 = q > 0 ? q - 1 : 0, q2 = q > 1 ? q - 2 : 0, q3 = q > 2 ? q - 3 : 0;
						const unsigned long long v0 = win[q * 64], v1 = win[q1 * 64], v2 = win[q2 * 64], v3 = win[q3 * 64];
						const bool t0 = (uint32_t)(v0 >> 32) <= run_hi;
						run_hi = t0 ? (uint32_t)(v0 >> 32) : run_hi;
						run_lo = t0 ? (uint32_t)v0 : run_lo;
						win[q * 64] = ((unsigned long long)run_hi << 32) | run_lo;
						const bool t1 = (uint32_t)(v1 >> 32) <= run_hi;
						run_hi = t1 ? (uint32_t)(v1 >> 32) : run_hi;
						run_lo = t1 ? (uint32_t)v1 : run_lo;
						win[q1 * 64] = ((unsigned long long)run_hi << 32) | run_lo;
						const bool t2 = (uint32_t)(v2 >> 32) <= run_hi;
						run_hi = t2 ? (uint32_t)(v2 >> 32) : run_hi;
						run_lo = t2 ? (uint32_t)v2 : run_lo;
						win[q2 * 64] = ((unsigned long long)run_hi << 32) | run_lo;
						const bool t3 = (uint32_t)(v3 >> 32) <= run_hi;
						run_hi = t3 ? (uint32_t)(v3 >> 32) : run_hi;
						run_lo = t3 ? (uint32_t)v3 : run_lo;
						win[q3 * 64] = ((unsigned long long)run_hi << 32) | run_lo;
					}
					i = 0;
				}
				// assemble 64 queued records (all of them after the last step), lane per record
				while (qtail - qhead >= 64u || (j + 1 == nsteps && qtail != qhead)) {
					const uint32_t cnt = (qtail - qhead < 64u) ? qtail - qhead : 64u;
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
					__builtin_amdgcn_wave_barrier();
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
					if (chunk_used + cnt > REC_CHUNK) {
						if (have_chunk && !dead)
							for (uint32_t q = chunk_used + lane; q < REC_CHUNK; q += 64) rec_mark_invalid(&recs[chunk_base + q]);
						unsigned long long nbase = 0;
						if (lane == 0) nbase = atomicAdd(&counters[PC_REC_CURSOR], (unsigned long long)REC_CHUNK);
						nbase = wave_first(nbase);
						if (nbase + REC_CHUNK > rec_cap) {
							dead = true;
							if (lane == 0) atomicAdd(&counters[PC_REC_OVF], 1ull);
						}
						chunk_base = nbase;
						chunk_used = 0;
						have_chunk = true;
					}
					// all lanes take part in the shuffles (a lane that is switched off cannot be read from): the read of lane
					// src gives its length and bit offset over the lane crossbar, not through LDS arrays, whose 3 KB would
					// cost the kernel its fourth workgroup per CU
					const unsigned long long d = ring[(qhead + lane) & (SKL_RING - 1)];
					const uint32_t dl = (uint32_t)d;
					const uint32_t pr = dl & ((1u << SB) - 1u), len = ((dl >> SB) & ((1u << LNB) - 1u)) + 1u, src = (dl >> LANE_SH) & 63u, jrel = (dl >> MO_SH) & ((1u << LNB) - 1u);
					const uint32_t Ls = (uint32_t)__shfl((int)L, (int)src);
					const uint64_t b0 = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(bit0 >> 32), (int)src) << 32) | (uint32_t)__shfl((int)(uint32_t)bit0, (int)src);
					if (!dead && lane < cnt) {
						const uint32_t ns = Ls - (uint32_t)k + 1;
						const uint32_t hasL = pr > 0, hasR = (pr + len) < ns;
						const uint32_t first = pr - hasL;  // read position of the record's base 0
						const uint32_t nbases = hasL + len + (uint32_t)k - 1 + hasR;
						const uint64_t bit = b0 + 2ull * first;
						const uint32_t *__restrict__ wp = P32 + (bit >> 5);
						const uint32_t sf = (uint32_t)bit & 31u;
						const uint32_t ndw = (sf + 2u * nbases + 31u) >> 5;  // dwords that hold the record's bases
						uint32_t a[2 * NB + 1];
#pragma unroll
						for (int q = 0; q < 2 * NB + 1; ++q) a[q] = ((uint32_t)q < ndw) ? wp[q] : 0u;
						Rec<W> rec;
#pragma unroll
						for (int q = 0; q < NB; ++q) {
							uint64_t word = 0;
							if (nbases > 32u * q) {
								word = (uint64_t)__builtin_amdgcn_alignbit(a[2 * q + 1], a[2 * q], sf) |
								       ((uint64_t)__builtin_amdgcn_alignbit(a[2 * q + 2], a[2 * q + 1], sf) << 32);
								// bits past the last base are zeroed: equal super-k-mers are equal records (k_aggregate merges them)
								const uint32_t rem = nbases - 32u * q;
								if (rem < 32u) word &= (1ull << (2u * rem)) - 1;
							}
							rec.w[q] = word;
						}
						const uint32_t moff = jrel + hasL;
						rec.w[NB] = (uint64_t)pr | ((uint64_t)(len - 1) << 16) | ((uint64_t)hasL << 22) | ((uint64_t)hasR << 23) |
						            ((uint64_t)moff << 24) | ((read_id_base + rb + src) << 31);
						const RM packed = rec_pack(rec, (RM *)nullptr);
						recs[chunk_base + chunk_used + lane] = packed;
						const uint32_t bk = bucket_number(order_unmix((uint32_t)(d >> 32)));
						atomicAdd(&s_hist[modulo ? bk % modulo : bk >> hist_shift], 1u);
						sample_record<RM>(packed, bk, sample, sample_cap, counters);
					}
					chunk_used += cnt;
					qhead += cnt;
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
					__builtin_amdgcn_wave_barrier();
				}
			}
		}
	}
	if (have_chunk && !dead)
		for (uint32_t q = chunk_used + lane; q < REC_CHUNK; q += 64) rec_mark_invalid(&recs[chunk_base + q]);
#pragma unroll
	for (int d = 32; d; d >>= 1) occ += __shfl_xor(occ, d);
	if (lane == 0 && occ) atomicAdd(&counters[2 /*C_OCC*/], occ);
	__syncthreads();
	if (s_hist[threadIdx.x]) atomicAdd(&coarse_hist[threadIdx.x], (unsigned long long)s_hist[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// k_superkmers_lanes for one-word keys with the window length known at compile time (WS = k - m + 1: 17 for k = 31, the
// length every BASELINE configuration but one is run at).  Same algorithm, same records; what changes is what the step
// costs.  The generic kernel spends as many scalar as vector instructions per step -- a CU has ONE scalar unit for its
// four SIMDs, and on C2 both units are about 60 % busy, which is why more wavefronts per CU (6-byte slots: five workgroups
// instead of four) bought nothing.  Here a block of WS steps is unrolled: the slot of entry i is an immediate LDS offset,
// "i + 1 < w", "i == 0" and "i == w" are decided by the compiler, the first block needs no special case (p = j + 1 - w
// wraps and matches no read), the suffix pass has no loop, and the bases of a block -- WS <= 17 of them, 34 bits -- come
// from two dwords requested one block ahead.  The one place a step can leave the unrolled block is when 64 records are
// queued: they are assembled at a single site and the block is re-entered at the step it was left (the switch).
// FUSED: **the first partition level is folded in** (round 3).  The records do not go to chunks in the order they are cut
// but to 2^fbits runs per workgroup, by the top fbits bits of their bucket number: run (d, g) of digit d and workgroup g is the
// slice [(d * gridDim.x + g) * run_cap, ... + run_cap) of `recs`, filled through a cursor in LDS (the four wavefronts share
// the workgroup's runs), and the byte of bucket bits below the digit goes to `side` at the same index (the next level's
// histogram reads it).  64 open runs per workgroup of 20-byte stores merge into whole lines in L2 before they reach HBM
// (tools/microbench_streams.hip: at this kernel's pace 64 runs + side bytes cost nothing, 128 runs 4 ms), so a whole
// k_scatter level -- a read and a write of every record -- goes away.  run_count[d * gridDim.x + g] carries the cursors
// from launch to launch (several read batches) and tells the next level how much of each run is filled; a run that
// overflows is counted in PC_REC_OVF and the sweep is done again the classic way.
template <int WS, bool C, bool FUSED = false>
__global__ __launch_bounds__(256) void k_superkmers_lanes_s(const uint64_t *__restrict__ P, const uint64_t *__restrict__ offsets,
                                                            const uint32_t *__restrict__ lengths, uint64_t nreads, uint64_t read_id_base,
                                                            int k, int m, RecMem<1, C> *__restrict__ recs, uint64_t rec_cap,
                                                            unsigned long long *__restrict__ counters,
                                                            unsigned long long *__restrict__ coarse_hist, uint32_t modulo, uint32_t hist_shift,
                                                            RecMem<1, C> *__restrict__ sample, uint64_t sample_cap,
                                                            uint8_t *__restrict__ side = nullptr, uint32_t *__restrict__ run_count = nullptr,
                                                            uint32_t run_cap = 0, int fbits = 0) {
	static_assert(WS >= 2 && WS <= 17, "a block's bases must fit two dwords at any even bit offset");
	using RM = RecMem<1, C>;
	constexpr int NB = 2;
	constexpr uint32_t w = WS;
	extern __shared__ unsigned long long s_win_all[];  // [4 wavefronts][w entries][64 lanes]
	__shared__ unsigned long long s_q[4][SKL_RING];
	__shared__ uint32_t s_hist[NB_COARSE];
	const uint32_t *__restrict__ P32 = reinterpret_cast<const uint32_t *>(P);
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	unsigned long long *const win = s_win_all + (size_t)wave * w * 64 + lane;  // entry i of this lane: win[i * 64]
	constexpr uint32_t SB = 16u, LNB = 5u, LANE_SH = SB + LNB, MO_SH = LANE_SH + 6u;
	unsigned long long *const ring = s_q[wave];
	if constexpr (FUSED) s_hist[threadIdx.x] = threadIdx.x < (1u << fbits) ? run_count[threadIdx.x * gridDim.x + blockIdx.x] : 0u;  // the runs' cursors
	else s_hist[threadIdx.x] = 0;
	__syncthreads();
	const uint32_t maskm = (1u << (2 * m)) - 1;
	const uint32_t top = 2u * (uint32_t)(m - 1);
	const uint64_t nwaves = (uint64_t)gridDim.x * 4;
	const uint64_t ngroups = (nreads + 63) / 64;
	uint64_t chunk_base = 0;
	uint32_t chunk_used = REC_CHUNK;
	bool have_chunk = false, dead = false;
	unsigned long long occ = 0;
	uint32_t qhead = 0, qtail = 0;  // wave-uniform

	for (uint64_t g = (uint64_t)blockIdx.x * 4 + wave; g < ngroups; g += nwaves) {
		const uint64_t rb = g * 64;
		uint32_t L = 0;
		uint64_t bit0 = 0;
		if (rb + lane < nreads) {
			L = lengths[rb + lane];
			bit0 = offsets[rb + lane] * 8;
		}
		if (L < (uint32_t)k) L = 0;
		const uint32_t n = L ? L - (uint32_t)k + 1 : 0;
		occ += n;
		uint32_t maxn = n;
#pragma unroll
		for (int d = 32; d; d >>= 1) {
			const uint32_t o = __shfl_xor(maxn, d);
			maxn = o > maxn ? o : maxn;
		}
		maxn = (uint32_t)__builtin_amdgcn_readfirstlane((int)maxn);
		if (maxn == 0) continue;
		const uint32_t sh = (uint32_t)bit0 & 31u;  // a multiple of 8: reads start on byte boundaries
		const uint32_t *__restrict__ rp = P32 + (bit0 >> 5);
		const uint32_t lastw = L ? (sh + 2u * L - 1u) >> 5 : 0u;  // last dword that holds a base of this read
		const uint32_t nsteps = maxn + w;  // m-mers 0 .. max(nm): k-mer p is complete at j = p + w - 1, and p = n closes the last record
		uint32_t f = 0, r = 0;
		{  // the m - 1 bases in front of the first m-mer
			const uint32_t a0 = L ? rp[0] : 0u, a1 = (L && 1u <= lastw) ? rp[1] : 0u;
			uint32_t buf = __builtin_amdgcn_alignbit(a1, a0, sh);
			for (uint32_t t = 0; t + 1 < (uint32_t)m; ++t) {
				const uint32_t c = buf & 3u;
				buf >>= 2;
				f = (f >> 2) | (c << top);
				r = ((r << 2) | (c ^ 3u)) & maskm;
			}
		}
		// block b holds m-mers j = WS b .. WS b + WS - 1, completed by bases t = j + m - 1: 2 WS bits from bit sh + 2 (WS b + m - 1)
		uint32_t bitb = sh + 2u * (uint32_t)(m - 1);
		uint32_t idx = bitb >> 5;
		uint32_t d0 = (L && idx <= lastw) ? rp[idx] : 0u, d1 = (L && idx + 1 <= lastw) ? rp[idx + 1] : 0u;
		uint32_t pre_hi = ~0u, pre_lo = 0, cur_hi = 0, cur_lo = ~0u, dbase = 0;
		const uint32_t lane_term = lane << LANE_SH;
		uint32_t jb = 0, i0 = 0;
		bool final = false;
		uint32_t lo = 0, c16 = 0, mix = 0;
		unsigned long long sfx = ~0ull;  // S[i + 1] of the previous block for the step about to run: read one step ahead
		for (;;) {
			if (i0 == 0) {  // block start: its bases out of the two dwords, the next block's dwords requested
				sfx = win[1 * 64];
				const uint32_t o = bitb & 31u;
				lo = __builtin_amdgcn_alignbit(d1, d0, o);
				c16 = (d1 >> o) & 3u;
				bitb += 2u * WS;
				idx = bitb >> 5;
				asm volatile("" : "+v"(lo), "+v"(c16) : : "memory");  // the old dwords are used up before the new request
				d0 = (L && idx <= lastw) ? rp[idx] : 0u;
				d1 = (L && idx + 1 <= lastw) ? rp[idx + 1] : 0u;
			}
#define RG_SKS_STEP(I)                                                                                                           \
	case I: {                                                                                                                    \
		const uint32_t j = jb + (I);                                                                                             \
		if (j >= nsteps) {                                                                                                       \
			final = true;                                                                                                        \
			goto flush;                                                                                                          \
		}                                                                                                                        \
		const uint32_t c = ((I) < 16) ? ((lo >> (2 * ((I) & 15))) & 3u) : c16;                                                    \
		f = (f >> 2) | (c << top);                                                                                               \
		r = ((r << 2) | (c ^ 3u)) & maskm;                                                                                       \
		const unsigned long long sfx_now = sfx;                                                                                  \
		if ((I) + 2 < WS) sfx = win[((I) + 2) * 64]; /* for the next step; its slot is rewritten only then */                    \
		const uint32_t cm = f < r ? f : r;                                                                                       \
		mix = order_mix(cm);                                                                                                     \
		win[(I) * 64] = ((unsigned long long)mix << 32) | j;                                                                     \
		const bool newpre = (I) == 0 || mix < pre_hi;                                                                            \
		pre_hi = newpre ? mix : pre_hi;                                                                                          \
		pre_lo = newpre ? j : pre_lo;                                                                                            \
		const bool use_sfx = ((I) + 1 < WS) && (uint32_t)(sfx_now >> 32) <= pre_hi;                                              \
		const uint32_t res_lo = use_sfx ? (uint32_t)sfx_now : pre_lo;                                                                \
		const uint32_t p = j + 1u - w; /* wraps in the first block: matches no position of any read */                           \
		const bool changed = res_lo != cur_lo;                                                                                   \
		const bool push = p > 0 && p <= n && (p == n || changed);                                                                \
		const bool start = p < n && changed;                                                                                     \
		const unsigned long long bal = __builtin_amdgcn_ballot_w64(push);                                                        \
		if (__builtin_amdgcn_ballot_w64(push || start)) {                                                                                           \
			if (push) {                                                                                                          \
				const uint32_t d = dbase + ((p - 1u) << SB);                                                                     \
				const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u)); \
				ring[(qtail + rk) & (SKL_RING - 1)] = ((unsigned long long)cur_hi << 32) | d;                                    \
			}                                                                                                                    \
			qtail += (uint32_t)__popcll(bal);                                                                                    \
			if (start) {                                                                                                         \
				dbase = (res_lo << MO_SH) + (lane_term + p * (1u - (1u << SB) - (1u << MO_SH)));                                 \
				cur_lo = res_lo;                                                                                                 \
				cur_hi = use_sfx ? (uint32_t)(sfx_now >> 32) : pre_hi;                                                               \
			}                                                                                                                    \
			if (qtail - qhead >= 64u) {                                                                                          \
				i0 = (I) + 1;                                                                                                    \
				goto flush;                                                                                                      \
			}                                                                                                                    \
		}                                                                                                                        \
	}                                                                                                                            \
		[[fallthrough]];
			switch (i0) {
				RG_SKS_STEP(0)
				RG_SKS_STEP(1)
				RG_SKS_STEP(2)
				RG_SKS_STEP(3)
				RG_SKS_STEP(4)
				RG_SKS_STEP(5)
				RG_SKS_STEP(6)
				RG_SKS_STEP(7)
				RG_SKS_STEP(8)
				RG_SKS_STEP(9)
				RG_SKS_STEP(10)
				RG_SKS_STEP(11)
				RG_SKS_STEP(12)
				RG_SKS_STEP(13)
				RG_SKS_STEP(14)
				RG_SKS_STEP(15)
				RG_SKS_STEP(16)
			default:
				break;
			}
#undef RG_SKS_STEP
			static_assert(WS == 17, "the switch above lists the steps of a block of 17");
			{  // block complete: slot t becomes the minimum of entries t .. w-1 (slot 0 is nobody's S); the last entry is S[w-1] itself
				uint32_t run_hi = mix, run_lo = jb + WS - 1;
#pragma unroll
				for (int q = WS - 2; q >= 1; --q) {
					const unsigned long long v = win[q * 64];
					const bool t = (uint32_t)(v >> 32) <= run_hi;
					run_hi = t ? (uint32_t)(v >> 32) : run_hi;
					run_lo = t ? (uint32_t)v : run_lo;
					win[q * 64] = ((unsigned long long)run_hi << 32) | run_lo;
				}
			}
			jb += WS;
			i0 = 0;
			if (jb < nsteps) continue;
			final = true;
		flush:
			// assemble 64 queued records (all of them after the last step), lane per record
			while (qtail - qhead >= 64u || (final && qtail != qhead)) {
				const uint32_t cnt = (qtail - qhead < 64u) ? qtail - qhead : 64u;
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
				if (!FUSED && chunk_used + cnt > REC_CHUNK) {
					if (have_chunk && !dead)
						for (uint32_t q = chunk_used + lane; q < REC_CHUNK; q += 64) rec_mark_invalid(&recs[chunk_base + q]);
					unsigned long long nbase = 0;
					if (lane == 0) nbase = atomicAdd(&counters[PC_REC_CURSOR], (unsigned long long)REC_CHUNK);
					nbase = wave_first(nbase);
					if (nbase + REC_CHUNK > rec_cap) {
						dead = true;
						if (lane == 0) atomicAdd(&counters[PC_REC_OVF], 1ull);
					}
					chunk_base = nbase;
					chunk_used = 0;
					have_chunk = true;
				}
				const unsigned long long d = ring[(qhead + lane) & (SKL_RING - 1)];
				const uint32_t dl = (uint32_t)d;
				const uint32_t pr = dl & ((1u << SB) - 1u), len = ((dl >> SB) & ((1u << LNB) - 1u)) + 1u, src = (dl >> LANE_SH) & 63u, jrel = (dl >> MO_SH) & ((1u << LNB) - 1u);
				const uint32_t Ls = (uint32_t)__shfl((int)L, (int)src);
				const uint64_t b0 = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(bit0 >> 32), (int)src) << 32) | (uint32_t)__shfl((int)(uint32_t)bit0, (int)src);
				if (!dead && lane < cnt) {
					const uint32_t ns = Ls - (uint32_t)k + 1;
					const uint32_t hasL = pr > 0, hasR = (pr + len) < ns;
					const uint32_t first = pr - hasL;  // read position of the record's base 0
					const uint32_t nbases = hasL + len + (uint32_t)k - 1 + hasR;
					const uint64_t bit = b0 + 2ull * first;
					const uint32_t *__restrict__ wp = P32 + (bit >> 5);
					const uint32_t sf = (uint32_t)bit & 31u;
					const uint32_t ndw = (sf + 2u * nbases + 31u) >> 5;  // dwords that hold the record's bases
					uint32_t a[2 * NB + 1];
#pragma unroll
					for (int q = 0; q < 2 * NB + 1; ++q) a[q] = ((uint32_t)q < ndw) ? wp[q] : 0u;
					Rec<1> rec;
#pragma unroll
					for (int q = 0; q < NB; ++q) {
						uint64_t word = 0;
						if (nbases > 32u * q) {
							word = (uint64_t)__builtin_amdgcn_alignbit(a[2 * q + 1], a[2 * q], sf) |
							       ((uint64_t)__builtin_amdgcn_alignbit(a[2 * q + 2], a[2 * q + 1], sf) << 32);
							const uint32_t rem = nbases - 32u * q;
							if (rem < 32u) word &= (1ull << (2u * rem)) - 1;
						}
						rec.w[q] = word;
					}
					const uint32_t moff = jrel + hasL;
					rec.w[NB] = (uint64_t)pr | ((uint64_t)(len - 1) << 16) | ((uint64_t)hasL << 22) | ((uint64_t)hasR << 23) |
					            ((uint64_t)moff << 24) | ((read_id_base + rb + src) << 31);
					const RM packed = rec_pack(rec, (RM *)nullptr);
					const uint32_t bk = bucket_number(order_unmix((uint32_t)(d >> 32)));
					if constexpr (FUSED) {
						const uint32_t dg = bk >> (BUCKET_TOP - fbits);
						const uint32_t at = atomicAdd(&s_hist[dg], 1u);  // the place in run (dg, this workgroup)
						if (at < run_cap) {
							const uint64_t idx = ((uint64_t)dg * gridDim.x + blockIdx.x) * run_cap + at;
							recs[idx] = packed;
							side[idx] = (uint8_t)(bk >> (BUCKET_TOP - fbits - 8));
						} else if (at == run_cap) {
							atomicAdd(&counters[PC_REC_OVF], 1ull);
						}
					} else {
						recs[chunk_base + chunk_used + lane] = packed;
						atomicAdd(&s_hist[modulo ? bk % modulo : bk >> hist_shift], 1u);
					}
					sample_record<RM>(packed, bk, sample, sample_cap, counters);
				}
				chunk_used += cnt;
				qhead += cnt;
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
			}
			if (final) break;
			if (i0 >= WS) {  // left at the block's last step: the block is complete
				i0 = WS;     // (the switch falls to its default and the suffix pass runs)
			}
		}
	}
	if (!FUSED && have_chunk && !dead)
		for (uint32_t q = chunk_used + lane; q < REC_CHUNK; q += 64) rec_mark_invalid(&recs[chunk_base + q]);
#pragma unroll
	for (int d = 32; d; d >>= 1) occ += __shfl_xor(occ, d);
	if (lane == 0 && occ) atomicAdd(&counters[2 /*C_OCC*/], occ);
	__syncthreads();
	if constexpr (FUSED) {
		if (threadIdx.x < (1u << fbits)) run_count[threadIdx.x * gridDim.x + blockIdx.x] = s_hist[threadIdx.x] < run_cap ? s_hist[threadIdx.x] : run_cap;
	} else {
		if (s_hist[threadIdx.x]) atomicAdd(&coarse_hist[threadIdx.x], (unsigned long long)s_hist[threadIdx.x]);
	}
}

// ------------------------------------------------------------------------------------------------
// One partition level.  The input is cut into S segments (seg_begin[S+1]; level 1: the single segment
// [0, n_in) of k_superkmers output, holes included).  A tile of SCATTER_TILE records never crosses a
// segment (tile_first[s] = first tile of segment s).  digit = (bucket number >> shift) & (2^dbits - 1);
// the records of segment s with digit d go to the range that starts at cursors[(s << dbits) + d].
// 2560 records of 24 bytes + digits = 72 KB: still two workgroups per CU, runs a quarter longer than with 2048 (k_scatter x3
// 21.9-23.1 -> 21.0-22.3 ms on C2, same box; 3072 would leave one workgroup per CU)
static constexpr int SCATTER_TILE = 2560;
static constexpr int SCATTER_TILE_C = 3072;  // 20-byte records: the same 60 KB
// (two-word keys: 2048 records of 40 bytes, one workgroup of 1024 threads per CU -- 7.4 ms for the three levels on C3's
// shape against 7.6 with 2560 records and 512 threads)
template <int W, bool C>
constexpr int scatter_tile() { return (W == 1 && C) ? SCATTER_TILE_C : (W == 2 ? 2048 : SCATTER_TILE); }
// threads of a k_scatter workgroup.  The 20-byte tile with 1024 (three records each, 32 wavefronts per CU): the three levels
// on C2 19.8 -> 18.4 ms on a box where 512 threads are slow (boxes differ: 18.3-19.8 with 512), 768: 18.45
template <int W, bool C>
constexpr int scatter_threads() { return (W == 1 && C) || W == 2 ? 1024 : 512; }
static constexpr int SCATTER_THREADS = 512;  // threads of a scatter workgroup: 4 records each (256 threads x 8: 23-25 ms for the three levels on C2, 512 x 4: 20.5-22)

__device__ __forceinline__ uint32_t find_segment(const uint32_t *__restrict__ tile_first, uint32_t S, uint32_t tile) {
	uint32_t lo = 0, hi = S;  // largest s with tile_first[s] <= tile
	while (hi - lo > 1) {
		uint32_t mid = (lo + hi) >> 1;
		if (tile_first[mid] <= tile) lo = mid;
		else hi = mid;
	}
	return lo;
}

// TAG: 1 for the launches of the record sample's own small sweep, so that a profiler's per-kernel averages of the real
// partition levels are not diluted by them
template <int W, int TAG = 0, bool C = false>
__global__ __launch_bounds__((scatter_threads<W, C>())) void k_scatter(const RecMem<W, C> *__restrict__ in, int m, uint32_t S,
                                                 const unsigned long long *__restrict__ seg_begin,
                                                 const uint32_t *__restrict__ tile_first, int shift, int dbits,
                                                 unsigned long long *__restrict__ cursors, RecMem<W, C> *__restrict__ out, uint32_t modulo,
                                                 uint8_t *__restrict__ side, int next_shift, uint32_t next_mask,
                                                 const unsigned long long *__restrict__ seg_end = nullptr, uint32_t seg_div = 1) {
	// seg_end / seg_div: the input is the runs the record kernel left when it folded the first level in -- segment s is filled
	// up to seg_end[s] only, and seg_div consecutive segments (the runs of one digit, one per workgroup of that kernel) share
	// the cursors of output segment s / seg_div
	using RM = RecMem<W, C>;
	constexpr int SCATTER_TILE = scatter_tile<W, C>();
	constexpr int SCATTER_THREADS = scatter_threads<W, C>();
	static_assert(SCATTER_TILE % SCATTER_THREADS == 0, "whole records per thread");
	// modulo != 0: the digit is bucket % modulo (destination rank of the record exchange) instead of bits of the bucket
	// side != null: side[i] = the NEXT level's digit of out[i], so that the next level's histogram reads one byte per
	// record instead of the record (k_level_hist_side)
	constexpr int PER = SCATTER_TILE / SCATTER_THREADS;
	__shared__ RM s_rec[SCATTER_TILE];
	__shared__ uint8_t s_dig[SCATTER_TILE], s_next[SCATTER_TILE];
	__shared__ uint32_t s_cnt[256], s_off[256];
	__shared__ unsigned long long s_gbase[256];
	__shared__ uint32_t s_scan[256];
	const uint32_t seg = find_segment(tile_first, S, blockIdx.x);
	const uint64_t begin = seg_begin[seg] + (uint64_t)(blockIdx.x - tile_first[seg]) * SCATTER_TILE;
	const uint64_t send = seg_end ? seg_end[seg] : seg_begin[seg + 1];
	const uint64_t end = begin + SCATTER_TILE < send ? begin + SCATTER_TILE : send;
	const uint32_t oseg = seg / seg_div;
	const uint32_t dmask = (1u << dbits) - 1;
	if (threadIdx.x < 256) s_cnt[threadIdx.x] = 0;
	__syncthreads();
	RM mine[PER];
	uint32_t dig[PER], rank[PER], nxt[PER];
#pragma unroll
	for (int j = 0; j < PER; ++j) {
		uint64_t i = begin + threadIdx.x + (uint64_t)j * SCATTER_THREADS;
		dig[j] = 0xFFFFFFFFu;
		rank[j] = 0;
		nxt[j] = 0;
		if (i < end) {
			mine[j] = in[i];
			if (rec_valid(mine[j])) {
				const uint32_t bk = rec_bucket<W>(mine[j], m);
				nxt[j] = (bk >> next_shift) & next_mask;
				dig[j] = modulo ? bk % modulo : (bk >> shift) & dmask;
				rank[j] = atomicAdd(&s_cnt[dig[j]], 1u);
			}
		}
	}
	__syncthreads();
	// exclusive scan of the 256 digit counts (shuffles inside each wavefront, one barrier across the four) and one
	// cursor claim per non-empty digit
	{
		const bool low = threadIdx.x < 256;  // one thread per digit
		const uint32_t c = low ? s_cnt[threadIdx.x] : 0u;
		const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
		uint32_t incl = c;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t o = __shfl_up(incl, d);
			if (lane >= (uint32_t)d) incl += o;
		}
		if (lane == 63 && low) s_scan[wv] = incl;
		__syncthreads();
		uint32_t before = 0;
#pragma unroll
		for (uint32_t q = 0; q < 4; ++q)
			if (q < wv) before += s_scan[q];
		if (low) {
			s_off[threadIdx.x] = before + incl - c;
			if (threadIdx.x == 255) s_scan[4] = before + incl;
			unsigned long long g = 0;
			if (c) g = atomicAdd(&cursors[((uint64_t)oseg << dbits) + threadIdx.x], (unsigned long long)c);
			s_gbase[threadIdx.x] = g;
		}
	}
	__syncthreads();
	const uint32_t total = s_scan[4];
#pragma unroll
	for (int j = 0; j < PER; ++j)
		if (dig[j] != 0xFFFFFFFFu) {
			uint32_t pos = s_off[dig[j]] + rank[j];
			s_rec[pos] = mine[j];
			s_dig[pos] = (uint8_t)dig[j];
			s_next[pos] = (uint8_t)nxt[j];
		}
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < total; i += SCATTER_THREADS) {
		uint32_t d = s_dig[i];
		const unsigned long long dst = s_gbase[d] + (i - s_off[d]);
		out[dst] = s_rec[i];
		if (side) side[dst] = s_next[i];
	}
}

// digit histogram of one level from the side bytes the previous level's scatter wrote: hist[(seg << dbits) + digit]
// (TILE = the tile of the scatter kernel that follows: SCATTER_TILE records, or SCATTER64_TILE Bloom items)
static constexpr uint32_t HIST_SIDE_TILES = 8;  // tiles a workgroup of k_level_hist_side walks (one tile each: 222 K workgroups of 2.5 KB on C2, 1.15 ms for the five launches of a step; 8: 0.58 ms; 32: 0.64)
template <int TILE>
__global__ __launch_bounds__(256) void k_level_hist_side(const uint8_t *__restrict__ side, uint32_t S,
                                                         const unsigned long long *__restrict__ seg_begin,
                                                         const uint32_t *__restrict__ tile_first, int dbits,
                                                         unsigned long long *__restrict__ hist, uint32_t ntiles,
                                                         const unsigned long long *__restrict__ seg_end = nullptr, uint32_t seg_div = 1,
                                                         int side_shift = 0) {
	// seg_end: segments that do not touch (ranges sized in advance and filled in part); otherwise segment s ends where s + 1 begins
	// seg_div, side_shift: the runs of the record kernel's folded first level (see k_scatter) -- seg_div consecutive segments
	// count into one histogram segment, and the byte holds 8 bucket bits of which the level takes the top dbits
	__shared__ uint32_t s_cnt[256];
	const uint32_t t0 = blockIdx.x * HIST_SIDE_TILES;
	const uint32_t t1 = t0 + HIST_SIDE_TILES < ntiles ? t0 + HIST_SIDE_TILES : ntiles;
	uint32_t seg = find_segment(tile_first, S, t0), cur = seg / seg_div;
	s_cnt[threadIdx.x] = 0;
	__syncthreads();
	for (uint32_t t = t0; t < t1; ++t) {
		while (t >= tile_first[seg + 1]) ++seg;  // tile_first[S] = ntiles > t: ends; segments without tiles are passed over
		if (seg / seg_div != cur) {  // the counts so far belong to the segment left behind
			__syncthreads();
			if (threadIdx.x < (1u << dbits) && s_cnt[threadIdx.x]) atomicAdd(&hist[((uint64_t)cur << dbits) + threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
			s_cnt[threadIdx.x] = 0;
			__syncthreads();
			cur = seg / seg_div;
		}
		const uint64_t begin = seg_begin[seg] + (uint64_t)(t - tile_first[seg]) * TILE;
		const uint64_t send = seg_end ? seg_end[seg] : seg_begin[seg + 1];
		const uint64_t end = begin + TILE < send ? begin + TILE : send;
#pragma unroll
		for (int r = 0; r < (TILE + 2047) / 2048; ++r) {  // eight bytes per thread in one load (any alignment), the last ones of a segment one by one
			const uint64_t i = begin + (uint64_t)r * 2048 + (uint64_t)threadIdx.x * 8;
			if (i + 8 <= end) {
				uint64_t v;
				__builtin_memcpy(&v, side + i, 8);
#pragma unroll
				for (int b = 0; b < 8; ++b) atomicAdd(&s_cnt[((uint32_t)(v >> (8 * b)) & 255u) >> side_shift], 1u);
			} else {
				for (uint64_t j = i; j < end; ++j) atomicAdd(&s_cnt[(uint32_t)side[j] >> side_shift], 1u);
			}
		}
	}
	__syncthreads();
	if (threadIdx.x < (1u << dbits) && s_cnt[threadIdx.x])
		atomicAdd(&hist[((uint64_t)cur << dbits) + threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
}

// digit histogram of one level: hist[(seg << dbits) + digit]
template <int W, bool C = false>
__global__ __launch_bounds__(256) void k_level_hist(const RecMem<W, C> *__restrict__ in, int m, uint32_t S,
                                                    const unsigned long long *__restrict__ seg_begin,
                                                    const uint32_t *__restrict__ tile_first, int shift, int dbits,
                                                    unsigned long long *__restrict__ hist, RecMem<W, C> *__restrict__ sample = nullptr,
                                                    uint64_t sample_cap = 0, unsigned long long *__restrict__ counters = nullptr) {
	using RM = RecMem<W, C>;
	constexpr int SCATTER_TILE = scatter_tile<W, C>();
	__shared__ uint32_t s_cnt[256];
	const uint32_t seg = find_segment(tile_first, S, blockIdx.x);
	const uint64_t begin = seg_begin[seg] + (uint64_t)(blockIdx.x - tile_first[seg]) * SCATTER_TILE;
	const uint64_t end = begin + SCATTER_TILE < seg_begin[seg + 1] ? begin + SCATTER_TILE : seg_begin[seg + 1];
	const uint32_t dmask = (1u << dbits) - 1;
	s_cnt[threadIdx.x] = 0;
	__syncthreads();
	for (uint64_t i = begin + threadIdx.x; i < end; i += 256) {
		const RM r = in[i];
		if (rec_valid(r)) {
			const uint32_t bk = rec_bucket<W>(r, m);
			atomicAdd(&s_cnt[(bk >> shift) & dmask], 1u);
			sample_record<RM>(r, bk, sample, sample_cap, counters);
		}
	}
	__syncthreads();
	if (threadIdx.x <= dmask && s_cnt[threadIdx.x])
		atomicAdd(&hist[((uint64_t)seg << dbits) + threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
}

// exclusive scan of n counters, three small kernels (block sums, scan of the sums, add back)
static constexpr int SCAN_PER_BLOCK = 2048;
__global__ __launch_bounds__(256) void k_scan_local(const unsigned long long *__restrict__ in, unsigned long long *__restrict__ out,
                                                    uint64_t n, unsigned long long *__restrict__ block_sums) {
	__shared__ unsigned long long s_t[256];
	const uint64_t base = (uint64_t)blockIdx.x * SCAN_PER_BLOCK + (uint64_t)threadIdx.x * 8;
	unsigned long long v[8], sum = 0;
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		v[j] = (base + j < n) ? in[base + j] : 0;
		sum += v[j];
	}
	s_t[threadIdx.x] = sum;
	__syncthreads();
	for (int d = 1; d < 256; d <<= 1) {
		unsigned long long t = (threadIdx.x >= (uint32_t)d) ? s_t[threadIdx.x - d] : 0;
		__syncthreads();
		s_t[threadIdx.x] += t;
		__syncthreads();
	}
	unsigned long long run = s_t[threadIdx.x] - sum;
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		if (base + j < n) out[base + j] = run;
		run += v[j];
	}
	if (threadIdx.x == 255) block_sums[blockIdx.x] = s_t[255];
}
// one wavefront (launched as <<<1, 64>>>): every lane sums a contiguous share of the block sums, the shares are scanned with
// shuffles, every lane writes its share back (one thread walking 2048 sums took 0.24 ms)
__global__ __launch_bounds__(64) void k_scan_sums(unsigned long long *__restrict__ block_sums, uint32_t nblocks, unsigned long long *__restrict__ total) {
	if (blockIdx.x != 0) return;
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t per = (nblocks + 63u) / 64u;
	const uint32_t lo = lane * per < nblocks ? lane * per : nblocks;
	const uint32_t hi = lo + per < nblocks ? lo + per : nblocks;
	unsigned long long sum = 0;
	for (uint32_t i = lo; i < hi; ++i) sum += block_sums[i];
	unsigned long long incl = sum;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const unsigned long long o = __shfl_up(incl, d);
		if (lane >= (uint32_t)d) incl += o;
	}
	unsigned long long run = incl - sum;
	for (uint32_t i = lo; i < hi; ++i) {
		const unsigned long long t = block_sums[i];
		block_sums[i] = run;
		run += t;
	}
	if (lane == 63) *total = incl;
}
__global__ __launch_bounds__(256) void k_scan_add(unsigned long long *__restrict__ out, uint64_t n,
                                                  const unsigned long long *__restrict__ block_sums,
                                                  const unsigned long long *__restrict__ total,
                                                  unsigned long long *__restrict__ copy) {
	const uint64_t base = (uint64_t)blockIdx.x * SCAN_PER_BLOCK;
	const unsigned long long add = block_sums[blockIdx.x];
	for (uint64_t i = base + threadIdx.x; i < base + SCAN_PER_BLOCK && i < n; i += 256) {
		unsigned long long v = out[i] + add;
		out[i] = v;
		if (copy) copy[i] = v;
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = *total;
}

// ------------------------------------------------------------------------------------------------
// One *wavefront* per bucket at a time (a bucket is a few hundred records; thousands of independent
// wavefronts keep enough loads in flight, and a wavefront needs no s_barrier).  Two LDS hash tables
// per wavefront:
//  1. a table of distinct *records*.  At coverage c the same super-k-mer arrives from ~c reads as
//     bit-identical records (apart from read id and position), so records are merged first: key =
//     bases + length + neighbour flags, value = multiplicity and smallest (read id, position).  Equal
//     records produce equal k-mers with equal edge bits, and the smallest stamp of k-mer i over them is
//     (smallest record stamp) + i, so expanding one representative with a weight is exact.
//  2. a table of distinct k-mers, fed by expanding every distinct record once (its k-mers laid out
//     flat, a lane per k-mer), then appended to the log.
// Slots claimed since the last flush are remembered in lists, so a flush costs O(distinct keys).  Log
// space is reserved LOG_CHUNK entries at a time per wavefront; unused entries of a chunk are EMPTY.
// Neither table can overfill: their fill is checked at points between which at most 64 keys arrive
// (the spill path of the k-mer table is for safety only).  The records of the next round (or of the
// next bucket's first round) are loaded before the current round is processed.
// The item of the Bloom admission's key partition (rg_bloomx.h): where a key's eight filter bits start from (o = hash2 & mask),
// where the key sits (slot of the table, or position in the log) and when it first arrived.
struct Item16 {
	unsigned long long a;  // o << 31 | low 31 bits of the slot index
	unsigned long long b;  // arrival stamp of the key (49 bits) | bit 31 of the slot index << 63: key arrays of up to 2^32 slots
};
template <int W>
__device__ __forceinline__ uint64_t xor_origin(uint64_t k0, uint64_t k1, uint64_t m_mask) {
	uint64_t K0, K1;
	from_internal<W>(k0, k1, K0, K1);
	return hash2<W>(K0, K1) & m_mask;
}
// k_aggregate can make those items itself, while an entry is on its way from LDS to the log (round 3): the admission then has
// no first pass over the log (k_xor_keys: 32 bytes read per entry to make 16).  They go to 32 runs per workgroup by the top
// 5 bits of o -- run (d, g) = slice [(d * n_wg + g) * run_cap, ...) -- like the records of the folded first level (see
// k_superkmers_lanes_s), with the next 8 bits of o in a byte beside them; run_count carries the cursors from launch to launch.
// items == nullptr: nothing of this happens.
struct AggItems {
	Item16 *items;
	uint8_t *side;
	uint32_t *run_count;
	uint32_t run_cap, n_wg;
	uint64_t m_mask;
	int shift1;  // log2(filter bits) - 5
	// one-word keys only: the log holds 16-byte entries (key, meta) -- the item carries the stamp, which nothing else reads
	// between here and the table when the admission starts from the items (half the bytes of the log, written and read)
	uint32_t compact;
};
// a compact log's entry (Ent16 of rg_table.h: key, meta)
__device__ __forceinline__ void log_store(Slot *__restrict__ log, unsigned long long dst, const Slot &e, bool compact) {
	if (compact) *reinterpret_cast<ulonglong2 *>(reinterpret_cast<uint64_t *>(log) + 2 * dst) = make_ulonglong2(e.k0, e.meta);
	else log[dst] = e;
}
static constexpr uint32_t AGG_ITEM_RUNS = 32;

static constexpr uint32_t AGG_SLOTS = 256;      // the k-mer table of a wavefront ...
static constexpr uint32_t AGG_SLOTS_BIG = 512;  // ... and of the variant for diverse inputs (the sample says a bucket holds about 100 k-mers or more even at 2^24 buckets:
                                                // reads with 1 % errors at 30x and beyond), two workgroups per CU instead of three, no bucket retried in two halves
// The k-mer table takes up to 64 new keys per round of the expansion (one lane per k-mer), so a round may only start
// with AGG_SLOTS - 64 slots claimed at most: a batch of records whose k-mers fit below AGG_LIMIT all together runs
// unchecked, otherwise the fill is looked at after every round and the table is declared too small for the (sub-pass
// of the) bucket above AGG_FLUSH.  With that no k-mer can fail to find a slot; should one all the same, the attempt
// is dropped like an overfull one (only the last-resort mode of partial flushes writes such a k-mer to the log on
// its own: a k-mer written before the attempt is dropped would be counted again by the retry).
static constexpr uint32_t AGG_LIMIT = 190;      // of 256 slots; scaled with the table (agg_limit<SLOTS>())
template <int SLOTS>
constexpr uint32_t agg_limit() { return AGG_LIMIT * (uint32_t)SLOTS / AGG_SLOTS; }
static constexpr uint32_t AGG_TARGET = 64;  // distinct k-mers per bucket the partition is sized for from the sample (C2 at 2^22 buckets: 24; the tail must stay below AGG_LIMIT)
static constexpr uint32_t AGG_FLUSH = 190;
static constexpr uint32_t AGG_BIG_ABOVE = 80;   // expected k-mers per bucket above which the sweep takes the AGG_SLOTS_BIG tables
static constexpr uint32_t RT_SLOTS = 128;
// distinct records the table may hold when a round of up to 64 new ones arrives: at most 15/16 full after it.  With 3/4 (32)
// a bucket of C2, some 31 distinct records, was often expanded twice, the first time for a few records: 13.5 -> 12.6 ms
// (7/8: 12.7)
static constexpr uint32_t RT_FLUSH = RT_SLOTS * 15 / 16 - 64;
static constexpr uint32_t LOG_CHUNK = 512;
static constexpr int RT_TAG_SHIFT = 48;  // RecTable: where the last word of a record carries the header bits
// wavefronts per workgroup (one per workgroup: 26 ms instead of 17).  Two-word keys have 16.1 KB of tables per wavefront,
// 8 wavefronts per CU with four per workgroup; three per workgroup (9 per CU) changed nothing (C3 shape: 25.4 ms both), five
// (80.7 KB: only one workgroup fits) 40.9 ms
template <int W>
constexpr uint32_t agg_waves() { return 4u; }

template <int W, int SLOTS = AGG_SLOTS>
struct AggTable {
	uint64_t k0[SLOTS];
	uint64_t k1[W == 2 ? SLOTS : 1];
	uint64_t stamp[SLOTS];
	uint32_t cnt[SLOTS];
	uint32_t edges[SLOTS];
	uint16_t list[SLOTS];  // the slots claimed since the last flush (their number: a register of the wavefront)
};

template <int W>
struct RecTable {
	unsigned long long fp[RT_SLOTS];  // 0 = free, else a 63-bit fingerprint of the key | 1
	uint64_t w[2 * W][RT_SLOTS];  // the record's bases; the top 16 bits of the last word hold header bits 16..30 (length, neighbour flags,
	                              // minimizer offset): a record has at most 2k - 13 <= 113 bases, so its last word has 26 bits or more to spare,
	                              // and what a slot is compared by and expanded from is one LDS access less
	uint64_t stamp[RT_SLOTS];
	uint32_t cnt[RT_SLOTS];
	uint16_t list[RT_SLOTS];
};

__device__ __forceinline__ void wave_sync() {  // orders this wavefront's LDS traffic; no s_barrier
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// inclusive scans over the 64 lanes with DPP row shifts and row broadcasts (VALU only: no trip through the LDS crossbar,
// where a __shfl_up ladder makes six dependent ones)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v) {
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v) {
	v += dpp_or_zero<0x111, 0xf>(v);  // row_shr:1
	v += dpp_or_zero<0x112, 0xf>(v);  // row_shr:2
	v += dpp_or_zero<0x114, 0xf>(v);  // row_shr:4
	v += dpp_or_zero<0x118, 0xf>(v);  // row_shr:8
	v += dpp_or_zero<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
	v += dpp_or_zero<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
	return v;
}
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t v) {
	v = max(v, dpp_or_zero<0x111, 0xf>(v));
	v = max(v, dpp_or_zero<0x112, 0xf>(v));
	v = max(v, dpp_or_zero<0x114, 0xf>(v));
	v = max(v, dpp_or_zero<0x118, 0xf>(v));
	v = max(v, dpp_or_zero<0x142, 0xa>(v));
	v = max(v, dpp_or_zero<0x143, 0xc>(v));
	return v;
}

template <int W>
__device__ __forceinline__ uint32_t agg_hash(uint64_t i0, uint64_t i1) {
	uint32_t x = (uint32_t)i0 ^ __builtin_amdgcn_alignbit((uint32_t)(i0 >> 32), (uint32_t)(i0 >> 32), 19);
	if (W == 2) x ^= __builtin_amdgcn_alignbit((uint32_t)i1, (uint32_t)i1, 7) ^ __builtin_amdgcn_alignbit((uint32_t)(i1 >> 32), (uint32_t)(i1 >> 32), 25);
	x ^= x >> 15;
	return x * 0x9E3779B1u;  // top 8 bits: slot of the LDS table; the 12 bits below: sub-pass of the bucket
}
static_assert(AGG_SLOTS == 256 && AGG_SLOTS_BIG == 512, "the slot is the top 8 or 9 bits of agg_hash");
static constexpr uint32_t AGG_MAX_SUBPASSES = 4096;

template <int W, int SLOTS>
__device__ __forceinline__ int lds_upsert(AggTable<W, SLOTS> &t, uint32_t hash, uint64_t i0, uint64_t i1, uint64_t stamp, uint32_t emask,
                                          uint32_t weight, bool &fresh) {  // -> the k-mer's slot, or -1: no slot left; fresh: the slot became this key's here
	// The probe is the compare-and-swap itself: it returns the key word that holds the slot -- one LDS round trip per probe,
	// where a read in front made two of every claim.  The caller lists and counts the slots its lanes made theirs after the
	// lanes have met again (ballot + rank on a fill kept in a scalar register): nothing here reads a counter back.
	// Exactly one lane is `fresh` per slot: the one whose compare-and-swap set the key's last word.
	uint32_t s = hash >> (SLOTS == 512 ? 23 : 24);
	fresh = false;
	for (uint32_t probe = 0; probe < (uint32_t)SLOTS; ++probe) {
		const uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long *>(&t.k0[s]), EMPTY, i0);
		if (prev == EMPTY || prev == i0) {
			bool match = true, set_last = prev == EMPTY;
			if (W == 2) {
				const uint64_t p1 = atomicCAS(reinterpret_cast<unsigned long long *>(&t.k1[s]), EMPTY, i1);
				set_last = p1 == EMPTY;
				match = set_last || p1 == i1;
			}
			if (match) {
				fresh = set_last;
				atomicMin(reinterpret_cast<unsigned long long *>(&t.stamp[s]), stamp);
				atomicAdd(&t.cnt[s], weight);
				atomicOr(&t.edges[s], emask);
				return (int)s;
			}
		}
		s = (s + 1) & ((uint32_t)SLOTS - 1);
	}
	return -1;
}

// Anchors: the records that a read end cut short are the many that nothing merges -- no other read repeats them -- and
// with long windows (k = 63: w = 49) nearly every k-mer is expanded on their account (C3's shape: 18 of the 25 ms of
// k_aggregate, measured by leaving them out).  But such a record is a stretch of the complete super-k-mer that other
// reads of the same locus and strand brought whole, aligned at the minimizer.  So a bucket is walked twice: first the
// records that no read end cut, merged and expanded as ever -- and the first ANCHORS distinct ones of them remembered with
// the table slot of each of their k-mers; then the cut ones: one that lies inside an anchor, base for base, adds its
// multiplicity and its arrival stamps to the anchor's slots (two LDS atomics per k-mer: no extraction, reverse complement,
// hash or probe; its edge bits are a subset of the anchor's), any other goes the usual way.
// Two-word keys only.  With one-word keys (w <= 17) a cut record holds five k-mers on average, and finding its anchor costs
// more than expanding them: built and measured on C2 with 16 anchors per wavefront in the 1 KB of LDS that packing the
// table's edge bytes four to a word frees (three workgroups per CU kept), all 8.0e7 cut records anchored: k_aggregate
// 13.5 -> 18.0 ms (and 13.7 with the packed bytes alone, so they went again too).
template <int W>
struct AnchorTable {
	static constexpr uint32_t N = 32;
	static constexpr uint32_t KMERS = 52;  // >= the longest window (w <= 49), a multiple of 4
	uint64_t w[2 * W][N];
	uint32_t mm[N];    // the minimizer as it stands in the record (not canonical)
	uint16_t tag[N];   // as RecTable's
	uint8_t kslot[N][KMERS];
	uint32_t fill;
};
template <int W, bool SLOW, int SLOTS>
constexpr bool agg_anchors() { return W == 2 && !SLOW && SLOTS == 256; }  // (the slot index is a byte; the 512-slot tables are for inputs whose records rarely repeat)

// SLOW == false: every bucket once, no sub-passes; a bucket whose k-mers do not fit the table is dropped unwritten
// and its number appended to retry_list (counters[PC_AGG_RETRY] of them).  SLOW == true: the buckets of
// bucket_list, with sub-passes.  Two instantiations so that the common path carries none of the retry machinery.
template <int W, bool SLOW, int SLOTS = AGG_SLOTS, bool C = false>
__global__ __launch_bounds__(64 * agg_waves<W>()) void k_aggregate(const RecMem<W, C> *__restrict__ recs,
                                                              const unsigned long long *__restrict__ bucket_begin, uint32_t nbuckets_max,
                                                              int k, Slot *__restrict__ log, uint64_t log_cap,
                                                              unsigned long long *__restrict__ counters, uint32_t flag_unique,
                                                              const uint32_t *__restrict__ bucket_list, uint32_t *__restrict__ retry_list,
                                                              int m, uint32_t rbits, AggItems ai, const unsigned long long *__restrict__ list_len = nullptr) {
	// list_len (SLOW): the length of bucket_list is read from there (at most nbuckets_max) -- the launch that retries the
	// buckets another launch has just listed follows it without the host reading their number in between
	uint32_t nbuckets = nbuckets_max;
	if (SLOW && list_len) {
		const unsigned long long n = *list_len;
		nbuckets = n < nbuckets_max ? (uint32_t)n : nbuckets_max;
	}
	// flag_unique: the caller guarantees that nothing outside this launch's log carries the same keys on their way
	// to the same table; the entries are then marked LOG_UNIQUE (every k-mer lives in exactly one bucket, see
	// k_superkmers, and a bucket is flushed once per sub-pass, each sub-pass owning a disjoint share of the k-mers).
	// Sub-passes: when the distinct k-mers of a bucket do not fit the LDS table, nothing is flushed half-way (the
	// occurrences of a k-mer are spread over the whole bucket, so partial flushes would multiply the log); the
	// attempt is dropped and the bucket is redone in several passes over its records, each taking a share of them
	// (see the loop at the end).  The first rbits levels of that split take whole records, by the bits of the
	// minimizer's bucket number that the partition levels left unused (m = minimizer length): a record that is not
	// of the pass costs its load and nothing else; further levels take k-mers by bits of their hash.
	constexpr int NB = 2 * W;
	__shared__ AggTable<W, SLOTS> s_tables[agg_waves<W>()];
	constexpr uint32_t LIMIT = agg_limit<SLOTS>();
	__shared__ RecTable<W> s_rtables[agg_waves<W>()];
	// 20-byte records: a round's 64 records are loaded as 5 x 64 consecutive words (whole cache lines, where a lane loading
	// its own 20 bytes straddles them: 14.3 ms on C2 against 12.9 with the 24-byte records) and handed to their lanes through LDS
	// (also expand()'s hand-over of a batch of records to their k-mers' lanes: 64 four-word tuples and 64 marks)
	__shared__ __align__(16) uint32_t s_stage[agg_waves<W>()][320];
	constexpr bool ANCH = agg_anchors<W, SLOW, SLOTS>();
	constexpr uint32_t ANCHORS = AnchorTable<W>::N;
	__shared__ AnchorTable<W> s_anchors[ANCH ? agg_waves<W>() : 1];
	__shared__ uint32_t s_icur[AGG_ITEM_RUNS];  // cursors of this workgroup's item runs (AggItems)
	if (ai.items) {
		if (threadIdx.x < AGG_ITEM_RUNS) s_icur[threadIdx.x] = ai.run_count[threadIdx.x * ai.n_wg + blockIdx.x];
		__syncthreads();
	}
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	AggTable<W, SLOTS> &T = s_tables[wave];
	RecTable<W> &R = s_rtables[wave];
	AnchorTable<W> &A = s_anchors[ANCH ? wave : 0];
	const int mlen = m;
	const uint64_t mask0 = (2 * k >= 64) ? ~0ull : ((1ull << (2 * k)) - 1);
	const uint64_t mask1 = (k > 32) ? ((1ull << (2 * k - 64)) - 1) : 0ull;
	const unsigned long long NO_CHUNK = ~0ull;
	unsigned long long chunk_base = NO_CHUNK;  // wave-uniform
	RecMem<W, C> nxt = {};
	// records [first, end) of a round (at most 64) -> nxt: a lane's own record, or (C) words lane, lane + 64, ... of the round
	auto load_round = [&](uint64_t first, uint64_t end) {
		if constexpr (C) {
			const uint32_t *__restrict__ p = reinterpret_cast<const uint32_t *>(recs + first);
			const uint32_t nd = 5u * (uint32_t)(end - first < 64 ? end - first : 64);
#pragma unroll
			for (uint32_t j = 0; j < 5; ++j)
				if (lane + 64u * j < nd) nxt.w[j] = p[lane + 64u * j];
		} else {
			if (first + lane < end) nxt = recs[first + lane];
		}
	};
	auto take_round = [&]() -> Rec<W> {
		if constexpr (C) {
			uint32_t *st = s_stage[wave];
#pragma unroll
			for (uint32_t j = 0; j < 5; ++j) st[lane + 64u * j] = nxt.w[j];
			wave_sync();
			Rec20 c;
#pragma unroll
			for (uint32_t j = 0; j < 5; ++j) c.w[j] = st[5u * lane + j];
			wave_sync();
			return rec_unpack(c);
		} else {
			return nxt;
		}
	};
	uint32_t chunk_used = 0;
	unsigned long long keys_written = 0;
	uint32_t multi_written = 0;  // entries written with a count of two or more (what admission stores whatever the filter says)
	uint32_t nonunique_flushes = 0;
	uint32_t cut_anchored = 0, cut_expanded = 0;  // statistics (lane 0): records cut by a read end that lay inside an anchor / were expanded
	bool log_full = false;         // wave-uniform
	bool partial_flushes = false;  // last resort for a bucket that more than AGG_MAX_SUBPASSES passes cannot hold
	// what the two tables hold, and whether a k-mer of the current bucket went to the log on its own (table full): wave-uniform
	// registers, kept up by ballots where the lanes meet again -- no counter in LDS is read back on the way
	uint32_t tfill = 0, rfill = 0;
	bool tspilled = false;

	// flush() and expand() are wave-uniform: their conditions come from fill counters read after wave_sync().
	// flush: the k-mer table's entries continue the wavefront's current log chunk and spill over into a new one.
	auto flush = [&]() {
		const uint32_t n = tfill;
		if (n == 0) return;
		const uint64_t mark = (flag_unique && !partial_flushes && !tspilled) ? LOG_UNIQUE : 0ull;
		// what the host needs to know about the log: its number of entries, and whether all are unique keys (summed per
		// wavefront, added once at the end: the counters share a cache line with the log cursor)
		keys_written += n;
		if (!mark) nonunique_flushes++;
		const uint32_t room = (chunk_base == NO_CHUNK) ? 0u : LOG_CHUNK - chunk_used;
		const unsigned long long old_dst = chunk_base + chunk_used;
		unsigned long long new_dst = 0;
		if (n > room) {
			if (lane == 0) {
				new_dst = atomicAdd(&counters[PC_LOG_CURSOR], (unsigned long long)LOG_CHUNK);
				if (new_dst + LOG_CHUNK > log_cap) atomicAdd(&counters[PC_LOG_OVF], 1ull);
			}
			new_dst = wave_first(new_dst);
			if (new_dst + LOG_CHUNK > log_cap) log_full = true;  // the launch is void: this wavefront stops after this bucket
		}
		for (uint32_t i = lane; i < n; i += 64) {
			const uint32_t sl = T.list[i];
			const unsigned long long dst = (i < room) ? old_dst + i : new_dst + (i - room);
			if (dst < log_cap) {
				Slot e;
				e.k0 = T.k0[sl];
				e.k1 = (W == 2) ? T.k1[sl] : EMPTY;
				e.stamp = T.stamp[sl];
				e.meta = (uint64_t)T.cnt[sl] | ((uint64_t)(T.edges[sl] & 0xFFu) << PEND_SHIFT) | mark;
				log_store(log, dst, e, W == 1 && ai.compact);
				multi_written += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64((uint32_t)e.meta >= 2u));  // (the count is the low 32 bits here)
				if (ai.items) {
					const uint64_t o = xor_origin<W>(e.k0, e.k1, ai.m_mask);
					const uint32_t d = (uint32_t)(o >> ai.shift1);
					const uint32_t at = atomicAdd(&s_icur[d], 1u);
					if (at < ai.run_cap) {
						const uint64_t idx = ((uint64_t)d * ai.n_wg + blockIdx.x) * ai.run_cap + at;
						*reinterpret_cast<ulonglong2 *>(&ai.items[idx]) = make_ulonglong2((o << 31) | (dst & 0x7FFFFFFFull), e.stamp | ((dst >> 31) << 63));
						ai.side[idx] = (uint8_t)(o >> (ai.shift1 - 8));
					} else if (at == ai.run_cap) {
						atomicAdd(&counters[PC_ITEM_OVF], 1ull);
					}
				}
			}
			T.k0[sl] = EMPTY;
			if (W == 2) T.k1[sl] = EMPTY;
			T.stamp[sl] = INF_STAMP;
			T.cnt[sl] = 0;
			T.edges[sl] = 0;
		}
		if (n > room) {
			chunk_base = new_dst;
			chunk_used = n - room;
		} else {
			chunk_used += n;
		}
		tfill = 0;
		wave_sync();
	};

	// drop what both tables hold without writing anything (a bucket attempt that overflowed)
	auto discard = [&]() {
		const uint32_t n = tfill;
		for (uint32_t i = lane; i < n; i += 64) {
			const uint32_t sl = T.list[i];
			T.k0[sl] = EMPTY;
			if (W == 2) T.k1[sl] = EMPTY;
			T.stamp[sl] = INF_STAMP;
			T.cnt[sl] = 0;
			T.edges[sl] = 0;
		}
		const uint32_t nr = rfill;
		for (uint32_t i = lane; i < nr; i += 64) {
			const uint32_t sl = R.list[i];
			R.fp[sl] = 0;
			R.cnt[sl] = 0;
			R.stamp[sl] = INF_STAMP;
		}
		tfill = 0;
		tspilled = false;
		rfill = 0;
		wave_sync();
	};

	// expand: every distinct record of the record table goes through the k-mer table once, weighted; of its k-mers
	// only those of sub-pass `sub` (of `nsub`, a power of two) are taken.  Returns false when the k-mer table would
	// overfill: the caller then drops the attempt.
	// The k-mers of a batch of 64 distinct records are laid out flat, one *lane per k-mer*: records hold 1 to w k-mers
	// (the ones cut short by a read end are the many), so a lane per record walking its own k-mers would leave most
	// lanes idle for most of the w iterations.  Lane t of a round finds its record by a binary search over the prefix
	// sums of the lengths (held in registers, read through the lane crossbar) and takes its k-mer and the two
	// neighbour bases straight out of the record's words.
	// anchoring (ANCH): the records are whole super-k-mers; the first ANCHORS of the bucket become anchors
	auto expand = [&](uint32_t sub, uint32_t nsub, bool anchoring) -> bool {
		const uint32_t n = rfill;
		if (n == 0) return true;
		for (uint32_t e0 = 0; e0 < n; e0 += 64) {
			const uint32_t e = e0 + lane;
			uint32_t rlen = 0, rsl = 0, rweight = 0, ranch = ANCHORS;
			uint64_t rstamp = 0;
			if (e < n) {
				rsl = R.list[e];
				rlen = ((uint32_t)(R.w[NB - 1][rsl] >> RT_TAG_SHIFT) & 63u) + 1;
				rweight = R.cnt[rsl];
				rstamp = R.stamp[rsl];
				R.fp[rsl] = 0;  // the slot is free again (its words and tag stay readable until it is claimed anew)
				R.cnt[rsl] = 0;
				R.stamp[rsl] = INF_STAMP;
				if (ANCH && anchoring) {
					const uint32_t a = atomicAdd(&A.fill, 1u);
					if (a < ANCHORS) {
						ranch = a;
						uint64_t aw[NB];
#pragma unroll
						for (int q = 0; q < NB; ++q) aw[q] = R.w[q][rsl];
						const uint32_t tg = (uint32_t)(aw[NB - 1] >> RT_TAG_SHIFT);
						aw[NB - 1] &= (1ull << RT_TAG_SHIFT) - 1;
#pragma unroll
						for (int q = 0; q < NB; ++q) A.w[q][a] = aw[q];
						A.tag[a] = (uint16_t)tg;
						A.mm[a] = (uint32_t)(rec_ext<W>(aw, (tg >> 8) & 127u) & ((1ull << (2 * mlen)) - 1));
					}
				}
			}
			const uint32_t incl = wave_scan_add(rlen);  // inclusive prefix sums of the lengths
			const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
			const uint32_t rpre = incl - rlen;
			// what a k-mer's lane needs of its record, handed over through LDS in one 16-byte read: slot, first k-mer's number,
			// anchor, multiplicity, smallest stamp
			uint32_t *const st = s_stage[wave];
			{
				uint4 tp;
				tp.x = rsl | (ranch << 8) | (rpre << 16);
				tp.y = rweight;
				tp.z = (uint32_t)rstamp;
				tp.w = (uint32_t)(rstamp >> 32);
				reinterpret_cast<uint4 *>(st)[lane] = tp;
			}
			const bool checked = tfill + total > LIMIT;  // else no overfill possible
			for (uint32_t t0 = 0; t0 < total; t0 += 64) {
				const uint32_t t = t0 + lane;
				// owner of k-mer t = the last record that starts at or before it: the records that start inside this round mark
				// their first lane (LDS), a running maximum carries the marks to the lanes behind (DPP), the record that holds
				// t0 covers the lanes in front of the first mark.  (A binary search over the prefix sums through the lane
				// crossbar was six dependent LDS round trips per round.)
				st[256 + lane] = 0;
				const bool rvalid = e < n;
				if (rvalid && rpre >= t0 && rpre < t0 + 64u) st[256 + (rpre - t0)] = lane + 1u;
				const uint32_t own0 = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(rvalid && rpre <= t0));  // 1-based: the record that holds t0
				wave_sync();
				uint32_t own = max(wave_scan_max(st[256 + lane]), own0) - 1u;
				own = own < 63u ? own : 63u;  // (a round has a k-mer, so own0 >= 1; lanes past the last k-mer do nothing)
				const uint4 tp = reinterpret_cast<const uint4 *>(st)[own];
				const uint32_t sl = tp.x & 0xFFu;
				const uint32_t i = t - (tp.x >> 16);
				const uint32_t weight = tp.y;
				const uint32_t anch = ANCH ? ((tp.x >> 8) & 0xFFu) : ANCHORS;
				const uint64_t stamp0 = ((uint64_t)tp.w << 32) | tp.z;
				int kslot = 0;
				bool fresh = false, nospace = false;  // this lane's k-mer took a free slot / found none
				if (t < total) {
					uint64_t bw[NB];
#pragma unroll
					for (int q = 0; q < NB; ++q) bw[q] = R.w[q][sl];
					const uint32_t tag = (uint32_t)(bw[NB - 1] >> RT_TAG_SHIFT);
					bw[NB - 1] &= (1ull << RT_TAG_SHIFT) - 1;
					const uint32_t len = (tag & 63u) + 1, hasL = (tag >> 6) & 1u, hasR = (tag >> 7) & 1u;
					const uint32_t pos = hasL + i;  // the k-mer's first base in the record
					uint64_t f0, f1 = 0, r0, r1;
					f0 = rec_ext<W>(bw, pos) & mask0;
					if (W == 2 && k > 32) f1 = rec_ext<W>(bw, pos + 32) & mask1;
					revcomp<W>(f0, f1, k, r0, r1);
					const bool use_rc = kmer_less<W>(r0, r1, f0, f1);
					uint32_t em = 0;
					if (pos > 0) {  // a k-mer precedes this one in its read
						const uint32_t prev = rec_base<W>(bw, pos - 1);
						em = 1u << (use_rc ? 7u - prev : prev);
					}
					if (i + 1 < len || hasR) {
						const uint32_t nb = rec_base<W>(bw, pos + (uint32_t)k);
						em |= 1u << (use_rc ? 3u - nb : 4u + nb);
					}
					uint64_t i0, i1;
					to_internal<W>(use_rc ? r0 : f0, use_rc ? r1 : f1, i0, i1);
					const uint32_t hx = agg_hash<W>(i0, i1);
					if (SLOW && ((hx >> (SLOTS == 512 ? 11 : 12)) & (nsub - 1)) != sub) {
						// another sub-pass owns this k-mer
					} else if ((kslot = lds_upsert<W, SLOTS>(T, hx, i0, i1, stamp0 + i, em, weight, fresh)) < 0) {
						nospace = true;  // no slot left
						if (!partial_flushes) {
							// the attempt is void
						} else {
						// LDS table full: this k-mer goes to the log on its own
						unsigned long long g = atomicAdd(&counters[PC_LOG_CURSOR], 1ull);
						if (g < log_cap) {
							Slot en;
							en.k0 = i0;
							en.k1 = (W == 2) ? i1 : EMPTY;
							en.stamp = stamp0 + i;
							en.meta = (uint64_t)weight | ((uint64_t)em << PEND_SHIFT);
							log_store(log, g, en, W == 1 && ai.compact);  // (such a log is not used: the entry is not unique)
						} else {
							atomicAdd(&counters[PC_LOG_OVF], 1ull);
						}
						atomicAdd(&counters[PC_LDS_SPILL], 1ull);
						}
					} else {
						if constexpr (ANCH)
							if (anch < ANCHORS && i < AnchorTable<W>::KMERS) A.kslot[anch][i] = (uint8_t)kslot;  // (i < w <= 49 for records this library cut)
					}
				}
				{  // the slots this round made their keys': listed and counted where the lanes meet again
					const unsigned long long made = __builtin_amdgcn_ballot_w64(fresh);
					if (fresh) T.list[tfill + __builtin_amdgcn_mbcnt_hi((uint32_t)(made >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)made, 0u))] = (uint16_t)kslot;
					tfill += (uint32_t)__popcll(made);
				}
				if (checked) {
					if (__builtin_amdgcn_ballot_w64(nospace)) {
						if (!partial_flushes) return false;
						tspilled = true;
					}
					if (tfill > LIMIT) {
						if (!partial_flushes) return false;
						wave_sync();
						flush();
					}
				}
			}
			wave_sync();
		}
		rfill = 0;
		wave_sync();
		return true;
	};

	for (uint32_t i = lane; i < (uint32_t)SLOTS; i += 64) {
		T.k0[i] = EMPTY;
		if (W == 2) T.k1[i] = EMPTY;
		T.stamp[i] = INF_STAMP;
		T.cnt[i] = 0;
		T.edges[i] = 0;
	}
	for (uint32_t i = lane; i < RT_SLOTS; i += 64) {
		R.fp[i] = 0;
		R.cnt[i] = 0;
		R.stamp[i] = INF_STAMP;
	}
	wave_sync();

	const uint32_t stride = gridDim.x * agg_waves<W>();
	// nbuckets = number of buckets to visit: all of them, or (SLOW) the length of bucket_list
	uint32_t b = blockIdx.x * agg_waves<W>() + wave;
	uint64_t cb = 0, ce = 0;
	if (b < nbuckets) {
		const uint32_t bb = (SLOW && bucket_list) ? bucket_list[b] : b;  // no list: every bucket
		cb = bucket_begin[bb];
		ce = bucket_begin[bb + 1];
	}
	bool prefetched = false;  // nxt holds records [cb, cb+64) of the current bucket
	while (b < nbuckets && !log_full) {
		const uint32_t b2 = b + stride;  // the bucket after this one: its bounds are fetched now, used at the last round
		uint64_t nb_begin = 0, nb_end = 0;
		if (b2 < nbuckets) {
			const uint32_t bb = (SLOW && bucket_list) ? bucket_list[b2] : b2;
			nb_begin = bucket_begin[bb];
			nb_end = bucket_begin[bb + 1];
		}
		// one pass over the bucket's records for sub-pass `sub` of `nsub`; true = the k-mer table overflowed
		bool next_ready = false;  // nxt already holds the next bucket's first round
		// phase (ANCH only): 0 = the records no read end cut, which become anchors; 1 = the cut ones; 2 = all alike.
		// last: nothing of this bucket is read after this call (the next bucket's first round may be fetched)
		auto pass = [&](uint32_t sub, uint32_t nsub, int phase, bool last) -> bool {
			bool overflow = false;
			next_ready = false;
			if (ANCH && phase == 0) {
				if (lane == 0) A.fill = 0;
				wave_sync();
			}
			const uint32_t n_anch = (ANCH && phase == 1) ? min((uint32_t)__builtin_amdgcn_readfirstlane((int)A.fill), ANCHORS) : 0u;
			// record levels of the split (SLOW only), then hash levels
			const uint32_t rl = SLOW ? min(31u - (uint32_t)__clz((int)nsub), rbits) : 0u;
			const uint32_t rmask = (1u << rl) - 1u, rsub = sub & rmask;
			const uint32_t hsub = sub >> rl, hn = nsub >> rl;
			if (!prefetched) load_round(cb, ce);
			prefetched = false;
			for (uint64_t base = cb; base < ce && !overflow; base += 64) {
				const Rec<W> rec = take_round();
				bool pending = base + lane < ce;
				if (SLOW && rl && pending && (rec_bucket<W>(rec, m) & rmask) != rsub) pending = false;  // a record of another pass
				if (base + 64 < ce) {
					load_round(base + 64, ce);
				} else if (last && sub + 1 == nsub) {  // last round of the last sub-pass: the next bucket's first round
					load_round(nb_begin, nb_end);
					prefetched = true;
					next_ready = true;
				}
				uint64_t bw[NB], stamp0 = 0;
				unsigned long long fp = 0;
				uint32_t tag = 0, sl = 0;
				{
					const uint64_t hdr = rec.w[NB];
					tag = (uint32_t)(hdr >> 16) & 0x7FFFu;
					stamp0 = ((hdr >> 31) << 16) | ((uint32_t)hdr & 0xFFFFu);
#pragma unroll
					for (int i = 0; i < NB; ++i) bw[i] = rec.w[i];
				}
				if (ANCH && phase != 2) {
					const bool whole = ((tag >> 6) & 3u) == 3u;
					if (pending && whole != (phase == 0)) pending = false;  // the other phase's record
				}
				if (ANCH && phase == 1 && n_anch) {
					// a cut record inside an anchor, base for base: its k-mer j is the anchor's k-mer dd + j
					const uint32_t lenT = (tag & 63u) + 1, hasLT = (tag >> 6) & 1u, hasRT = (tag >> 7) & 1u, moffT = (tag >> 8) & 127u;
					const uint32_t nbT = hasLT + lenT + (uint32_t)k - 1 + hasRT;
					const uint32_t mmT = (uint32_t)(rec_ext<W>(bw, moffT) & ((1ull << (2 * mlen)) - 1));
					uint32_t found = ANCHORS, dd = 0, from = 0;
					bool searching = pending;
					while (__any(searching)) {
						// the next anchor with this record's minimizer (every lane reads the same words) ...
						uint32_t cand = ANCHORS;
						for (uint32_t a = 0; a < n_anch; ++a) {
							const uint32_t mma = A.mm[a];
							if (searching && cand == ANCHORS && a >= from && mma == mmT) cand = a;
						}
						if (searching && cand == ANCHORS) searching = false;
						// ... compared base for base, every lane against its own candidate
						if (searching) {
							const uint32_t tga = A.tag[cand];
							const uint32_t moffS = (tga >> 8) & 127u, lenS = (tga & 63u) + 1;
							const uint32_t delta = moffS - moffT;  // the record's base 0 is the anchor's base delta
							const uint32_t d1 = delta + hasLT;     // its first k-mer starts there; the anchor's k-mer 0 at base 1
							bool same = moffS >= moffT && d1 >= 1u && d1 - 1u + lenT <= lenS && lenS <= AnchorTable<W>::KMERS && delta + nbT <= lenS + (uint32_t)k + 1u;
							if (same) {
								uint64_t sw[NB];
#pragma unroll
								for (int q = 0; q < NB; ++q) sw[q] = A.w[q][cand];
#pragma unroll
								for (int q = 0; q < NB; ++q)
									if (nbT > 32u * q) {
										uint64_t x = rec_ext<W>(sw, delta + 32u * q);
										const uint32_t rem = nbT - 32u * q;
										if (rem < 32u) x &= (1ull << (2u * rem)) - 1;
										same = same && x == bw[q];
									}
							}
							if (same) {
								found = cand;
								dd = d1 - 1u;
								searching = false;
							} else {
								from = cand + 1;
							}
						}
					}
					const bool anchored = pending && found < ANCHORS;
					const uint32_t clen = anchored ? lenT : 0u;
					uint32_t incl = clen;
#pragma unroll
					for (int d = 1; d < 64; d <<= 1) {
						const uint32_t o = __shfl_up(incl, d);
						if (lane >= (uint32_t)d) incl += o;
					}
					const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
					const uint32_t cpre = incl - clen;
					for (uint32_t t0 = 0; t0 < total; t0 += 64) {  // a lane per k-mer, as in expand()
						const uint32_t t = t0 + lane;
						uint32_t own = 0;
#pragma unroll
						for (uint32_t st = 32; st; st >>= 1) {
							const uint32_t v = (uint32_t)__shfl((int)incl, (int)(own + st - 1));
							if (v <= t) own += st;
						}
						own = own < 63u ? own : 63u;
						const uint32_t a = (uint32_t)__shfl((int)found, (int)own);
						const uint32_t i = t - (uint32_t)__shfl((int)cpre, (int)own);
						const uint32_t d0 = (uint32_t)__shfl((int)dd, (int)own);
						const uint64_t st0 = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(stamp0 >> 32), (int)own) << 32) | (uint32_t)__shfl((int)(uint32_t)stamp0, (int)own);
						if (t < total) {
							const uint32_t ks = A.kslot[a][d0 + i];
							atomicAdd(&T.cnt[ks], 1u);
							atomicMin(reinterpret_cast<unsigned long long *>(&T.stamp[ks]), (unsigned long long)(st0 + i));
						}
					}
					if (anchored) pending = false;
					cut_anchored += (uint32_t)__popcll(__ballot(anchored));
					cut_expanded += (uint32_t)__popcll(__ballot(pending));
					if (total) wave_sync();
				}
				if (!ANCH || __any(pending)) {  // (a round of cut records that all lay inside anchors has nothing to insert)
					uint64_t h = (uint64_t)tag * 0x9E3779B97F4A7C15ull;
#pragma unroll
					for (int i = 0; i < NB; ++i) {
						h = (h ^ bw[i]) * 0xC2B2AE3D27D4EB4Full;
						h ^= h >> 29;
					}
					h *= 0xBF58476D1CE4E5B9ull;
					h ^= h >> 32;
					fp = h | 1ull;
					sl = (uint32_t)(h >> 40) & (RT_SLOTS - 1);
				}
				if (!ANCH || __any(pending))
				do {
					bool owner = false;
					if (pending) {
						for (;;) {  // ends: the table is never more than 15/16 full
							unsigned long long cur = R.fp[sl];
							if (cur == 0) {
								cur = atomicCAS(&R.fp[sl], 0ull, fp);
								if (cur == 0) {
									owner = true;
									break;
								}
							}
							if (cur == fp) break;
							sl = (sl + 1) & (RT_SLOTS - 1);
						}
					}
					{  // the lanes that claimed a slot number themselves (ballot + rank on the fill, a scalar register)
						const unsigned long long claimed = __builtin_amdgcn_ballot_w64(owner);
						if (owner) {
#pragma unroll
							for (int i = 0; i < NB; ++i) R.w[i][sl] = (i == NB - 1) ? (bw[i] | ((uint64_t)tag << RT_TAG_SHIFT)) : bw[i];
							R.list[rfill + __builtin_amdgcn_mbcnt_hi((uint32_t)(claimed >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)claimed, 0u))] = (uint16_t)sl;
						}
						rfill += (uint32_t)__popcll(claimed);
					}
					wave_sync();
					if (pending) {
						bool same = true;
						if (!owner) {
#pragma unroll
							for (int i = 0; i < NB; ++i) same = same && (R.w[i][sl] == ((i == NB - 1) ? (bw[i] | ((uint64_t)tag << RT_TAG_SHIFT)) : bw[i]));
						}
						if (same) {
							atomicAdd(&R.cnt[sl], 1u);
							atomicMin(reinterpret_cast<unsigned long long *>(&R.stamp[sl]), (unsigned long long)stamp0);
							pending = false;
						} else {
							sl = (sl + 1) & (RT_SLOTS - 1);  // same fingerprint, different record: keep probing
						}
					}
					wave_sync();
				} while (__any(pending));
				if (rfill > RT_FLUSH) overflow = !expand(hsub, hn, phase == 0);
			}
			if (!overflow) overflow = !expand(hsub, hn, phase == 0);
			return overflow;
		};
		partial_flushes = false;
		if (!SLOW) {
			if (ANCH ? !(pass(0, 1, 0, false) || pass(0, 1, 1, true)) : !pass(0, 1, 2, true)) {
				flush();  // once per bucket: the entries are unique in the log
			} else {      // leave the bucket to the second launch; nothing of it was written
				discard();
				if (lane == 0) retry_list[atomicAdd(&counters[PC_AGG_RETRY], 1ull)] = b;
				prefetched = next_ready;
			}
		} else {
			// The bucket's k-mers are split by bits -- of their record's bucket number for the first rbits levels, of their
			// own hash (from bit 12 up) below that: node (nsub, sub) holds those whose log2(nsub) split bits equal sub.  A node whose k-mers do not fit the table has written nothing yet; it is replaced
			// by its two halves (2*nsub, sub) and (2*nsub, sub + nsub), visited depth first without a stack: after a node
			// that is a low half comes its sibling, after a high half the parent is complete.  Only a node that cannot
			// be halved any more is redone with partial flushes, and its entries alone lose the unique mark.
			uint32_t nsub = 2, sub = 0;  // a listed bucket is known not to fit in one pass
			for (;;) {
				if (!pass(sub, nsub, 2, true)) {
					flush();  // once per (bucket, node): the entries are unique in the log
					partial_flushes = false;
					bool done = false;
					for (;;) {
						if (nsub == 1) {
							done = true;
							break;
						}
						const uint32_t top = nsub >> 1;
						if (sub & top) {  // high half done: so is the parent
							sub &= top - 1;
							nsub = top;
						} else {
							sub |= top;
							break;
						}
					}
					if (done) break;
					continue;
				}
				discard();  // nothing of this node was written
				prefetched = false;
				if ((nsub >> min(31u - (uint32_t)__clz((int)nsub), rbits)) < AGG_MAX_SUBPASSES) nsub *= 2;  // low half first: sub keeps its value
				else partial_flushes = true;
			}
		}
		tspilled = false;
		b = b2;
		cb = nb_begin;
		ce = nb_end;
	}
	if (lane == 0) {
		if (keys_written) atomicAdd(&counters[PC_LOG_KEYS], keys_written);
		if (multi_written) atomicAdd(&counters[PC_LOG_MULTI], (unsigned long long)multi_written);
		if (nonunique_flushes) atomicAdd(&counters[PC_NONUNIQUE], (unsigned long long)nonunique_flushes);
		if (cut_anchored) atomicAdd(&counters[PC_CUT_ANCHORED], (unsigned long long)cut_anchored);
		if (cut_expanded) atomicAdd(&counters[PC_CUT_EXPANDED], (unsigned long long)cut_expanded);
	}
	// close the last chunk
	if (chunk_base != NO_CHUNK)
		for (uint32_t i = chunk_used + lane; i < LOG_CHUNK; i += 64)
			if (chunk_base + i < log_cap) {
				if (W == 1 && ai.compact) reinterpret_cast<uint64_t *>(log)[2 * (chunk_base + i)] = EMPTY;
				else log[chunk_base + i].k0 = EMPTY;
			}
	if (ai.items) {
		__syncthreads();
		if (threadIdx.x < AGG_ITEM_RUNS) ai.run_count[threadIdx.x * ai.n_wg + blockIdx.x] = s_icur[threadIdx.x] < ai.run_cap ? s_icur[threadIdx.x] : ai.run_cap;
	}
}

// ------------------------------------------------------------------------------------------------
// One *workgroup* per bucket: the aggregation over a two-level partition (2^16 buckets of ~9 K records on C2 instead
// of 2^22 of ~140), so that the third partition level, its histogram and its read + write of every record go away.
// Same two-table scheme as k_aggregate, the tables shared by all wavefronts of the workgroup:
//  1. the record table merges bit-identical records.  A slot is claimed with a compare-and-swap on a 64-bit
//     fingerprint; whether a record really equals the slot's (fingerprints can collide) is checked after a barrier,
//     when the owner's words are in place.  New slots are ticketed (rfill): a record that finds the table at its
//     limit stays pending, the table is expanded into the k-mer table and emptied, and the record tries again;
//  2. the expansion hands the distinct records out to the wavefronts 64 at a time; inside a wavefront their k-mers
//     are laid out flat, a lane per k-mer, exactly as in k_aggregate, and go to the shared k-mer table with LDS
//     atomics.  A round of 64 k-mers only starts while the table holds at most KLIMIT keys, so with THREADS k-mers
//     in flight it never fills up; above that the attempt is void (nothing was written) and the bucket goes to the
//     retry list (SLOW == false) or is split further (SLOW == true), as in k_aggregate;
//  3. the k-mer table is flushed once per bucket (or per node of the split): one global atomic reserves exactly its
//     number of entries in the log, so the log has no holes.  There are no partial flushes: a share of a bucket that
//     4096 hash sub-passes below the record levels cannot hold is counted in PC_LDS_SPILL and the host falls back.
template <int W, int KS, int RS>
struct WgTables {
	uint64_t k0[KS];
	uint64_t k1[W == 2 ? KS : 1];
	uint64_t stamp[KS];
	uint32_t cnt[KS];
	uint32_t edges[KS / 4];  // one byte per slot
	uint16_t klist[KS];
	unsigned long long rfp[RS];  // 0 = free, else a 63-bit fingerprint of the record | 1
	uint64_t rw[2 * W][RS];
	uint64_t rstamp[RS];
	uint32_t rcnt[RS];
	uint16_t rtag[RS];  // header bits 16..30: length, neighbour flags, minimizer offset
	uint16_t rlist[RS];
	uint32_t kfill, rfill, rnlist, flag;
	uint32_t any[2];  // "a thread still holds a record", alternating between rounds
	unsigned long long log_base;
};

// Workgroup barrier for data that travels through LDS only.  __syncthreads() also waits for every global load of the
// wavefront (vmcnt(0)): the records prefetched for the next round would be waited for at the first barrier after their
// request, with all 16 wavefronts of the CU idle meanwhile.  LDS operations of a wavefront complete in order, so waiting
// for its own (lgkmcnt(0)) before the barrier is what visibility inside the workgroup needs.
__device__ __forceinline__ void lds_barrier() {
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int KS>
__device__ __forceinline__ uint32_t wg_slot(uint32_t hash) {
	return hash >> (32 - __builtin_ctz(KS));
}

// One LDS atomic per wavefront instead of one per lane on a counter that every lane wants to bump: the lanes for which
// pred holds get consecutive values.  Called by the whole wavefront (all 64 lanes active).
__device__ __forceinline__ uint32_t wave_claim(uint32_t *counter, bool pred, uint32_t lane) {
	const unsigned long long bal = __ballot(pred);
	if (bal == 0) return 0;  // uniform
	const uint32_t leader = (uint32_t)__ffsll((long long)bal) - 1u;
	const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
	uint32_t base = 0;
	if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(bal));
	base = (uint32_t)__shfl((int)base, (int)leader);
	return base + rank;
}

// Merge (stamp, weight, edge bits) of the lanes for which `active` holds into the shared k-mer table.  Called by the whole
// wavefront; returns false for a lane whose key found no slot.
template <int W, int KS, int RS>
__device__ __forceinline__ bool wg_upsert(WgTables<W, KS, RS> &t, bool active, uint32_t hash, uint64_t i0, uint64_t i1, uint64_t stamp,
                                          uint32_t emask, uint32_t weight, uint32_t lane) {
	// double hashing: the step is odd, so a probe sequence visits every slot; with linear probing the longest cluster of a
	// table decides how long a wavefront (its slowest lane) probes
	uint32_t s = wg_slot<KS>(hash);
	const uint32_t step = ((hash >> 4) & (KS - 1)) | 1u;
	bool searching = active;
	for (uint32_t probe = 0; probe < (uint32_t)KS && __any(searching); ++probe) {
		uint64_t c0 = searching ? t.k0[s] : 0;
		bool claimed = false;
		if (searching && c0 == EMPTY) {
			const uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long *>(&t.k0[s]), EMPTY, i0);
			claimed = prev == EMPTY;
			c0 = claimed ? i0 : prev;
		}
		const uint32_t idx = wave_claim(&t.kfill, claimed, lane);
		if (claimed) t.klist[idx] = (uint16_t)s;
		if (searching && c0 == i0) {
			bool match = true;
			if (W == 2) {
				uint64_t c1 = t.k1[s];
				if (c1 == EMPTY) {
					const uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long *>(&t.k1[s]), EMPTY, i1);
					c1 = (prev == EMPTY) ? i1 : prev;
				}
				match = (c1 == i1);
			}
			if (match) {
				atomicMin(reinterpret_cast<unsigned long long *>(&t.stamp[s]), stamp);
				atomicAdd(&t.cnt[s], weight);
				if (emask) atomicOr(&t.edges[s >> 2], emask << ((s & 3u) * 8u));
				searching = false;
			}
		}
		if (searching) s = (s + step) & (KS - 1);
	}
	return !searching;
}

template <int W, int THREADS, int KS, int RS, bool SLOW, bool C = false>
__global__ __launch_bounds__(THREADS) void k_aggregate_wg(const RecMem<W, C> *__restrict__ recs, const unsigned long long *__restrict__ bucket_begin,
                                                          uint32_t nbuckets, int k, Slot *__restrict__ log, uint64_t log_cap,
                                                          unsigned long long *__restrict__ counters, uint32_t flag_unique,
                                                          const uint32_t *__restrict__ bucket_list, uint32_t *__restrict__ retry_list, int m,
                                                          uint32_t rbits) {
	constexpr int NB = 2 * W;
	constexpr uint32_t NWAVES = THREADS / 64;
	constexpr uint32_t KLIMIT = KS * 5 / 8;  // a round of the expansion starts only at or below this: at most THREADS more keys arrive
	constexpr uint32_t RLIMIT = RS / 2;      // distinct records the record table takes before it is expanded
	constexpr uint32_t GROUP = 32;           // distinct records a wavefront expands at a time
	static_assert(KLIMIT + THREADS < KS, "the k-mer table must hold a full round above its limit");
	static_assert((KS & (KS - 1)) == 0 && (RS & (RS - 1)) == 0 && KS <= 65536 && RS <= 65536, "table sizes");
	__shared__ WgTables<W, KS, RS> T;
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const uint64_t mask0 = (2 * k >= 64) ? ~0ull : ((1ull << (2 * k)) - 1);
	const uint64_t mask1 = (k > 32) ? ((1ull << (2 * k - 64)) - 1) : 0ull;
	unsigned long long keys_written = 0;  // thread 0 only
	bool log_full = false;                // uniform
	volatile unsigned long long *const vfp = T.rfp;

	auto clear_all = [&]() {
		for (uint32_t i = tid; i < (uint32_t)KS; i += THREADS) {
			T.k0[i] = EMPTY;
			if (W == 2) T.k1[i] = EMPTY;
			T.stamp[i] = INF_STAMP;
			T.cnt[i] = 0;
			if (i < (uint32_t)KS / 4) T.edges[i] = 0;
		}
		for (uint32_t i = tid; i < (uint32_t)RS; i += THREADS) {
			T.rfp[i] = 0;
			T.rcnt[i] = 0;
			T.rstamp[i] = INF_STAMP;
		}
		if (tid == 0) {
			T.kfill = 0;
			T.rfill = 0;
			T.rnlist = 0;
			T.flag = 0;
			T.any[0] = T.any[1] = 0;
		}
		lds_barrier();
	};

	// the k-mer table's entries go to the log; called by all threads between barriers
	auto flush = [&]() {
		const uint32_t n = T.kfill;
		lds_barrier();
		if (n == 0) return;
		const uint64_t mark = flag_unique ? LOG_UNIQUE : 0ull;
		if (tid == 0) {
			const unsigned long long base = atomicAdd(&counters[PC_LOG_CURSOR], (unsigned long long)n);
			if (base + n > log_cap) atomicAdd(&counters[PC_LOG_OVF], 1ull);
			T.log_base = base;
			keys_written += n;
			if (!mark) atomicAdd(&counters[PC_NONUNIQUE], 1ull);
		}
		lds_barrier();
		const unsigned long long base = T.log_base;
		if (base + n > log_cap) log_full = true;  // the launch is void: this workgroup stops after this bucket
		for (uint32_t i = tid; i < n; i += THREADS) {
			const uint32_t sl = T.klist[i];
			if (base + i < log_cap) {
				Slot e;
				e.k0 = T.k0[sl];
				e.k1 = (W == 2) ? T.k1[sl] : EMPTY;
				e.stamp = T.stamp[sl];
				const uint32_t eb = (T.edges[sl >> 2] >> ((sl & 3u) * 8u)) & 0xFFu;
				e.meta = (uint64_t)T.cnt[sl] | ((uint64_t)eb << PEND_SHIFT) | mark;
				log[base + i] = e;
			}
		}
		lds_barrier();  // every edge byte has been read before its word is cleared
		for (uint32_t i = tid; i < n; i += THREADS) {
			const uint32_t sl = T.klist[i];
			T.k0[sl] = EMPTY;
			if (W == 2) T.k1[sl] = EMPTY;
			T.stamp[sl] = INF_STAMP;
			T.cnt[sl] = 0;
			T.edges[sl >> 2] = 0;
		}
		if (tid == 0) T.kfill = 0;
		lds_barrier();
	};

	// every distinct record of the record table goes through the k-mer table once, weighted; of its k-mers only those
	// of sub-pass `sub` (of `nsub`, a power of two) are taken.  Called by all threads behind a barrier; returns false when
	// the k-mer table would overfill (the caller drops the attempt).  Leaves the record table empty.  The wavefronts
	// fetch groups of GROUP records from a counter (T.any[0]) until none is left.
	auto expand = [&](uint32_t sub, uint32_t nsub) -> bool {
		const uint32_t n = T.rnlist;  // settled: the callers are behind a barrier
		for (;;) {
			uint32_t g0 = 0;
			if (lane == 0) g0 = atomicAdd(&T.any[0], GROUP);
			g0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)g0);
			if (g0 >= n) break;
			const uint32_t e = g0 + lane;
			uint32_t rlen = 0, rsl = 0, rweight = 0;
			uint64_t rstamp = 0;
			if (lane < GROUP && e < n) {
				rsl = T.rlist[e];
				rlen = ((uint32_t)T.rtag[rsl] & 63u) + 1;
				rweight = T.rcnt[rsl];
				rstamp = T.rstamp[rsl];
				T.rfp[rsl] = 0;  // the slot is free again (its words and tag stay readable: nothing is claimed during an expansion)
				T.rcnt[rsl] = 0;
				T.rstamp[rsl] = INF_STAMP;
			}
			uint32_t incl = rlen;  // inclusive prefix sums of the lengths
#pragma unroll
			for (int d = 1; d < (int)GROUP; d <<= 1) {
				const uint32_t o = __shfl_up(incl, d);
				if (lane >= (uint32_t)d) incl += o;
			}
			const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, GROUP - 1);
			const uint32_t rpre = incl - rlen;
			for (uint32_t t0 = 0; t0 < total; t0 += 64) {
				// the fill as this wavefront sees it now; whatever it misses is in flight and covered by the margin above the limit
				if ((uint32_t)__builtin_amdgcn_readfirstlane((int)*(volatile uint32_t *)&T.kfill) > KLIMIT) {
					if (lane == 0) T.flag = 2;  // the attempt is void
					break;
				}
				const uint32_t t = t0 + lane;
				uint32_t own = 0;  // number of records whose prefix sum is <= t (all lanes take part in the shuffles)
#pragma unroll
				for (uint32_t st = GROUP / 2; st; st >>= 1) {
					const uint32_t v = (uint32_t)__shfl((int)incl, (int)(own + st - 1));
					if (v <= t) own += st;
				}
				own = own < GROUP - 1 ? own : GROUP - 1;
				const uint32_t sl = (uint32_t)__shfl((int)rsl, (int)own);
				const uint32_t i = t - (uint32_t)__shfl((int)rpre, (int)own);
				const uint32_t weight = (uint32_t)__shfl((int)rweight, (int)own);
				const uint64_t stamp0 = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(rstamp >> 32), (int)own) << 32) | (uint32_t)__shfl((int)(uint32_t)rstamp, (int)own);
				bool take = t < total;
				uint64_t i0 = 0, i1 = 0, kstamp = 0;
				uint32_t hx = 0, em = 0;
				if (take) {
					uint64_t bw[NB];
#pragma unroll
					for (int q = 0; q < NB; ++q) bw[q] = T.rw[q][sl];
					const uint32_t tag = T.rtag[sl];
					const uint32_t len = (tag & 63u) + 1, hasL = (tag >> 6) & 1u, hasR = (tag >> 7) & 1u;
					const uint32_t pos = hasL + i;  // the k-mer's first base in the record
					uint64_t f0, f1 = 0, r0, r1;
					f0 = rec_ext<W>(bw, pos) & mask0;
					if (W == 2 && k > 32) f1 = rec_ext<W>(bw, pos + 32) & mask1;
					revcomp<W>(f0, f1, k, r0, r1);
					const bool use_rc = kmer_less<W>(r0, r1, f0, f1);
					if (pos > 0) {  // a k-mer precedes this one in its read
						const uint32_t prev = rec_base<W>(bw, pos - 1);
						em = 1u << (use_rc ? 7u - prev : prev);
					}
					if (i + 1 < len || hasR) {
						const uint32_t nb = rec_base<W>(bw, pos + (uint32_t)k);
						em |= 1u << (use_rc ? 3u - nb : 4u + nb);
					}
					to_internal<W>(use_rc ? r0 : f0, use_rc ? r1 : f1, i0, i1);
					hx = agg_hash<W>(i0, i1);
					kstamp = stamp0 + i;
					if (SLOW && (((hx * 0x85ebca6bu) >> 16) & (nsub - 1)) != sub) take = false;  // another sub-pass owns this k-mer
				}
				if (!wg_upsert<W, KS, RS>(T, take, hx, i0, i1, kstamp, em, weight, lane)) T.flag = 2;  // no slot left (cannot happen below the limit): the attempt is void
			}
		}
		lds_barrier();
		const bool ok = T.flag != 2;
		lds_barrier();
		if (tid == 0) {
			T.rfill = 0;
			T.rnlist = 0;
			T.any[0] = 0;
		}
		lds_barrier();
		return ok;
	};

	clear_all();
	for (uint32_t b = blockIdx.x; b < nbuckets && !log_full; b += gridDim.x) {
		const uint32_t bb = (SLOW && bucket_list) ? bucket_list[b] : b;  // no list: every bucket
		const uint64_t cb = bucket_begin[bb], ce = bucket_begin[bb + 1];
		const uint32_t nblk = (uint32_t)((ce - cb + 63) / 64);  // blocks of 64 records, dealt to the wavefronts from a counter (T.any[1])
		// One pass over the bucket's records for sub-pass `sub` of `nsub`; true = the k-mer table overflowed.  The wavefronts
		// work through blocks on their own (no barrier per block: a slot of the record table is claimed busy, filled and then
		// published ready by its owner, everybody else waits for ready before comparing) and meet only when the record
		// table has reached its limit -- the wavefront that needs a new slot waits at the barrier, the others arrive as
		// they run into the limit too or run out of blocks -- for the expansion, and at the end.
		auto pass = [&](uint32_t sub, uint32_t nsub) -> bool {
			const uint32_t rl = SLOW ? min(31u - (uint32_t)__clz((int)nsub), rbits) : 0u;  // record levels of the split, then hash levels
			const uint32_t rmask = (1u << rl) - 1u, rsub = sub & rmask;
			const uint32_t hsub = sub >> rl, hn = nsub >> rl;
			if (tid == 0) {
				T.any[1] = NWAVES;  // the first NWAVES blocks are taken
				T.flag = (T.flag == 2) ? 2u : 0u;
			}
			if (tid == 0) T.log_base = 0;  // counts the wavefronts that have run out of blocks (the word is free until the flush)
			lds_barrier();
			uint32_t blk = wave, next_blk = 0;
			bool finished = false, counted = false, carry = false;
			RecMem<W, C> nxt = {};
			if (blk < nblk && cb + (uint64_t)blk * 64 + lane < ce) nxt = recs[cb + (uint64_t)blk * 64 + lane];
			uint64_t bw[NB], stamp0 = 0;
			unsigned long long fp = 0;
			uint32_t tag = 0, sl = 0, home = 0, rstep = 1;
			bool pending = false;
			for (;;) {
				while (!finished) {
					if (!carry) {
						if (blk >= nblk) {
							finished = true;
							break;
						}
						const Rec<W> rec = rec_unpack(nxt);
						pending = cb + (uint64_t)blk * 64 + lane < ce;
						if (SLOW && rl && pending && (rec_bucket<W>(rec, m) & rmask) != rsub) pending = false;  // a record of another pass
						next_blk = 0;
						if (lane == 0) next_blk = atomicAdd(&T.any[1], 1u);
						next_blk = (uint32_t)__builtin_amdgcn_readfirstlane((int)next_blk);
						if (next_blk < nblk && cb + (uint64_t)next_blk * 64 + lane < ce) nxt = recs[cb + (uint64_t)next_blk * 64 + lane];
						const uint64_t hdr = rec.w[NB];
						tag = (uint32_t)(hdr >> 16) & 0x7FFFu;
						stamp0 = ((hdr >> 31) << 16) | ((uint32_t)hdr & 0xFFFFu);
						uint64_t h = (uint64_t)tag * 0x9E3779B97F4A7C15ull;
#pragma unroll
						for (int i = 0; i < NB; ++i) {
							bw[i] = rec.w[i];
							h = (h ^ bw[i]) * 0xC2B2AE3D27D4EB4Full;
							h ^= h >> 29;
						}
						h *= 0xBF58476D1CE4E5B9ull;
						h ^= h >> 32;
						fp = h | (1ull << 63) | 1ull;  // ready form; the busy form has bit 0 clear and is never 0
						home = (uint32_t)(h >> 40) & (RS - 1);
						rstep = ((uint32_t)(h >> 20) & (RS - 1)) | 1u;
					}
					sl = home;  // (again after an expansion: the table is empty then)
					// claim or find this record's slot, count it
					bool searching = pending, blocked = false;
					for (uint32_t probe = 0; probe < 4u * RS && __any(searching); ++probe) {
						unsigned long long cur = searching ? vfp[sl] : 1ull;
						const bool empty = searching && cur == 0;
						const uint32_t ticket = wave_claim(&T.rfill, empty, lane);  // a new slot needs a ticket below the limit
						const bool full = empty && ticket >= RLIMIT;
						bool won = false;
						if (empty && !full) {
							cur = atomicCAS(&T.rfp[sl], 0ull, fp & ~1ull);
							won = cur == 0;
						}
						const uint32_t back = (uint32_t)__popcll(__ballot(empty && !won));  // tickets that took no slot
						if (back && lane == 0) atomicSub(&T.rfill, back);
						const uint32_t li = wave_claim(&T.rnlist, won, lane);
						if (won) {
#pragma unroll
							for (int i = 0; i < NB; ++i) T.rw[i][sl] = bw[i];
							T.rtag[sl] = (uint16_t)tag;
							T.rlist[li] = (uint16_t)sl;
							asm volatile("" ::: "memory");  // LDS operations of a wavefront are carried out in order: the slot is complete before it reads as ready
							vfp[sl] = fp;
							atomicAdd(&T.rcnt[sl], 1u);
							atomicMin(reinterpret_cast<unsigned long long *>(&T.rstamp[sl]), (unsigned long long)stamp0);
							searching = false;
							pending = false;
						} else if (full) {
							searching = false;  // stays pending: the table is at its limit, wait for the expansion
							blocked = true;
						} else if (searching && (cur | 1ull) == fp) {
							if (cur & 1ull) {  // published: is it this record?
								asm volatile("" ::: "memory");
								bool same = ((uint32_t)T.rtag[sl] == tag);
#pragma unroll
								for (int i = 0; i < NB; ++i) same = same && (T.rw[i][sl] == bw[i]);
								if (same) {
									atomicAdd(&T.rcnt[sl], 1u);
									atomicMin(reinterpret_cast<unsigned long long *>(&T.rstamp[sl]), (unsigned long long)stamp0);
									searching = false;
									pending = false;
								} else {
									sl = (sl + rstep) & (RS - 1);  // same fingerprint, different record
								}
							}
							// else: its owner is still filling it in -- look again
						} else if (searching) {
							sl = (sl + rstep) & (RS - 1);
						}
					}
					if (__any(blocked || searching)) {  // (searching: the probe bound ran out -- never seen; the expansion empties the table)
						carry = true;                    // the lanes that are still pending try again after the expansion
						break;
					}
					carry = false;
					blk = next_blk;
				}
				if (finished && !counted) {
					if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long *>(&T.log_base), 1ull);
					counted = true;
				}
				lds_barrier();
				const bool all_done = T.log_base == (unsigned long long)NWAVES;
				if (!expand(hsub, hn)) return true;
				if (all_done) break;
			}
			return false;
		};
		if (!SLOW) {
			if (!pass(0, 1)) {
				flush();  // once per bucket: the entries are unique in the log
			} else {      // leave the bucket to the second launch; nothing of it was written
				clear_all();
				if (tid == 0) retry_list[atomicAdd(&counters[PC_AGG_RETRY], 1ull)] = b;
			}
		} else {
			// node (nsub, sub) = the k-mers whose log2(nsub) split bits equal sub; see k_aggregate
			uint32_t nsub = 2, sub = 0;  // a listed bucket is known not to fit in one pass
			for (;;) {
				if (!pass(sub, nsub)) {
					flush();
					bool done = false;
					for (;;) {
						if (nsub == 1) {
							done = true;
							break;
						}
						const uint32_t top = nsub >> 1;
						if (sub & top) {  // high half done: so is the parent
							sub &= top - 1;
							nsub = top;
						} else {
							sub |= top;
							break;
						}
					}
					if (done) break;
					continue;
				}
				clear_all();  // nothing of this node was written
				if ((nsub >> min(31u - (uint32_t)__clz((int)nsub), rbits)) < AGG_MAX_SUBPASSES) {
					nsub *= 2;  // low half first: sub keeps its value
				} else {
					// 2^rbits x 4096 shares and one of them still holds more k-mers than the table: no input that fits the
					// device gets here; the host then counts with the direct sweep instead (nothing wrong is ever written)
					if (tid == 0) atomicAdd(&counters[PC_LDS_SPILL], 1ull);
					break;
				}
			}
		}
		if (tid == 0) T.flag = 0;
		lds_barrier();
	}
	if (tid == 0 && keys_written) atomicAdd(&counters[PC_LOG_KEYS], keys_written);
}

// Fallback of the record path: every k-mer of every record goes straight to the vertex table (one upsert per
// occurrence), used when an estimate of the partitioned path was exceeded.
// The distinct k-mers of the record sample, counted in ONE launch right behind the kernel that took the sample (its size is
// read from the device: no host round trip in between): every k-mer of every sampled record claims a slot of a small
// open-addressing set of 64-bit keys (two-word keys: a 64-bit mix of both words; a collision there would cost the estimate one
// k-mer).  counters[PC_SAMPLE_DISTINCT] += new keys; counters[PC_SAMPLE_FAIL] when the set is too full to be trusted.
// (A nested partitioned sweep over the sample gave the same number through six host round trips and a dozen launches.)
template <int W, bool C>
__global__ __launch_bounds__(256) void k_sample_distinct(const RecMem<W, C> *__restrict__ sample, uint64_t sample_cap, int k,
                                                         unsigned long long *__restrict__ set, uint64_t set_cap /* power of two */,
                                                         unsigned long long *__restrict__ counters) {
	constexpr int NB = 2 * W;
	const uint64_t n = counters[PC_SAMPLE] < sample_cap ? counters[PC_SAMPLE] : sample_cap;
	const uint64_t mask0 = (2 * k >= 64) ? ~0ull : ((1ull << (2 * k)) - 1);
	const uint64_t mask1 = (k > 32) ? ((1ull << (2 * k - 64)) - 1) : 0ull;
	uint32_t fresh = 0;
	bool failed = false;
	for (uint64_t ri = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; ri < n; ri += (uint64_t)gridDim.x * blockDim.x) {
		const Rec<W> rec = rec_unpack(sample[ri]);
		if (rec.w[NB] == REC_INVALID) continue;
		uint64_t bw[NB];
#pragma unroll
		for (int i = 0; i < NB; ++i) bw[i] = rec.w[i];
		const uint64_t hdr = rec.w[NB];
		const uint32_t len = ((uint32_t)(hdr >> 16) & 63u) + 1;
		const uint32_t hasL = (uint32_t)(hdr >> 22) & 1u;
		for (uint32_t i = 0; i < len; ++i) {
			uint64_t f0 = rec_ext<W>(bw, hasL + i) & mask0, f1 = 0, r0, r1;
			if (W == 2 && k > 32) f1 = rec_ext<W>(bw, hasL + i + 32) & mask1;
			revcomp<W>(f0, f1, k, r0, r1);
			const bool use_rc = kmer_less<W>(r0, r1, f0, f1);
			uint64_t key = use_rc ? r0 : f0;
			if (W == 2) key = (key ^ ((use_rc ? r1 : f1) * 0x9E3779B97F4A7C15ull)) * 0xC2B2AE3D27D4EB4Full;
			if (key == ~0ull) key = 0;  // ~0 marks a free slot
			uint64_t h = key * 0xBF58476D1CE4E5B9ull;
			h ^= h >> 31;
			uint64_t s = h & (set_cap - 1);
			uint32_t probe = 0;
			for (; probe < 256; ++probe) {
				unsigned long long cur = set[s];
				if (cur == ~0ull) {
					cur = atomicCAS(&set[s], ~0ull, (unsigned long long)key);
					if (cur == ~0ull) {
						fresh++;
						break;
					}
				}
				if (cur == key) break;
				s = (s + 1) & (set_cap - 1);
			}
			if (probe == 256) failed = true;
		}
	}
#pragma unroll
	for (int d = 32; d; d >>= 1) fresh += __shfl_xor(fresh, d);
	if ((threadIdx.x & 63u) == 0 && fresh) atomicAdd(&counters[PC_SAMPLE_DISTINCT], (unsigned long long)fresh);
	if (failed) atomicAdd(&counters[PC_SAMPLE_FAIL], 1ull);
}

template <int W, bool C = false>
__global__ __launch_bounds__(256) void k_count_records(const RecMem<W, C> *__restrict__ recs, uint64_t n, int k, Slot *__restrict__ tab,
                                                       uint64_t cap, unsigned long long *__restrict__ counters, int c_fill) {
	constexpr int NB = 2 * W;
	__shared__ uint32_t s_claimed;
	if (threadIdx.x == 0) s_claimed = 0;
	__syncthreads();
	const uint64_t mask0 = (2 * k >= 64) ? ~0ull : ((1ull << (2 * k)) - 1);
	const uint64_t mask1 = (k > 32) ? ((1ull << (2 * k - 64)) - 1) : 0ull;
	uint32_t claimed = 0;
	bool failed = false;
	for (uint64_t ri = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; ri < n; ri += (uint64_t)gridDim.x * blockDim.x) {
		const Rec<W> rec = rec_unpack(recs[ri]);
		if (rec.w[NB] == REC_INVALID) continue;
		uint64_t bw[NB];
#pragma unroll
		for (int i = 0; i < NB; ++i) bw[i] = rec.w[i];
		const uint64_t hdr = rec.w[NB];
		const uint32_t len = ((uint32_t)(hdr >> 16) & 63u) + 1;
		const uint32_t hasL = (uint32_t)(hdr >> 22) & 1u, hasR = (uint32_t)(hdr >> 23) & 1u;
		const uint64_t stamp0 = ((hdr >> 31) << 16) | ((uint32_t)hdr & 0xFFFFu);
		uint64_t f0 = rec_ext<W>(bw, hasL) & mask0, f1 = 0;
		if (W == 2 && k > 32) f1 = rec_ext<W>(bw, hasL + 32) & mask1;
		uint64_t r0, r1;
		revcomp<W>(f0, f1, k, r0, r1);
		uint32_t prev = hasL ? (uint32_t)bw[0] & 3u : 0;
		bool prev_valid = hasL;
		for (uint32_t i = 0; i < len; ++i) {
			const bool has_next = (i + 1 < len) || hasR;
			const uint32_t nb = has_next ? rec_base<W>(bw, hasL + i + (uint32_t)k) : 0;
			const bool use_rc = kmer_less<W>(r0, r1, f0, f1);
			uint32_t em = 0;
			if (prev_valid) em = 1u << (use_rc ? 7u - prev : prev);
			if (has_next) em |= 1u << (use_rc ? 3u - nb : 4u + nb);
			uint64_t i0, i1;
			to_internal<W>(use_rc ? r0 : f0, use_rc ? r1 : f1, i0, i1);
			if (!upsert<W>(tab, cap, i0, i1, stamp0 + i, 1ull, em, claimed)) failed = true;
			prev = (uint32_t)f0 & 3u;
			prev_valid = true;
			if (i + 1 < len) {
				if (W == 1) {
					f0 = (f0 >> 2) | ((uint64_t)nb << (2 * (k - 1)));
					r0 = ((r0 << 2) | (uint64_t)(3u - nb)) & mask0;
				} else {
					f0 = (f0 >> 2) | (f1 << 62);
					if (k > 32) {
						f1 = (f1 >> 2) | ((uint64_t)nb << (2 * (k - 1) - 64));
						r1 = ((r1 << 2) | (r0 >> 62)) & mask1;
						r0 = (r0 << 2) | (uint64_t)(3u - nb);
					} else {
						f0 |= (uint64_t)nb << (2 * (k - 1));
						r0 = ((r0 << 2) | (uint64_t)(3u - nb)) & mask0;
					}
				}
			}
		}
	}
	if (claimed) atomicAdd(&s_claimed, claimed);
	__syncthreads();
	if (threadIdx.x == 0 && s_claimed) atomicAdd(&counters[c_fill], (unsigned long long)s_claimed);
	if (failed) atomicAdd(&counters[1 /*C_ERR*/], 1ull);
}

}  // namespace rg
