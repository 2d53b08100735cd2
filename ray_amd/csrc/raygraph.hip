// raygraph.hip -- kernels and C ABI of the gfx950 graph-construction path (include/raygraph.h).
//
// Path: A0 packed reads -> A1-A4 encode / reverse complement / canonical /
// owner hash -> A5+A7+A8 count into the vertex table (+A9-A11 edge bits in the same sweep) -> A6+A7
// Bloom admission resolved from arrival stamps -> A12 histogram -> export.  All arithmetic is
// 64-bit integer; the kernels are bound by random 32-byte sector traffic and atomics, not by
// ALU, and there is no dense contraction anywhere (no MFMA).
//
// No CPU fallback exists: every entry point needs a HIP device and fails with RG_ERR_HIP otherwise.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/raygraph.h"
#include "rg_device.h"
#include "rg_partition.h"
#include "rg_bloom.h"
#include "rg_table.h"
#include "rg_bloomx.h"

using namespace rg;

// ================================================================================ kernels

static constexpr int BLOCK = 256;
static constexpr int TILE_READS = 32;  // reads per workgroup tile in the k-mer sweep

enum Counter {
	C_FILL = 0,     // claimed slots (main table)
	C_ERR,          // probe limit hit
	C_OCC,          // k-mer occurrences streamed
	C_RETAINED,     // stored pairs after admission
	C_INSERTIONS,   // Bloom insertions (distinct keys whose first sight was not a false positive)
	C_SETBITS,      // Bloom bits set
	C_OVF,          // histogram overflow entries
	C_FILL_LOCAL,   // claimed slots (local aggregation table)
	C_EXPORT,       // export cursor
	C_ADOPT_BAD,    // received entries that are not marked unique (rg_merge_entries)
	C_LOG_MULTI,    // = 10: entries of the pending log seen twice or more (rg_partition.h PC_LOG_MULTI; k_adopt_entries counts them too)
	C_FP_COUNT,     // = 11: keys the Bloom key partition marked as false positives (k_xor_fp)
	C_ENT_PRESIZE,  // = 12: admitted entries that found no room in a table range sized in advance (k_scatter_entries)
	C_NUM = 32
};

__global__ void k_init_table(Slot *tab, uint64_t cap) {
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (; i < cap; i += stride) {
		ulonglong2 a = make_ulonglong2(EMPTY, EMPTY), b = make_ulonglong2(INF_STAMP, 0);
		*reinterpret_cast<ulonglong2 *>(&tab[i].k0) = a;
		*reinterpret_cast<ulonglong2 *>(&tab[i].stamp) = b;
	}
}

// A1-A5, A7 (count part), A8, A9-A11: one sweep over a range of reads.
// A workgroup takes TILE_READS reads at a time, prefix-sums their k-mer counts in LDS and deals the
// tile's k-mer positions round-robin to its 256 lanes, so every wavefront stays full whatever the
// read lengths are (ragged and short reads included).  Each lane funnel-shifts its forward k-mer
// out of the packed read, derives the reverse complement with bit reversal, keeps the lower of
// the two (Kmer.cpp:38-47), folds the neighbouring bases into an edge mask in the lower k-mer's
// orientation (Vertex.cpp:110-173) and merges (stamp, +1, mask) into the table.
template <int W>
__global__ __launch_bounds__(BLOCK) void k_count(const uint64_t *__restrict__ P, const uint64_t *__restrict__ offsets,
                                                 const uint32_t *__restrict__ lengths, uint64_t read_begin, uint64_t read_end,
                                                 uint64_t stamp_base, int k, Slot *__restrict__ tab, uint64_t cap,
                                                 unsigned long long *__restrict__ counters, int fill_counter) {
	__shared__ uint32_t s_pre[TILE_READS + 1];
	__shared__ uint64_t s_off[TILE_READS];
	__shared__ uint32_t s_claimed;
	const uint8_t *P8 = reinterpret_cast<const uint8_t *>(P);
	const uint64_t mask0 = (2 * k >= 64) ? ~0ull : ((1ull << (2 * k)) - 1);
	const uint64_t mask1 = (k > 32) ? ((1ull << (2 * k - 64)) - 1) : 0ull;
	const uint64_t nreads = read_end - read_begin;
	const uint64_t ntiles = (nreads + TILE_READS - 1) / TILE_READS;
	uint32_t claimed = 0;
	bool failed = false;
	unsigned long long occ = 0;
	if (threadIdx.x == 0) s_claimed = 0;

	for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
		__syncthreads();
		if (threadIdx.x < TILE_READS) {
			uint64_t r = read_begin + tile * TILE_READS + threadIdx.x;
			uint32_t n = 0;
			uint64_t off = 0;
			if (r < read_end) {
				uint32_t L = lengths[r];
				n = (L >= (uint32_t)k) ? (L - (uint32_t)k + 1) : 0;  // KmerAcademyBuilder.cpp:130-137
				off = offsets[r];
			}
			s_off[threadIdx.x] = off;
			// inclusive scan over the 32 lanes of the tile header
			uint32_t v = n;
			for (int d = 1; d < TILE_READS; d <<= 1) {
				uint32_t t = __shfl_up(v, d, TILE_READS);
				if ((int)threadIdx.x >= d) v += t;
			}
			s_pre[threadIdx.x + 1] = v;
			if (threadIdx.x == 0) s_pre[0] = 0;
		}
		__syncthreads();
		const uint32_t total = s_pre[TILE_READS];
		if (threadIdx.x == 0) occ += total;
		for (uint32_t t = threadIdx.x; t < total; t += BLOCK) {
			// read index: largest i with s_pre[i] <= t
			int lo = 0, hi = TILE_READS;
#pragma unroll
			for (int step = 0; step < 5; ++step) {
				int mid = (lo + hi) >> 1;
				if (s_pre[mid] <= t) lo = mid;
				else hi = mid;
			}
			const uint32_t p = t - s_pre[lo];
			const uint32_t n = s_pre[lo + 1] - s_pre[lo];
			const uint64_t off = s_off[lo];
			const uint64_t bit = off * 8 + 2ull * p;
			uint64_t f0 = ext64(P, bit) & mask0;
			uint64_t f1 = 0;
			if (W == 2 && k > 32) f1 = ext64(P, bit + 64) & mask1;
			uint64_t r0, r1;
			revcomp<W>(f0, f1, k, r0, r1);
			const bool use_rc = kmer_less<W>(r0, r1, f0, f1);  // KmerAcademyBuilder.cpp:162-165
			const uint64_t K0 = use_rc ? r0 : f0, K1 = use_rc ? r1 : f1;
			// edge bits this occurrence contributes (VerticesExtractor.cpp:169-253): the previous k-mer's
			// first base is an ingoing edge, the next k-mer's last base an outgoing edge
			uint32_t em = 0;
			if (p > 0) {
				uint32_t f = base_at(P8, off, p - 1);
				em |= use_rc ? (1u << (7 - f)) : (1u << f);
			}
			if (p + 1 < n) {
				uint32_t c = base_at(P8, off, p + (uint32_t)k);
				em |= use_rc ? (1u << (3 - c)) : (1u << (4 + c));
			}
			uint64_t i0, i1;
			to_internal<W>(K0, K1, i0, i1);
			const uint64_t stamp = ((stamp_base + tile * TILE_READS + (uint64_t)lo) << 16) | p;
			if (!upsert<W>(tab, cap, i0, i1, stamp, 1ull, em, claimed)) failed = true;
		}
	}
	if (claimed) atomicAdd(&s_claimed, claimed);
	__syncthreads();
	if (threadIdx.x == 0) {
		if (s_claimed) atomicAdd(&counters[fill_counter], (unsigned long long)s_claimed);
		if (occ) atomicAdd(&counters[C_OCC], occ);
	}
	if (failed) atomicAdd(&counters[C_ERR], 1ull);
}

// re-insert every entry of an old table into a bigger one (keys are unique: plain payload stores)
template <int W>
__global__ __launch_bounds__(BLOCK) void k_rehash(const Slot *__restrict__ old, uint64_t oldcap, Slot *__restrict__ tab, uint64_t cap,
                                                  unsigned long long *__restrict__ counters) {
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	bool failed = false;
	for (; i < oldcap; i += stride) {
		Slot v = old[i];
		if (v.k0 == EMPTY) continue;
		uint32_t dummy = 0;
		uint64_t s = slot_home(slot_hash<W>(v.k0, v.k1), cap);
		bool done = false;
		for (uint32_t probe = 0; probe < MAX_PROBE && !done; ++probe) {
			Slot *p = tab + s;
			uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long *>(&p->k0), EMPTY, v.k0);
			if (prev == EMPTY) {
				p->k1 = v.k1;
				p->stamp = v.stamp;
				p->meta = v.meta;
				done = true;
			}
			s = slot_next(s, cap);
		}
		(void)dummy;
		if (!done) failed = true;
	}
	if (failed) atomicAdd(&counters[C_ERR], 1ull);
}

// receiver half of the owner exchange: merge 32-byte entries (same layout as Slot) into the table
template <int W>
__global__ __launch_bounds__(BLOCK) void k_merge(const Slot *__restrict__ in, uint64_t n, Slot *__restrict__ tab, uint64_t cap,
                                                 unsigned long long *__restrict__ counters) {
	__shared__ uint32_t s_claimed;
	if (threadIdx.x == 0) s_claimed = 0;
	__syncthreads();
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	uint32_t claimed = 0;
	bool failed = false;
	for (; i < n; i += stride) {
		Slot v = in[i];
		if (v.k0 == EMPTY) continue;  // unused entry of a log chunk
		uint64_t add = v.meta & COUNT_MASK & ~LOG_UNIQUE;
		uint32_t em = (uint32_t)(v.meta >> PEND_SHIFT) & 0xFFu;
		if (!upsert<W>(tab, cap, v.k0, W == 2 ? v.k1 : 0, v.stamp, add, em, claimed, (v.meta & LOG_UNIQUE) != 0)) failed = true;
	}
	if (claimed) atomicAdd(&s_claimed, claimed);
	__syncthreads();
	if (threadIdx.x == 0 && s_claimed) atomicAdd(&counters[C_FILL], (unsigned long long)s_claimed);
	if (failed) atomicAdd(&counters[C_ERR], 1ull);
}

// owner = hash1(lower) % size (Kmer.cpp:165-170, Mock/Parameters.cpp:2116-2118)
template <int W>
__device__ __forceinline__ uint32_t owner_of(const Slot &v, uint32_t size) {
	uint64_t K0, K1;
	from_internal<W>(v.k0, v.k1, K0, K1);
	return (uint32_t)(hash1<W>(K0, K1) % (uint64_t)size);
}

template <int W>
__global__ __launch_bounds__(BLOCK) void k_owner_counts(const Slot *__restrict__ tab, uint64_t cap, uint32_t size,
                                                        unsigned long long *__restrict__ counts) {
	extern __shared__ uint32_t s_cnt[];
	for (uint32_t j = threadIdx.x; j < size; j += blockDim.x) s_cnt[j] = 0;
	__syncthreads();
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (; i < cap; i += stride) {
		Slot v = tab[i];
		if (v.k0 == EMPTY) continue;
		atomicAdd(&s_cnt[owner_of<W>(v, size)], 1u);
	}
	__syncthreads();
	for (uint32_t j = threadIdx.x; j < size; j += blockDim.x)
		if (s_cnt[j]) atomicAdd(&counts[j], (unsigned long long)s_cnt[j]);
}

// scatter entries into per-owner buckets; cursors[r] starts at the bucket's first entry
// Entries grouped by owner.  A workgroup ranks the entries of a tile per owner in LDS and claims its
// share of every owner's range with one global atomic per (tile, owner): the cursors are a handful of
// addresses, one atomic per entry on them would serialise the whole kernel.
static constexpr int OWNER_TILE = 8;  // entries per thread and tile
template <int W>
__global__ __launch_bounds__(BLOCK) void k_owner_scatter(const Slot *__restrict__ tab, uint64_t cap, uint32_t size,
                                                         unsigned long long *__restrict__ cursors, Slot *__restrict__ out) {
	extern __shared__ unsigned long long s_dyn[];
	unsigned long long *s_base = s_dyn;                                   // [size]
	uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_dyn + size);          // [size]
	const uint64_t tile = (uint64_t)blockDim.x * OWNER_TILE;
	for (uint64_t t0 = (uint64_t)blockIdx.x * tile; t0 < cap; t0 += (uint64_t)gridDim.x * tile) {
		for (uint32_t j = threadIdx.x; j < size; j += blockDim.x) s_cnt[j] = 0;
		__syncthreads();
		uint32_t own[OWNER_TILE], rank[OWNER_TILE];
#pragma unroll
		for (int q = 0; q < OWNER_TILE; ++q) {
			const uint64_t i = t0 + threadIdx.x + (uint64_t)q * blockDim.x;
			own[q] = 0xFFFFFFFFu;
			if (i < cap) {
				const Slot v = tab[i];
				if (v.k0 != EMPTY) {
					own[q] = owner_of<W>(v, size);
					rank[q] = atomicAdd(&s_cnt[own[q]], 1u);
				}
			}
		}
		__syncthreads();
		for (uint32_t j = threadIdx.x; j < size; j += blockDim.x)
			s_base[j] = s_cnt[j] ? atomicAdd(&cursors[j], (unsigned long long)s_cnt[j]) : 0ull;
		__syncthreads();
#pragma unroll
		for (int q = 0; q < OWNER_TILE; ++q)
			if (own[q] != 0xFFFFFFFFu) {
				const uint64_t i = t0 + threadIdx.x + (uint64_t)q * blockDim.x;
				out[s_base[own[q]] + rank[q]] = tab[i];
			}
		__syncthreads();
	}
}

// A6: first-touch stamps of the Bloom bits.  bit_i = (hash2(lower) ^ M_i) % m_bits
// (BloomFilter.cpp:124-200).  A bit is set at the first arrival of any key that touches it, so
// stamp[bit] = min over touching keys of their first-arrival stamp.
template <int W>
__global__ __launch_bounds__(BLOCK) void k_bloom_touch(const Slot *__restrict__ tab, uint64_t cap, uint64_t m_bits,
                                                       unsigned long long *__restrict__ bitstamp,
                                                       unsigned long long *__restrict__ counters) {
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	uint32_t newly = 0;
	for (; i < cap; i += stride) {
		Slot v = tab[i];
		if (v.k0 == EMPTY) continue;
		uint64_t K0, K1;
		from_internal<W>(v.k0, v.k1, K0, K1);
		const uint64_t origin = hash2<W>(K0, K1);
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			uint64_t b = (origin ^ BLOOM_MAGIC[j]) % m_bits;
			if (bitstamp[b] > v.stamp) {
				unsigned long long old = atomicMin(&bitstamp[b], (unsigned long long)v.stamp);
				if (old == INF_STAMP) newly++;
			}
		}
	}
	// wave-level reduction of the set-bit count
	for (int d = 32; d > 0; d >>= 1) newly += __shfl_down(newly, d);
	if ((threadIdx.x & 63) == 0 && newly) atomicAdd(&counters[C_SETBITS], (unsigned long long)newly);
}

// A7: admission + coverage (MessageProcessor.cpp:741-804).  With the filter on, a key whose first
// sight found all 8 bits already set (a false positive) is stored from its first sight: coverage
// 1 + N.  Otherwise the first sight only feeds the filter: stored iff N >= 2, coverage N.  With
// the filter off every sight counts: coverage N.  Coverage saturates at 2^32-1 (Vertex.cpp:83-88).
static constexpr uint32_t HIST_LDS = 4096;
static constexpr uint32_t HIST_DENSE = 65536;
static constexpr uint32_t HIST_OVF_CAP = 1u << 20;

// coverage histogram of the stored vertices (CoverageGatherer.cpp:124-158), gathered by the kernel that writes the
// coverages: bins below HIST_LDS in LDS, below HIST_DENSE in a dense global array, the rest as a list
__device__ __forceinline__ void hist_add(uint32_t *s_h, uint64_t cov, unsigned long long *__restrict__ dense, uint32_t *__restrict__ ovf,
                                         unsigned long long *__restrict__ counters) {
	if (cov == 0) return;
	if (cov < HIST_LDS) atomicAdd(&s_h[cov], 1u);
	else if (cov < HIST_DENSE) atomicAdd(&dense[cov], 1ull);
	else {
		unsigned long long pos = atomicAdd(&counters[C_OVF], 1ull);
		if (pos < HIST_OVF_CAP) ovf[pos] = (uint32_t)cov;
	}
}

template <int W, bool BLOOM>
__global__ __launch_bounds__(BLOCK) void k_admit(Slot *__restrict__ tab, uint64_t cap, uint64_t m_bits,
                                                 const unsigned long long *__restrict__ bitstamp,
                                                 unsigned long long *__restrict__ counters, unsigned long long *__restrict__ hist_dense,
                                                 uint32_t *__restrict__ hist_ovf) {
	__shared__ uint32_t s_h[HIST_LDS];
	for (uint32_t j = threadIdx.x; j < HIST_LDS; j += blockDim.x) s_h[j] = 0;
	__syncthreads();
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	uint32_t retained = 0, inserted = 0;
	for (; i < cap; i += stride) {
		Slot v = tab[i];
		if (v.k0 == EMPTY) continue;
		const uint64_t N = v.meta & COUNT_MASK & ~LOG_UNIQUE;  // (the slot may be a log entry: drop its mark)
		uint64_t cov;
		if (BLOOM) {
			uint64_t K0, K1;
			from_internal<W>(v.k0, v.k1, K0, K1);
			const uint64_t origin = hash2<W>(K0, K1);
			bool fp = true;  // all 8 bits were set before this key first arrived
			for (int j = 0; j < 8 && fp; ++j) {
				uint64_t b = (origin ^ BLOOM_MAGIC[j]) % m_bits;
				if (bitstamp[b] == v.stamp) fp = false;  // this key's first arrival set the bit: one proof is enough
			}
			if (fp) cov = N + 1;
			else {
				inserted++;
				cov = (N >= 2) ? N : 0;
			}
		} else {
			cov = N;
		}
		if (cov > 0xFFFFFFFFull) cov = 0xFFFFFFFFull;
		if (cov) retained++;
		tab[i].meta = admitted_meta(v.meta, cov);
		hist_add(s_h, cov, hist_dense, hist_ovf, counters);
	}
	for (int d = 32; d > 0; d >>= 1) {
		retained += __shfl_down(retained, d);
		inserted += __shfl_down(inserted, d);
	}
	if ((threadIdx.x & 63) == 0) {
		if (retained) atomicAdd(&counters[C_RETAINED], (unsigned long long)retained);
		if (inserted) atomicAdd(&counters[C_INSERTIONS], (unsigned long long)inserted);
	}
	__syncthreads();
	for (uint32_t j = threadIdx.x; j < HIST_LDS; j += blockDim.x)
		if (s_h[j]) atomicAdd(&hist_dense[j], (unsigned long long)s_h[j]);
}

// A7 with the "lost" marks of rg_bloom.h: a key whose 8 bit checks all found the bit set earlier is a
// false positive
// WRITE: the coverage goes into the slot (admission in place); otherwise the pass only counts -- vertices stored, Bloom
// insertions, the coverage histogram and, per top byte of the slot hash, the admitted entries (first_hist: the first
// level of the table partition, rg_table.h) -- and the kernels that read the log next apply the rule again.
// STATS == false: only the write (the fallback after a counting pass).
template <int W, bool WRITE = true, bool STATS = true>
__global__ __launch_bounds__(BLOCK) void k_admit_lost(Slot *__restrict__ tab, uint64_t cap, const uint32_t *__restrict__ lost,
                                                      unsigned long long *__restrict__ counters, unsigned long long *__restrict__ hist_dense,
                                                      uint32_t *__restrict__ hist_ovf, unsigned long long *__restrict__ first_hist = nullptr) {
	__shared__ uint32_t s_h[HIST_LDS];
	__shared__ uint32_t s_first[256];
	for (uint32_t j = threadIdx.x; j < HIST_LDS; j += blockDim.x) s_h[j] = 0;
	if (threadIdx.x < 256) s_first[threadIdx.x] = 0;
	__syncthreads();
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	uint32_t retained = 0, inserted = 0;
	for (; i < cap; i += stride) {
		Slot v = tab[i];
		if (v.k0 == EMPTY) continue;
		const uint64_t N = v.meta & COUNT_MASK & ~LOG_UNIQUE;  // (the slot may be a log entry: drop its mark)
		const uint32_t nlost = (lost[i >> 2] >> ((i & 3u) * 8u)) & 0xFFu;
		uint64_t cov;
		if (nlost == 8) cov = N + 1;
		else {
			inserted++;
			cov = (N >= 2) ? N : 0;
		}
		if (cov > 0xFFFFFFFFull) cov = 0xFFFFFFFFull;
		if (cov) retained++;
		if (WRITE) tab[i].meta = admitted_meta(v.meta, cov);
		if (STATS) hist_add(s_h, cov, hist_dense, hist_ovf, counters);
		if (!WRITE && cov) atomicAdd(&s_first[(uint32_t)(slot_hash<W>(v.k0, W == 2 ? v.k1 : 0) >> 56)], 1u);
	}
	if (!STATS) return;
	for (int d = 32; d > 0; d >>= 1) {
		retained += __shfl_down(retained, d);
		inserted += __shfl_down(inserted, d);
	}
	if ((threadIdx.x & 63) == 0) {
		if (retained) atomicAdd(&counters[C_RETAINED], (unsigned long long)retained);
		if (inserted) atomicAdd(&counters[C_INSERTIONS], (unsigned long long)inserted);
	}
	__syncthreads();
	for (uint32_t j = threadIdx.x; j < HIST_LDS; j += blockDim.x)
		if (s_h[j]) atomicAdd(&hist_dense[j], (unsigned long long)s_h[j]);
	if (!WRITE && first_hist && threadIdx.x < 256 && s_first[threadIdx.x]) atomicAdd(&first_hist[threadIdx.x], (unsigned long long)s_first[threadIdx.x]);
}

// A12: coverage histogram of the stored pairs.  Low coverages are binned in LDS per workgroup.
__global__ __launch_bounds__(BLOCK) void k_histogram(const Slot *__restrict__ tab, uint64_t cap,
                                                     unsigned long long *__restrict__ dense, uint32_t *__restrict__ ovf,
                                                     unsigned long long *__restrict__ counters) {
	__shared__ uint32_t s_h[HIST_LDS];
	for (uint32_t j = threadIdx.x; j < HIST_LDS; j += blockDim.x) s_h[j] = 0;
	__syncthreads();
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (; i < cap; i += stride) {
		ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(&tab[i].k0);
		if (a.x == EMPTY) continue;
		uint64_t cov = tab[i].meta & COUNT_MASK;
		if (cov == 0) continue;
		if (cov < HIST_LDS) atomicAdd(&s_h[cov], 1u);
		else if (cov < HIST_DENSE) atomicAdd(&dense[cov], 1ull);
		else {
			unsigned long long pos = atomicAdd(&counters[C_OVF], 1ull);
			if (pos < HIST_OVF_CAP) ovf[pos] = (uint32_t)cov;
		}
	}
	__syncthreads();
	for (uint32_t j = threadIdx.x; j < HIST_LDS; j += blockDim.x)
		if (s_h[j]) atomicAdd(&dense[j], (unsigned long long)s_h[j]);
}


// multiword helpers in the reference key form
template <int W>
__device__ __forceinline__ void canonical_internal(uint64_t f0, uint64_t f1, int k, uint64_t &i0, uint64_t &i1, bool &is_rc) {
	uint64_t r0, r1;
	revcomp<W>(f0, f1, k, r0, r1);
	is_rc = kmer_less<W>(r0, r1, f0, f1);
	to_internal<W>(is_rc ? r0 : f0, is_rc ? r1 : f1, i0, i1);
}

// EdgePurgerWorker::work (EdgePurger/EdgePurgerWorker.cpp:28-102): drop an edge bit when the
// neighbour it names is not stored.  Neighbours as Kmer::getIngoingEdges/getOutgoingEdges
// (Kmer.cpp:176-282).  Reads only coverage of other slots, writes only its own edge byte.
template <int W>
__global__ __launch_bounds__(BLOCK) void k_purge(Slot *__restrict__ tab, uint64_t cap, int k) {
	const uint64_t mask0 = (2 * k >= 64) ? ~0ull : ((1ull << (2 * k)) - 1);
	const uint64_t mask1 = (k > 32) ? ((1ull << (2 * k - 64)) - 1) : 0ull;
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (; i < cap; i += stride) {
		Slot v = tab[i];
		if (v.k0 == EMPTY || (v.meta & COUNT_MASK) == 0) continue;
		uint32_t e = (uint32_t)(v.meta >> PUB_SHIFT) & 0xFFu;
		if (!e) continue;
		uint64_t K0, K1;
		from_internal<W>(v.k0, v.k1, K0, K1);
		uint32_t keep = e;
		// a lane walks its own set bits (two on a linear stretch of the genome), so that the lanes of a wavefront look their
		// j-th neighbours up together; a loop over the eight bits ran every look-up with a quarter of the lanes (14.5 ms on C2)
		for (uint32_t rem = e; rem; rem &= rem - 1) {
			const uint32_t bit = (uint32_t)__builtin_ctz(rem), c = bit & 3u;
			uint64_t n0, n1 = 0;
			if (bit < 4) {  // parent: shift left one base, base c enters at position 0
				n0 = ((K0 << 2) | c) & mask0;
				if (W == 2) n1 = ((K1 << 2) | (K0 >> 62)) & mask1;
			} else {  // child: shift right one base, base c enters at position k-1
				n0 = K0 >> 2;
				if (W == 2) {
					n0 |= K1 << 62;
					n1 = K1 >> 2;
				}
				const int pos = 2 * (k - 1);
				if (pos < 64) n0 |= (uint64_t)c << pos;
				else n1 |= (uint64_t)c << (pos - 64);
			}
			uint64_t i0, i1;
			bool rc;
			canonical_internal<W>(n0, n1, k, i0, i1, rc);
			const Slot *q = lookup<W>(tab, cap, i0, i1);
			if (!q || (q->meta & COUNT_MASK) == 0) keep &= ~(1u << bit);
		}
		if (keep != e) tab[i].meta = (v.meta & ~(0xFFull << PUB_SHIFT)) | ((uint64_t)keep << PUB_SHIFT);
	}
}

// ---- read ingest: Read::constructor with Read::trim (SequencesLoader/Read.cpp:31-159) in bulk --------
__device__ __forceinline__ bool is_acgt_any_case(uint8_t c) {
	c &= 0xDFu;  // a,c,g,t -> A,C,G,T (Read.cpp:35-44); no other byte maps onto these four
	return c == 'A' || c == 'C' || c == 'G' || c == 'T';
}
// one thread per raw sequence: where its first and last ACGT symbols are (Read.cpp:45-66)
__global__ __launch_bounds__(BLOCK) void k_trim_reads(const uint8_t *__restrict__ text, const uint64_t *__restrict__ starts,
                                                      const uint32_t *__restrict__ raw_len, uint64_t n,
                                                      uint64_t *__restrict__ tstart, uint32_t *__restrict__ tlen,
                                                      unsigned long long *__restrict__ nbytes, uint64_t text_bytes) {
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (; i < n; i += stride) {
		// a sequence that does not lie inside the text, or is too long to store (uint16_t m_length), is read as empty:
		// nothing outside [text, text + text_bytes) is ever touched (rg_add_reads_text rejects such input on the host;
		// rg_pack_reads_noalloc, whose arrays live on the device, relies on this)
		const uint64_t st = starts[i];
		const bool inside = raw_len[i] < 65535u && st <= text_bytes && (uint64_t)raw_len[i] <= text_bytes - st;
		const uint8_t *p = text + (inside ? st : 0);
		const uint32_t len = inside ? raw_len[i] : 0u;
		uint32_t first = 0;
		while (first < len && !is_acgt_any_case(p[first])) first++;
		uint32_t last = first;  // one past the last ACGT symbol
		for (uint32_t j = len; j > first; --j)
			if (is_acgt_any_case(p[j - 1])) {
				last = j;
				break;
			}
		tstart[i] = (inside ? st : 0) + first;
		tlen[i] = last - first;
		nbytes[i] = (last - first + 3) / 4;  // Read::getRequiredBytes (Read.cpp:206-220)
	}
}
// 64 reads per workgroup, one thread per output byte: 4 symbols -> 2 bits each, anything that is not
// C, G or T encodes as A (Read.cpp:117-141)
__global__ __launch_bounds__(BLOCK) void k_pack_reads(const uint8_t *__restrict__ text, const uint64_t *__restrict__ tstart,
                                                      const uint32_t *__restrict__ tlen, const unsigned long long *__restrict__ off,
                                                      uint64_t n, uint8_t *__restrict__ out, uint64_t out_cap) {
	__shared__ unsigned long long s_off[65];
	const uint64_t r0 = (uint64_t)blockIdx.x * 64;
	const uint32_t nr = (n - r0 < 64) ? (uint32_t)(n - r0) : 64u;
	if (threadIdx.x <= nr) s_off[threadIdx.x] = off[r0 + threadIdx.x];
	__syncthreads();
	const unsigned long long b0 = s_off[0], b1 = s_off[nr];
	for (unsigned long long b = b0 + threadIdx.x; b < b1; b += blockDim.x) {
		uint32_t lo = 0, hi = nr;  // largest r with s_off[r] <= b
		while (hi - lo > 1) {
			const uint32_t mid = (lo + hi) >> 1;
			if (s_off[mid] <= b) lo = mid;
			else hi = mid;
		}
		const uint64_t r = r0 + lo;
		const uint32_t j = (uint32_t)(b - s_off[lo]) * 4;
		const uint32_t len = tlen[r];
		const uint8_t *p = text + tstart[r];
		uint32_t v = 0;
#pragma unroll
		for (uint32_t q = 0; q < 4; ++q)
			if (j + q < len) {
				const uint8_t c = p[j + q] & 0xDFu;
				const uint32_t code = (c == 'C') ? 1u : (c == 'G') ? 2u : (c == 'T') ? 3u : 0u;
				v |= code << (2 * q);
			}
		if (b < out_cap) out[b] = (uint8_t)v;  // out_cap: the caller's buffer (rg_pack_reads_noalloc); ~0 when the library sized it
	}
}

// ---- SequencesIndexer markers (IndexerWorker.cpp:67-248,328-362; build without CONFIG_USE_COVERAGE_DISTRIBUTION) --
// One wavefront per read.  The threshold is 2*v for the v in 10..1 that maximises #(coverage <= v) +
// #(coverage >= 2v) (ties: the larger v); the forward marker is the first position whose coverage reaches
// it, the reverse marker the last one, stored as length-k-position.  All ten candidates are tracked in one
// sweep over the read, so every k-mer is looked up once.
template <int W>
__global__ __launch_bounds__(256) void k_read_markers(const uint64_t *__restrict__ P, const uint64_t *__restrict__ offsets,
                                                      const uint32_t *__restrict__ lengths, uint64_t nreads, int k,
                                                      const Slot *__restrict__ tab, uint64_t cap,
                                                      int32_t *__restrict__ forward, int32_t *__restrict__ reverse) {
	__shared__ uint32_t s_le[4][10], s_ge[4][10], s_first[4][10], s_last[4][10];
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint64_t mask0 = (2 * k >= 64) ? ~0ull : ((1ull << (2 * k)) - 1);
	const uint64_t mask1 = (k > 32) ? ((1ull << (2 * k - 64)) - 1) : 0ull;
	const uint64_t nwaves = (uint64_t)gridDim.x * 4;
	for (uint64_t r = (uint64_t)blockIdx.x * 4 + wave; r < nreads; r += nwaves) {
		const uint32_t L = lengths[r];
		const uint32_t n = (L >= (uint32_t)k) ? L - (uint32_t)k + 1 : 0;
		if (lane < 10) {
			s_le[wave][lane] = 0;
			s_ge[wave][lane] = 0;
			s_first[wave][lane] = 0xFFFFFFFFu;
			s_last[wave][lane] = 0;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		const uint64_t bit0 = offsets[r] * 8;
		uint32_t le[10], ge[10], first[10], last1[10];  // last1 = last position + 1 (0 = none)
#pragma unroll
		for (int v = 0; v < 10; ++v) {
			le[v] = 0;
			ge[v] = 0;
			first[v] = 0xFFFFFFFFu;
			last1[v] = 0;
		}
		for (uint32_t p = lane; p < n; p += 64) {
			uint64_t f0 = ext64(P, bit0 + 2ull * p) & mask0, f1 = 0;
			if (W == 2 && k > 32) f1 = ext64(P, bit0 + 2ull * p + 64) & mask1;
			uint64_t i0, i1;
			bool rc;
			canonical_internal<W>(f0, f1, k, i0, i1, rc);
			const Slot *q = lookup<W>(tab, cap, i0, i1);
			// the reference compares the coverage as an int (IndexerWorker.cpp:92)
			const int cov = q ? (int)(uint32_t)(q->meta & COUNT_MASK) : 0;
#pragma unroll
			for (int v = 0; v < 10; ++v) {
				if (cov <= v + 1) le[v]++;
				if (cov >= 2 * (v + 1)) {
					ge[v]++;
					if (first[v] == 0xFFFFFFFFu) first[v] = p;
					last1[v] = p + 1;
				}
			}
		}
#pragma unroll
		for (int v = 0; v < 10; ++v) {
			if (le[v]) atomicAdd(&s_le[wave][v], le[v]);
			if (ge[v]) {
				atomicAdd(&s_ge[wave][v], ge[v]);
				atomicMin(&s_first[wave][v], first[v]);
				atomicMax(&s_last[wave][v], last1[v]);
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		if (lane == 0) {
			int bestEntry = -1;
			uint32_t bestScore = 0;
			for (int value = 10; value >= 1; --value) {
				const uint32_t score = s_le[wave][value - 1] + s_ge[wave][value - 1];
				if (bestEntry == -1 || score > bestScore) {
					bestEntry = value;
					bestScore = score;
				}
			}
			const uint32_t fp = s_first[wave][bestEntry - 1], lp1 = s_last[wave][bestEntry - 1];
			forward[r] = (fp == 0xFFFFFFFFu) ? -1 : (int32_t)fp;
			reverse[r] = (lp1 == 0) ? -1 : (int32_t)(L - (uint32_t)k - (lp1 - 1));
		}
		__builtin_amdgcn_wave_barrier();
	}
}

// ---- markers across ranks: the coverage requests of IndexerWorker.cpp:76-96 in bulk ---------------------
// WRITE == false: counts[owner] += k-mers of the local reads that owner must answer for; WRITE == true: canonical keys
// (internal form) written at cursors[owner]++, and qidx[kprefix[read] + position] = where that k-mer's query went
// (kprefix = exclusive scan of the reads' k-mer counts: memory is per k-mer that exists, not per read x longest read)
template <int W, bool WRITE>
__global__ __launch_bounds__(256) void k_marker_queries(const uint64_t *__restrict__ P, const uint64_t *__restrict__ offsets,
                                                        const uint32_t *__restrict__ lengths, uint64_t nreads, int k, uint32_t size,
                                                        unsigned long long *__restrict__ counts_or_cursors, uint64_t *__restrict__ keys,
                                                        const unsigned long long *__restrict__ kprefix, uint64_t *__restrict__ qidx) {
	extern __shared__ unsigned long long s_dyn[];  // [size] global bases (WRITE), then [size] u32 counts
	unsigned long long *s_base = s_dyn;
	uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_dyn + size);
	__shared__ uint32_t s_nmax;
	for (uint32_t j = threadIdx.x; j < size; j += 256) s_cnt[j] = 0;
	if (threadIdx.x == 0) s_nmax = 0;
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint64_t mask0 = (2 * k >= 64) ? ~0ull : ((1ull << (2 * k)) - 1);
	const uint64_t mask1 = (k > 32) ? ((1ull << (2 * k - 64)) - 1) : 0ull;
	// four reads per step, one per wavefront; in WRITE mode every 256 k-mers reserve their places with one global atomic
	// per owner (the ranks inside come from LDS), so the loops are uniform over the workgroup
	for (uint64_t r0 = (uint64_t)blockIdx.x * 4; r0 < nreads; r0 += (uint64_t)gridDim.x * 4) {
		const uint64_t r = r0 + wave;
		const uint32_t L = (r < nreads) ? lengths[r] : 0u;
		const uint32_t n = (L >= (uint32_t)k) ? L - (uint32_t)k + 1 : 0;
		const uint64_t bit0 = (r < nreads) ? offsets[r] * 8 : 0;
		if (WRITE) {
			if (lane == 0 && n) atomicMax(&s_nmax, n);
			__syncthreads();
		}
		const uint32_t nmax = WRITE ? s_nmax : n;
		for (uint32_t p0 = 0; p0 < nmax; p0 += 64) {
			const uint32_t p = p0 + lane;
			uint64_t i0 = 0, i1 = 0;
			uint32_t owner = 0, rank = 0;
			const bool have = p < n;
			if (have) {
				uint64_t f0 = ext64(P, bit0 + 2ull * p) & mask0, f1 = 0;
				if (W == 2 && k > 32) f1 = ext64(P, bit0 + 2ull * p + 64) & mask1;
				bool rc;
				canonical_internal<W>(f0, f1, k, i0, i1, rc);
				uint64_t K0, K1;
				from_internal<W>(i0, i1, K0, K1);
				owner = (uint32_t)(hash1<W>(K0, K1) % (uint64_t)size);
				rank = atomicAdd(&s_cnt[owner], 1u);
			}
			if (WRITE) {
				__syncthreads();
				for (uint32_t j = threadIdx.x; j < size; j += 256) {
					const uint32_t c = s_cnt[j];
					if (c) s_base[j] = atomicAdd(&counts_or_cursors[j], (unsigned long long)c);
				}
				__syncthreads();
				if (have) {
					const unsigned long long pos = s_base[owner] + rank;
					keys[pos * W] = i0;
					if (W == 2) keys[pos * W + 1] = i1;
					qidx[kprefix[r] + p] = pos;
				}
				__syncthreads();
				for (uint32_t j = threadIdx.x; j < size; j += 256) s_cnt[j] = 0;
				if (threadIdx.x == 0) s_nmax = 0;
				__syncthreads();
			}
		}
		if (WRITE && nmax == 0) {  // s_nmax was read by everybody before anybody can raise it for the next step
			__syncthreads();
		}
	}
	if (!WRITE) {
		__syncthreads();
		for (uint32_t j = threadIdx.x; j < size; j += 256)
			if (s_cnt[j]) atomicAdd(&counts_or_cursors[j], (unsigned long long)s_cnt[j]);
	}
}

// the owner's answer: the coverage, 0 for a k-mer that is not stored (MessageProcessor.cpp:1224-1256)
template <int W>
__global__ __launch_bounds__(BLOCK) void k_marker_answer(const Slot *__restrict__ tab, uint64_t cap, const uint64_t *__restrict__ keys, uint64_t n,
                                                         uint32_t *__restrict__ coverage) {
	uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (; q < n; q += stride) {
		const Slot *p = lookup<W>(tab, cap, keys[q * W], W == 2 ? keys[q * W + 1] : 0);
		coverage[q] = p ? (uint32_t)(p->meta & COUNT_MASK) : 0u;
	}
}

// k-mers of every read (0 for reads shorter than k), the input of the scan that gives kprefix
__global__ __launch_bounds__(BLOCK) void k_marker_kmer_counts(const uint32_t *__restrict__ lengths, uint64_t n, int k, unsigned long long *__restrict__ out) {
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (; i < n; i += stride) {
		const uint32_t L = lengths[i];
		out[i] = (L >= (uint32_t)k) ? (unsigned long long)(L - (uint32_t)k + 1) : 0ull;
	}
}

// k_read_markers with the coverages given (one wavefront per read; see there for the rule)
// answers[qidx[kprefix[r] + p]] = coverage of k-mer p of read r, as the owners returned it
__global__ __launch_bounds__(256) void k_markers_from_coverage(const uint32_t *__restrict__ lengths, uint64_t nreads, int k,
                                                               const uint32_t *__restrict__ answers, const unsigned long long *__restrict__ kprefix,
                                                               const uint64_t *__restrict__ qidx, int32_t *__restrict__ forward,
                                                               int32_t *__restrict__ reverse) {
	__shared__ uint32_t s_le[4][10], s_ge[4][10], s_first[4][10], s_last[4][10];
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint64_t nwaves = (uint64_t)gridDim.x * 4;
	for (uint64_t r = (uint64_t)blockIdx.x * 4 + wave; r < nreads; r += nwaves) {
		const uint32_t L = lengths[r];
		const uint32_t n = (L >= (uint32_t)k) ? L - (uint32_t)k + 1 : 0;
		if (lane < 10) {
			s_le[wave][lane] = 0;
			s_ge[wave][lane] = 0;
			s_first[wave][lane] = 0xFFFFFFFFu;
			s_last[wave][lane] = 0;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		uint32_t le[10], ge[10], first[10], last1[10];
#pragma unroll
		for (int v = 0; v < 10; ++v) {
			le[v] = 0;
			ge[v] = 0;
			first[v] = 0xFFFFFFFFu;
			last1[v] = 0;
		}
		for (uint32_t p = lane; p < n; p += 64) {
			const int cov = (int)answers[qidx[kprefix[r] + p]];  // the reference compares the coverage as an int (IndexerWorker.cpp:92)
#pragma unroll
			for (int v = 0; v < 10; ++v) {
				if (cov <= v + 1) le[v]++;
				if (cov >= 2 * (v + 1)) {
					ge[v]++;
					if (first[v] == 0xFFFFFFFFu) first[v] = p;
					last1[v] = p + 1;
				}
			}
		}
#pragma unroll
		for (int v = 0; v < 10; ++v) {
			if (le[v]) atomicAdd(&s_le[wave][v], le[v]);
			if (ge[v]) {
				atomicAdd(&s_ge[wave][v], ge[v]);
				atomicMin(&s_first[wave][v], first[v]);
				atomicMax(&s_last[wave][v], last1[v]);
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		if (lane == 0) {
			int bestEntry = -1;
			uint32_t bestScore = 0;
			for (int value = 10; value >= 1; --value) {
				const uint32_t score = s_le[wave][value - 1] + s_ge[wave][value - 1];
				if (bestEntry == -1 || score > bestScore) {
					bestEntry = value;
					bestScore = score;
				}
			}
			const uint32_t fp = s_first[wave][bestEntry - 1], lp1 = s_last[wave][bestEntry - 1];
			forward[r] = (fp == 0xFFFFFFFFu) ? -1 : (int32_t)fp;
			reverse[r] = (lp1 == 0) ? -1 : (int32_t)(L - (uint32_t)k - (lp1 - 1));
		}
		__builtin_amdgcn_wave_barrier();
	}
}

// Kmer::hash_function_1 / _2 as the kernels compute them, for rg_hash_kmers (keys = the lower k-mer's words, W per key)
template <int W>
__global__ __launch_bounds__(BLOCK) void k_hash_kmers(const uint64_t *__restrict__ keys, uint64_t n, uint64_t *__restrict__ h1, uint64_t *__restrict__ h2) {
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint64_t K0 = keys[i * W], K1 = (W == 2) ? keys[i * W + 1] : 0;
	h1[i] = hash1<W>(K0, K1);
	h2[i] = hash2<W>(K0, K1);
}

// Vertex::load for a whole checkpoint (Vertex.cpp:422-436, MessageProcessor.cpp:596-655): records in Vertex::dump
// layout go straight into the table with their coverage and edge byte
template <int W>
__global__ __launch_bounds__(BLOCK) void k_import(const uint8_t *__restrict__ rec, uint64_t n, Slot *__restrict__ tab, uint64_t cap,
                                                  unsigned long long *__restrict__ counters) {
	constexpr int REC = W * 8 + 5;
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	bool failed = false;
	uint32_t claimed = 0;
	for (; i < n; i += stride) {
		const uint8_t *p = rec + i * REC;
		uint64_t K0 = 0, K1 = 0;
		for (int b = 0; b < 8; ++b) K0 |= (uint64_t)p[b] << (8 * b);
		if (W == 2)
			for (int b = 0; b < 8; ++b) K1 |= (uint64_t)p[8 + b] << (8 * b);
		uint64_t cov = 0;
		for (int b = 0; b < 4; ++b) cov |= (uint64_t)p[W * 8 + b] << (8 * b);
		const uint64_t e = p[W * 8 + 4];
		uint64_t i0, i1;
		to_internal<W>(K0, K1, i0, i1);
		// the keys of a checkpoint are distinct: claim a free slot and store the payload plainly (as k_rehash does)
		uint64_t sl = slot_home(slot_hash<W>(i0, W == 2 ? i1 : 0), cap);
		bool done = false;
		for (uint32_t probe = 0; probe < MAX_PROBE && !done; ++probe) {
			Slot *q = tab + sl;
			const uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long *>(&q->k0), EMPTY, i0);
			if (prev == EMPTY) {
				q->k1 = (W == 2) ? i1 : 0;
				q->stamp = 0;
				q->meta = cov | (e << PUB_SHIFT);
				claimed++;
				done = true;
			}
			sl = slot_next(sl, cap);
		}
		if (!done) failed = true;
	}
	if (claimed) atomicAdd(&counters[C_FILL], (unsigned long long)claimed);
	if (failed) atomicAdd(&counters[C_ERR], 1ull);
}

// admitted log entries -> vertex table.  The entries are distinct keys whose coverage and edge bits are final
// (the admission ran on the log itself), so a slot is claimed and the payload stored plainly; entries that were
// not admitted (coverage 0) never reach the table -- like the reference, whose Bloom filter stands before its table.
template <int W>
__global__ __launch_bounds__(BLOCK) void k_insert_admitted(const Slot *__restrict__ log, uint64_t n, Slot *__restrict__ tab, uint64_t cap,
                                                           unsigned long long *__restrict__ counters) {
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	bool failed = false;
	uint32_t claimed = 0;
	for (; i < n; i += stride) {
		const Slot v = log[i];
		if (v.k0 == EMPTY || (v.meta & COUNT_MASK) == 0) continue;
		uint64_t sl = slot_home(slot_hash<W>(v.k0, W == 2 ? v.k1 : 0), cap);
		bool done = false;
		for (uint32_t probe = 0; probe < MAX_PROBE && !done; ++probe) {
			Slot *q = tab + sl;
			const uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long *>(&q->k0), EMPTY, v.k0);
			if (prev == EMPTY) {
				q->k1 = v.k1;
				*reinterpret_cast<ulonglong2 *>(&q->stamp) = make_ulonglong2(v.stamp, v.meta);
				claimed++;
				done = true;
			}
			sl = slot_next(sl, cap);
		}
		if (!done) failed = true;
	}
	for (int d = 32; d > 0; d >>= 1) claimed += __shfl_down(claimed, d);
	if ((threadIdx.x & 63) == 0 && claimed) atomicAdd(&counters[C_FILL], (unsigned long long)claimed);
	if (failed) atomicAdd(&counters[C_ERR], 1ull);
}

// ---- purge across ranks: the neighbour-existence queries of EdgePurgerWorker.cpp:36-44,73-81 in bulk ------
// neighbour (in canonical internal form) named by edge bit `bit` of the stored key (K0,K1)
template <int W>
__device__ __forceinline__ void edge_neighbour(uint64_t K0, uint64_t K1, uint32_t bit, int k, uint64_t mask0, uint64_t mask1,
                                               uint64_t &i0, uint64_t &i1) {
	uint64_t n0, n1 = 0;
	const uint64_t c = bit & 3u;
	if (bit < 4) {  // parent: shift left one base, base c enters at position 0
		n0 = ((K0 << 2) | c) & mask0;
		if (W == 2) n1 = ((K1 << 2) | (K0 >> 62)) & mask1;
	} else {  // child: shift right one base, base c enters at position k-1
		n0 = K0 >> 2;
		if (W == 2) {
			n0 |= K1 << 62;
			n1 = K1 >> 2;
		}
		const int pos = 2 * (k - 1);
		if (pos < 64) n0 |= c << pos;
		else n1 |= c << (pos - 64);
	}
	bool rc;
	canonical_internal<W>(n0, n1, k, i0, i1, rc);
}

// WRITE == false: counts[owner] += queries; WRITE == true: keys/origin written at cursors[owner]++.
// Up to PURGE_LDS_RANKS ranks a workgroup counts its queries per owner in LDS first and claims its places with one global
// atomic per owner and BLOCK slots (one per query would be 2e8 atomics on `size` addresses for C2's graph); a lane walks its
// own set edge bits, so the lanes of a wavefront work on their j-th neighbours together.
static constexpr uint32_t PURGE_LDS_RANKS = 1024;
template <int W, bool WRITE>
__global__ __launch_bounds__(BLOCK) void k_purge_queries(const Slot *__restrict__ tab, uint64_t cap, int k, uint32_t size,
                                                         unsigned long long *__restrict__ counts_or_cursors,
                                                         uint64_t *__restrict__ keys, uint64_t *__restrict__ origin) {
	__shared__ uint32_t s_cnt[PURGE_LDS_RANKS];
	__shared__ unsigned long long s_base[PURGE_LDS_RANKS];
	const bool in_lds = size <= PURGE_LDS_RANKS;  // uniform
	const uint64_t mask0 = (2 * k >= 64) ? ~0ull : ((1ull << (2 * k)) - 1);
	const uint64_t mask1 = (k > 32) ? ((1ull << (2 * k - 64)) - 1) : 0ull;
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	if (in_lds)
		for (uint32_t r = threadIdx.x; r < size; r += blockDim.x) s_cnt[r] = 0;
	__syncthreads();
	for (uint64_t i0s = (uint64_t)blockIdx.x * blockDim.x; i0s < cap; i0s += stride) {  // uniform bounds: barriers inside
		const uint64_t i = i0s + threadIdx.x;
		uint32_t e = 0;
		uint64_t K0 = 0, K1 = 0;
		if (i < cap) {
			const Slot v = tab[i];
			if (v.k0 != EMPTY && (v.meta & COUNT_MASK) != 0) {
				e = (uint32_t)(v.meta >> PUB_SHIFT) & 0xFFu;
				from_internal<W>(v.k0, v.k1, K0, K1);
			}
		}
		auto owner_of = [&](uint32_t bit, uint64_t &q0, uint64_t &q1) -> uint32_t {
			edge_neighbour<W>(K0, K1, bit, k, mask0, mask1, q0, q1);
			uint64_t N0, N1;
			from_internal<W>(q0, q1, N0, N1);
			return (uint32_t)(hash1<W>(N0, N1) % (uint64_t)size);
		};
		if (!in_lds) {
			for (uint32_t rem = e; rem; rem &= rem - 1) {
				const uint32_t bit = (uint32_t)__builtin_ctz(rem);
				uint64_t q0, q1;
				const uint32_t owner = owner_of(bit, q0, q1);
				const unsigned long long pos = atomicAdd(&counts_or_cursors[owner], 1ull);
				if (WRITE) {
					keys[pos * W] = q0;
					if (W == 2) keys[pos * W + 1] = q1;
					origin[pos] = (i << 3) | bit;
				}
			}
			continue;
		}
		for (uint32_t rem = e; rem; rem &= rem - 1) {
			uint64_t q0, q1;
			atomicAdd(&s_cnt[owner_of((uint32_t)__builtin_ctz(rem), q0, q1)], 1u);
		}
		if (!WRITE) continue;  // counts: flushed once, below
		__syncthreads();
		for (uint32_t r = threadIdx.x; r < size; r += blockDim.x) {
			const uint32_t c = s_cnt[r];
			s_base[r] = c ? atomicAdd(&counts_or_cursors[r], (unsigned long long)c) : 0ull;
			s_cnt[r] = 0;
		}
		__syncthreads();
		for (uint32_t rem = e; rem; rem &= rem - 1) {
			const uint32_t bit = (uint32_t)__builtin_ctz(rem);
			uint64_t q0, q1;
			const uint32_t owner = owner_of(bit, q0, q1);
			const unsigned long long pos = s_base[owner] + atomicAdd(&s_cnt[owner], 1u);
			keys[pos * W] = q0;
			if (W == 2) keys[pos * W + 1] = q1;
			origin[pos] = (i << 3) | bit;
		}
		__syncthreads();
		for (uint32_t r = threadIdx.x; r < size; r += blockDim.x) s_cnt[r] = 0;
		__syncthreads();
	}
	if (in_lds && !WRITE) {
		__syncthreads();
		for (uint32_t r = threadIdx.x; r < size; r += blockDim.x)
			if (s_cnt[r]) atomicAdd(&counts_or_cursors[r], (unsigned long long)s_cnt[r]);
	}
}

// the owner's side of RAY_MPI_TAG_REQUEST_VERTEX_COVERAGE (MessageProcessor.cpp:1224-1256): 1 = stored, 0 = not
template <int W>
__global__ __launch_bounds__(BLOCK) void k_purge_answer(const Slot *__restrict__ tab, uint64_t cap, const uint64_t *__restrict__ keys,
                                                        uint64_t n, uint8_t *__restrict__ answers) {
	uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (; q < n; q += stride) {
		const Slot *p = lookup<W>(tab, cap, keys[q * W], W == 2 ? keys[q * W + 1] : 0);
		answers[q] = (p && (p->meta & COUNT_MASK) != 0) ? 1 : 0;
	}
}

// coverage 0: the edge is deleted (EdgePurgerWorker.cpp:54-56,91-93)
__global__ __launch_bounds__(BLOCK) void k_purge_apply(Slot *__restrict__ tab, const uint64_t *__restrict__ origin,
                                                       const uint8_t *__restrict__ answers, uint64_t n) {
	uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (; q < n; q += stride)
		if (!answers[q]) {
			const uint64_t o = origin[q];
			atomicAnd(reinterpret_cast<unsigned long long *>(&tab[o >> 3].meta), ~(1ull << (PUB_SHIFT + (o & 7u))));
		}
}

// bulk GridTable::find + Vertex::getCoverage/getEdges
template <int W>
__global__ __launch_bounds__(BLOCK) void k_find(const Slot *__restrict__ tab, uint64_t cap, int k, const uint64_t *__restrict__ q,
                                                uint64_t n, uint32_t *__restrict__ cov, uint8_t *__restrict__ edges, uint32_t edge_mask) {
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (; i < n; i += stride) {
		uint64_t f0 = q[i * W], f1 = (W == 2) ? q[i * W + 1] : 0;
		uint64_t i0, i1;
		bool rc;
		canonical_internal<W>(f0, f1, k, i0, i1, rc);
		const Slot *s = lookup<W>(tab, cap, i0, i1);
		uint32_t c = 0, e = 0;
		if (s) {
			c = (uint32_t)(s->meta & COUNT_MASK);
			if (c) {
				e = (uint32_t)(s->meta >> PUB_SHIFT) & edge_mask;  // 0 until rg_add_edges: pass 2 has not happened yet
				if (rc) e = convert_bitmap(e);
			}
		}
		cov[i] = c;
		edges[i] = (uint8_t)e;
	}
}

// 5x5 in/out-degree table over both strands (MachineHelper.cpp:548-578)
__global__ __launch_bounds__(BLOCK) void k_degrees(const Slot *__restrict__ tab, uint64_t cap, unsigned long long *__restrict__ out,
                                                   uint32_t edge_mask) {
	__shared__ uint32_t s_d[25];
	if (threadIdx.x < 25) s_d[threadIdx.x] = 0;
	__syncthreads();
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (; i < cap; i += stride) {
		uint64_t m = tab[i].meta;
		if (tab[i].k0 == EMPTY || (m & COUNT_MASK) == 0) continue;
		uint32_t e = (uint32_t)(m >> PUB_SHIFT) & edge_mask;
		uint32_t parents = __popc(e & 0xFu), children = __popc(e >> 4);
		atomicAdd(&s_d[parents * 5 + children], 1u);
		atomicAdd(&s_d[children * 5 + parents], 1u);  // the reverse complement swaps the roles
	}
	__syncthreads();
	if (threadIdx.x < 25 && s_d[threadIdx.x]) atomicAdd(&out[threadIdx.x], (unsigned long long)s_d[threadIdx.x]);
}

// export, step 1: stored pairs per chunk of slots
__global__ __launch_bounds__(BLOCK) void k_export_count(const Slot *__restrict__ tab, uint64_t cap, uint64_t chunk,
                                                        unsigned long long *__restrict__ per_chunk) {
	__shared__ uint32_t s_n;
	if (threadIdx.x == 0) s_n = 0;
	__syncthreads();
	uint64_t begin = (uint64_t)blockIdx.x * chunk, end = begin + chunk < cap ? begin + chunk : cap;
	uint32_t n = 0;
	for (uint64_t i = begin + threadIdx.x; i < end; i += blockDim.x)
		if (tab[i].k0 != EMPTY && (tab[i].meta & COUNT_MASK)) n++;
	if (n) atomicAdd(&s_n, n);
	__syncthreads();
	if (threadIdx.x == 0) per_chunk[blockIdx.x] = s_n;
}

// export, step 2: Vertex::dump records (Vertex.cpp:438-454) in slot order, deterministic
template <int W>
__global__ __launch_bounds__(BLOCK) void k_export_write(const Slot *__restrict__ tab, uint64_t cap, uint64_t chunk,
                                                        const unsigned long long *__restrict__ chunk_start, uint8_t *__restrict__ out,
                                                        uint32_t edge_mask) {
	constexpr int REC = W * 8 + 5;
	__shared__ uint32_t s_wave[BLOCK / 64];
	__shared__ uint32_t s_base;
	uint64_t begin = (uint64_t)blockIdx.x * chunk, end = begin + chunk < cap ? begin + chunk : cap;
	if (threadIdx.x == 0) s_base = 0;
	__syncthreads();
	const uint64_t out0 = chunk_start[blockIdx.x];
	for (uint64_t base = begin; base < end; base += blockDim.x) {
		uint64_t i = base + threadIdx.x;
		Slot v;
		bool keep = false;
		if (i < end) {
			v = tab[i];
			keep = v.k0 != EMPTY && (v.meta & COUNT_MASK);
		}
		unsigned long long ballot = __ballot(keep);
		uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
		uint32_t before = __popcll(ballot & ((1ull << lane) - 1));
		if (lane == 0) s_wave[wave] = __popcll(ballot);
		__syncthreads();
		uint32_t wbase = 0, tot = 0;
		for (uint32_t w = 0; w < BLOCK / 64; ++w) {
			if (w < wave) wbase += s_wave[w];
			tot += s_wave[w];
		}
		uint32_t blockbase = s_base;
		if (keep) {
			uint64_t K0, K1;
			from_internal<W>(v.k0, v.k1, K0, K1);
			uint8_t *p = out + (out0 + blockbase + wbase + before) * REC;
			uint32_t cov = (uint32_t)(v.meta & COUNT_MASK);
			uint8_t e = (uint8_t)((v.meta >> PUB_SHIFT) & edge_mask);
			for (int b = 0; b < 8; ++b) p[b] = (uint8_t)(K0 >> (8 * b));
			if (W == 2)
				for (int b = 0; b < 8; ++b) p[8 + b] = (uint8_t)(K1 >> (8 * b));
			for (int b = 0; b < 4; ++b) p[W * 8 + b] = (uint8_t)(cov >> (8 * b));
			p[W * 8 + 4] = e;
		}
		__syncthreads();
		if (threadIdx.x == 0) s_base = blockbase + tot;
		__syncthreads();
	}
}

// ================================================================================ host side

// ---- device memory pool -------------------------------------------------------------------------
// Freed blocks are kept per device and handed out again (best fit, at most 25 % larger than asked)
// instead of going back to the driver: releasing tens of GiB of touched VRAM costs hundreds of
// milliseconds per call on this stack, and a graph build allocates the same few sizes every time.
// rg_release_cached_memory() returns everything to the driver.
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
namespace {
struct Pool {
	std::mutex mu;
	std::multimap<uint64_t, void *> free_blocks[64];  // per device: size -> block
	std::map<void *, uint64_t> live[64];              // blocks handed out: ptr -> size
} g_pool;

hipError_t pool_alloc(int device, void **out, uint64_t bytes) {
	if (bytes == 0) bytes = 256;
	bytes = (bytes + 255) & ~255ull;
	device &= 63;
	{
		std::lock_guard<std::mutex> lk(g_pool.mu);
		auto &fb = g_pool.free_blocks[device];
		auto it = fb.lower_bound(bytes);
		if (it != fb.end() && it->first <= bytes + bytes / 4) {
			*out = it->second;
			g_pool.live[device][*out] = it->first;
			fb.erase(it);
			return hipSuccess;
		}
	}
	static const bool trace = getenv("RG_DEBUG_POOL") != nullptr;  // development: every trip to the driver, with its time
	auto t_start = std::chrono::steady_clock::now();
	hipError_t e = hipMalloc(out, bytes);
	if (trace) fprintf(stderr, "[raygraph] pool: hipMalloc %.3f GB -> %s, %.1f ms\n", bytes / 1e9, e == hipSuccess ? "ok" : "FULL", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
	if (e != hipSuccess) {
		// The device is full.  A cached block of up to three times the size is better than releasing memory (hipFree of a touched
		// multi-GiB block costs hundreds of milliseconds; a block many times the size is kept for the request it fits --
		// C5's share per GPU: 26 GB of records that sat in the 99 GB block left the log without room); failing that, cached
		// blocks go back to the driver, largest first, until the request fits.
		(void)hipGetLastError();
		{
			std::lock_guard<std::mutex> lk(g_pool.mu);
			auto &fb = g_pool.free_blocks[device];
			auto it = fb.lower_bound(bytes);
			// (up to three times since round 3: the 52.8 GB table of C5's share per GPU must be able to take the 106 GB block its log
			// left behind -- at "twice" it missed it by 0.4 %, and every step then paid 4.5 s for returning one of the two blocks
			// to the driver and mapping the other anew)
			if (it != fb.end() && it->first <= 3 * bytes) {
				*out = it->second;
				g_pool.live[device][*out] = it->first;
				fb.erase(it);
				return hipSuccess;
			}
		}
		for (;;) {
			void *victim = nullptr;
			{
				std::lock_guard<std::mutex> lk(g_pool.mu);
				auto &fb = g_pool.free_blocks[device];
				if (fb.empty()) break;
				auto last = std::prev(fb.end());
				victim = last->second;
				fb.erase(last);
			}
			auto t_free = std::chrono::steady_clock::now();
			(void)hipFree(victim);
			if (trace) fprintf(stderr, "[raygraph] pool: a cached block went back to the driver for a request of %.3f GB, %.1f ms\n", bytes / 1e9, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_free).count());
			auto t_again = std::chrono::steady_clock::now();
			e = hipMalloc(out, bytes);
			if (trace) fprintf(stderr, "[raygraph] pool: hipMalloc %.3f GB after that -> %s, %.1f ms\n", bytes / 1e9, e == hipSuccess ? "ok" : "FULL", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_again).count());
			if (e == hipSuccess) break;
			(void)hipGetLastError();
		}
		if (e != hipSuccess) {
			if (getenv("RG_DEBUG")) {
				size_t fr = 0, tot = 0;
				(void)hipMemGetInfo(&fr, &tot);
				uint64_t live = 0;
				std::lock_guard<std::mutex> lk(g_pool.mu);
				for (auto &kv : g_pool.live[device]) live += kv.second;
				fprintf(stderr, "[raygraph] pool: %.2f GB asked, %.2f GB live in %zu blocks, nothing cached, device reports %.2f of %.2f GB free\n", bytes / 1e9,
				        live / 1e9, g_pool.live[device].size(), fr / 1e9, tot / 1e9);
				for (auto &kv : g_pool.live[device])
					if (kv.second > (1ull << 30)) fprintf(stderr, "[raygraph] pool:   live block of %.2f GB\n", kv.second / 1e9);
			}
			return e;
		}
	}
	std::lock_guard<std::mutex> lk(g_pool.mu);
	g_pool.live[device][*out] = bytes;
	return hipSuccess;
}

void pool_free(int device, void *p) {
	if (!p) return;
	device &= 63;
	std::lock_guard<std::mutex> lk(g_pool.mu);
	auto it = g_pool.live[device].find(p);
	if (it == g_pool.live[device].end()) return;
	g_pool.free_blocks[device].insert({it->second, p});
	g_pool.live[device].erase(it);
}
}  // namespace

// ---- scratch of the entry point that allocates nothing (rg_pack_reads_noalloc) -----------------------
// One file-scope device array per GPU (every device the code object is loaded on holds its own copy, at its
// own address, until the process ends).  Host code never passes the array's name to a kernel: g_noalloc_ptr[d]
// is the address hipGetSymbolAddress returned with device d current, and that is what kernels, memsets and
// copies get.  Layout: trimmed start of every sequence (8 B), packed bytes of every sequence (8 B), the
// scan's block sums.
static constexpr uint64_t NOALLOC_MAX_READS = RG_NOALLOC_MAX_READS;
static constexpr uint64_t NOALLOC_SUMS = NOALLOC_MAX_READS / SCAN_PER_BLOCK + 2;
static constexpr uint64_t NOALLOC_BYTES = (2 * NOALLOC_MAX_READS + 2 * NOALLOC_SUMS) * sizeof(unsigned long long);
static __device__ __align__(256) unsigned char scratch[NOALLOC_BYTES];
namespace {
std::mutex g_noalloc_mu;
void *g_noalloc_ptr[64] = {nullptr};
hipEvent_t g_noalloc_done[64] = {nullptr};        // recorded after the last kernel a call queued (created once, at set-up)
hipStream_t g_noalloc_last_stream[64] = {nullptr}; // the stream that call was queued on
std::atomic<int> g_noalloc_busy[64];              // 1 while a host thread is inside rg_pack_reads_noalloc on that GPU
struct NoallocGuard {
	int dev;
	bool ok;
	explicit NoallocGuard(int d) : dev(d) {
		int expected = 0;
		ok = g_noalloc_busy[d].compare_exchange_strong(expected, 1);
	}
	~NoallocGuard() {
		if (ok) g_noalloc_busy[dev].store(0);
	}
};
// once per device, with that device current
hipError_t noalloc_init(int device) {
	if (device < 0 || device >= 64) return hipErrorInvalidDevice;
	std::lock_guard<std::mutex> lk(g_noalloc_mu);
	if (g_noalloc_ptr[device]) return hipSuccess;
	void *ptr = nullptr;
	hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
	if (e != hipSuccess) return e;
	if (!ptr) return hipErrorInvalidSymbol;
	e = hipEventCreateWithFlags(&g_noalloc_done[device], hipEventDisableTiming);
	if (e != hipSuccess) return e;
	g_noalloc_ptr[device] = ptr;
	return hipSuccess;
}
}  // namespace

struct Table {
	Slot *slots = nullptr;
	uint64_t cap = 0;
	uint64_t fill_known = 0;    // claimed slots at the last synchronisation
	uint64_t fill_pending = 0;  // upper bound of claims launched since
};

struct ReadBatch {
	const uint64_t *d_packed = nullptr;
	const uint64_t *d_off = nullptr;
	const uint32_t *d_len = nullptr;
	uint64_t n = 0;
	uint64_t first_id = 0;  // global id of the batch's first read
	uint32_t maxlen = 0;
	bool owned = false;
};

enum Phase { PH_CREATED = 0, PH_COUNTED_LOCAL, PH_MERGING, PH_RESOLVED, PH_EDGES, PH_PURGED };

struct rg_handle {
	rg_config cfg;
	int W = 1;
	hipStream_t stream = nullptr;
	bool own_stream = false;
	std::vector<ReadBatch> reads;
	uint64_t n_reads = 0;
	Table tab;    // this rank's partition of the vertex store
	Table local;  // size > 1: aggregation of this rank's own reads before the exchange
	unsigned long long *d_counters = nullptr;
	unsigned long long *h_counters = nullptr;  // pinned
	unsigned long long *d_owner = nullptr;     // [size] counts, [size] cursors
	std::vector<uint64_t> owner_counts;
	bool owner_counts_valid = false;           // set by rg_owner_counts, cleared by whatever changes the entries it counted
	Phase phase = PH_CREATED;
	uint32_t dbg = 0;  // rg_debug_set: internal routes forced by tests and experiments (0 = automatic)
	bool fuse_off = false;  // the sweep is being done again without the first level folded into the record kernel (a run overflowed)
	rg_stats stats{};
	std::string err;
	// optional per-kernel timing with HIP events on the launch stream (rg_set_profiling)
	bool profiling = false;
	std::map<std::string, std::vector<std::pair<hipEvent_t, hipEvent_t>>> prof;
	// rg_histogram is called twice by every caller (size query, then fill): the table is scanned once
	bool log_all_unique = false;         // the last sweep's log holds every key once
	uint64_t log_keys = 0;               // entries of the last sweep's log
	// the Bloom admission's items of that log, made by k_aggregate on the way (AggItems): valid for exactly this log and filter size
	struct HeldItems {
		Item16 *items = nullptr;
		uint8_t *side = nullptr;
		uint32_t *d_count = nullptr;
		std::vector<uint32_t> count;  // what each run holds
		uint32_t n_wg = 0, run_cap = 0;
		uint64_t m_bits = 0, total = 0;
		const Slot *log = nullptr;
		bool valid = false;
	} held;
	Slot *pending_log = nullptr;         // size == 1: the sweep's log of distinct k-mers, admitted before it is inserted
	uint64_t pending_multi = ~0ull;      // its entries seen twice or more (~0: not known): with the false positives they bound what admission stores
	bool pending_borrowed = false;       // the pending log is the caller's buffer (rg_adopt_entries): never freed here
	bool pending_compact = false;        // the pending log holds 16-byte entries (AggItems::compact): read as such by the one-pass admission, expanded for every other route
	bool log_compact = false;            // the last sweep's log is such a log
	uint64_t log_multi = ~0ull;          // its entries seen twice or more, as the sweep's last read of the counters had them
	std::vector<unsigned long long> hist_stage;  // the dense histogram bins on their way to the host, copied ahead of a wait that was due anyway
	bool hist_staged = false;
	uint64_t pending_n = 0, pending_keys = 0;
	bool merge_started = false;          // rg_merge_entries has been called with entries
	void *rec_buf = nullptr;             // rg_records_local .. rg_export_records: super-k-mer records of the local reads
	uint64_t rec_slots = 0;              // slots of rec_buf in use (holes included)
	std::vector<uint64_t> rec_counts;    // valid records per destination rank
	const char *prof_alias = nullptr;     // set: every kernel is profiled under this name (the sample's sweep inside the sweep)
	uint64_t *d_marker_qidx = nullptr;    // rg_markers_export_queries .. rg_markers_apply: where the query of (read, position) went
	unsigned long long *d_marker_kprefix = nullptr;  // rg_markers_query_counts ..: exclusive scan of the slice's k-mers per read
	uint32_t marker_part = 0, marker_nparts = 1;     // the slice of the reads the pending marker queries cover
	uint64_t marker_slice_reads = 0;
	uint64_t marker_total = 0;
	std::vector<uint64_t> marker_counts;
	uint64_t *d_purge_origin = nullptr;  // rg_purge_export_queries .. rg_purge_apply: (slot << 3 | edge bit) per query
	uint64_t purge_total = 0;
	std::vector<uint64_t> purge_counts;
	unsigned long long *d_hist_dense = nullptr;  // live inside rg_resolve_admission only
	uint32_t *d_hist_ovf = nullptr;
	bool hist_valid = false;
	std::vector<std::pair<uint32_t, uint64_t>> hist_cache;  // (coverage, stored pairs)
	void *export_stage[2] = {nullptr, nullptr};  // rg_export_stream: page-locked staging, kept for the next call
	uint64_t export_stage_bytes = 0;
};

// every wait of the host for the handle's stream goes through here (counted: rg_stats.host_syncs)
static inline hipError_t rg_stream_sync(rg_handle *h) {
	h->stats.host_syncs++;
	return hipStreamSynchronize(h->stream);
}

// brackets the launches issued during its lifetime with two HIP events on the handle's stream
struct ProfScope {
	rg_handle *h;
	const char *name;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	ProfScope(rg_handle *h_, const char *name_) : h(h_), name(h_->prof_alias ? h_->prof_alias : name_) {
		if (!h->profiling) return;
		if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
			e0 = nullptr;
			return;
		}
		(void)hipEventRecord(e0, h->stream);
	}
	~ProfScope() {
		if (!e0) return;
		(void)hipEventRecord(e1, h->stream);
		h->prof[name].emplace_back(e0, e1);
	}
};

static thread_local std::string g_create_error;

static int fail(rg_handle *h, int code, const char *fmt, ...) {
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	if (h) h->err = buf;
	else g_create_error = buf;
	return code;
}

#define HIP_TRY(h, call)                                                                                   \
	do {                                                                                                   \
		hipError_t e_ = (call);                                                                            \
		if (e_ != hipSuccess) return fail(h, e_ == hipErrorOutOfMemory ? RG_ERR_NOMEM : RG_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
	} while (0)

static void drop_held_items(rg_handle *h) {
	pool_free(h->cfg.device, h->held.items);
	pool_free(h->cfg.device, h->held.side);
	pool_free(h->cfg.device, h->held.d_count);
	h->held = rg_handle::HeldItems();
}

static int grid_for(uint64_t items) {
	uint64_t b = (items + BLOCK - 1) / BLOCK;
	if (b < 1) b = 1;
	if (b > 256 * 8) b = 256 * 8;  // 256 CUs x 8 workgroups, grid-stride the rest
	return (int)b;
}

static uint64_t next_pow2(uint64_t x) {
	uint64_t p = 1;
	while (p < x) p <<= 1;
	return p;
}

static int sync_counters(rg_handle *h) {
	HIP_TRY(h, hipMemcpyAsync(h->h_counters, h->d_counters, sizeof(unsigned long long) * C_NUM, hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(h, rg_stream_sync(h));
	if (h->h_counters[C_ERR]) return fail(h, RG_ERR_CAPACITY, "table probe limit reached (table full?)");
	return RG_OK;
}

static int table_alloc(rg_handle *h, Table &t, uint64_t cap) {
	HIP_TRY(h, pool_alloc(h->cfg.device, (void **)&t.slots, cap * sizeof(Slot)));
	t.cap = cap;
	hipLaunchKernelGGL(k_init_table, dim3(grid_for(cap)), dim3(BLOCK), 0, h->stream, t.slots, cap);
	HIP_TRY(h, hipGetLastError());
	return RG_OK;
}

static void table_free(rg_handle *h, Table &t) {
	if (t.slots) pool_free(h->cfg.device, t.slots);
	t = Table();
}

// make room for `need` more keys at a load factor of at most 0.7
static int table_reserve(rg_handle *h, Table &t, uint64_t need, int fill_counter) {
	const double LF = 0.7;
	if (!t.slots) {
		uint64_t cap = next_pow2((uint64_t)((double)need / LF) + 1);
		if (cap < 1024) cap = 1024;
		if (h->cfg.table_slots && &t == &h->tab && cap < next_pow2(h->cfg.table_slots)) cap = next_pow2(h->cfg.table_slots);
		int rc = table_alloc(h, t, cap);
		if (rc) return rc;
		t.fill_known = t.fill_pending = 0;
		return RG_OK;
	}
	if ((double)(t.fill_known + t.fill_pending + need) <= LF * (double)t.cap) return RG_OK;
	int rc = sync_counters(h);
	if (rc) return rc;
	t.fill_known = h->h_counters[fill_counter];
	t.fill_pending = 0;
	if ((double)(t.fill_known + need) <= LF * (double)t.cap) return RG_OK;
	uint64_t cap = next_pow2((uint64_t)((double)(t.fill_known + need) / LF) + 1);
	Table bigger;
	rc = table_alloc(h, bigger, cap);
	if (rc) return rc;
	if (h->W == 1) hipLaunchKernelGGL(k_rehash<1>, dim3(grid_for(t.cap)), dim3(BLOCK), 0, h->stream, t.slots, t.cap, bigger.slots, cap, h->d_counters);
	else hipLaunchKernelGGL(k_rehash<2>, dim3(grid_for(t.cap)), dim3(BLOCK), 0, h->stream, t.slots, t.cap, bigger.slots, cap, h->d_counters);
	HIP_TRY(h, hipGetLastError());
	HIP_TRY(h, rg_stream_sync(h));
	pool_free(h->cfg.device, t.slots);
	t.slots = bigger.slots;
	t.cap = cap;
	return RG_OK;
}

// ---- the partitioned sweep (rg_partition.h) ------------------------------------------------------
// Produces the log of aggregated entries for all reads of the handle.  *ok = false means a capacity
// estimate was exceeded (nothing is lost: the caller then runs the direct sweep instead).
// ext != nullptr: the records come from outside (record exchange between ranks: ext_n valid records
// received from all ranks) instead of from this handle's reads.

// 20-byte records (Rec20) whenever they can hold every record of this k; rg_debug_set bit 27 keeps the 24-byte form (tests).
// Every rank of a job decides alike: the form is a function of k.
static bool rec_compact(const rg_handle *h) { return rec_compact_for(h->W, h->cfg.k) && !(h->dbg & 0x8000000); }

// reads -> records: k_superkmers_lanes (a lane per read) for short reads and windows that fit its LDS block,
// k_superkmers (a wavefront per read) otherwise; rg_debug_set bit 16 forces the latter (tests)
static constexpr int SKL_HOST_MAX_W = 24;        // 4 wavefronts x w x 64 lanes x 8 B <= 48 KiB of dynamic LDS
static constexpr uint32_t SKL_MAX_READ = 1024;   // longer reads: lanes of one wavefront would idle behind the longest
template <int W, bool C>
struct SuperkmerLaunch {
	using RM = RecMem<W, C>;
	int grid_wave = 1, grid_lane = 0;
	size_t lds_lane = 0;
	bool wide = false;  // k_superkmers_lanes<W, true, C>: 6-byte window slots, windows of up to 64 m-mers
	bool unrolled = false;  // k_superkmers_lanes_s<17, C>: one-word keys, window of 17 m-mers (k = 31), blocks unrolled
	int grid_max() const { return grid_wave > grid_lane ? grid_wave : grid_lane; }
	void plan(rg_handle *h, int n_cu, int k, int m, uint64_t total_reads) {
		const int w = k - m + 1;
		int occ = 4;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (k_superkmers<W, C>), 256, 0) != hipSuccess || occ < 1) occ = 4;
		grid_wave = (int)std::min<uint64_t>((uint64_t)n_cu * occ, std::max<uint64_t>(1, (total_reads + 255) / 256));
		grid_lane = 0;
		if (w <= SKL_HOST_MAX_W && w <= SKL_MAX_W && !(h->dbg & 0x10000)) {
			lds_lane = (size_t)4 * w * 64 * sizeof(unsigned long long);
			int occl = 0;
			if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occl, (k_superkmers_lanes<W, false, C>), 256, lds_lane) == hipSuccess && occl >= 1)
				grid_lane = (int)std::min<uint64_t>((uint64_t)n_cu * occl, std::max<uint64_t>(1, (total_reads + 255) / 256));
			if constexpr (W == 1) {  // rg_debug_set bit 30 keeps the generic kernel (tests)
				int occs = 0;
				if (w == 17 && grid_lane > 0 && !(h->dbg & 0x40000000u) &&
				    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occs, (k_superkmers_lanes_s<17, C>), 256, lds_lane) == hipSuccess && occs >= 1) {
					grid_lane = (int)std::min<uint64_t>((uint64_t)n_cu * occs, std::max<uint64_t>(1, (total_reads + 255) / 256));
					unrolled = true;
				}
			}
		} else if (w <= SKL_MAX_W_WIDE && !(h->dbg & 0x10000)) {
			lds_lane = (size_t)4 * w * 64 * 6;
			int occl = 0;
			if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_superkmers_lanes<W, true, C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_lane) == hipSuccess &&
			    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occl, (k_superkmers_lanes<W, true, C>), 256, lds_lane) == hipSuccess && occl >= 1) {
				grid_lane = (int)std::min<uint64_t>((uint64_t)n_cu * occl, std::max<uint64_t>(1, (total_reads + 255) / 256));
				wide = true;
			}
		}
		(void)hipGetLastError();
	}
	bool lanes(const ReadBatch &b) const { return grid_lane > 0 && b.maxlen <= SKL_MAX_READ; }
	const char *name(const ReadBatch &b) const { return lanes(b) ? "k_superkmers_lanes" : "k_superkmers"; }
	// the unrolled kernel with the first partition level folded in: records into 2^fbits runs per workgroup, see k_superkmers_lanes_s
	bool can_fuse(const ReadBatch &b) const { return W == 1 && unrolled && lanes(b); }
	void launch_fused(rg_handle *h, const ReadBatch &b, int k, int m, RM *buf, RM *sample, uint64_t sample_cap, uint8_t *side, uint32_t *run_count,
	                  uint32_t run_cap, int fbits) const {
		if constexpr (W == 1)
			hipLaunchKernelGGL((k_superkmers_lanes_s<17, C, true>), dim3(grid_lane), dim3(256), lds_lane, h->stream, b.d_packed, b.d_off, b.d_len, b.n, b.first_id, k, m,
			                   buf, 0ull, h->d_counters, (unsigned long long *)nullptr, 0u, 0u, sample, sample_cap, side, run_count, run_cap, fbits);
	}
	void launch(rg_handle *h, const ReadBatch &b, uint64_t lo, uint64_t n, int k, int m, RM *buf, uint64_t rec_cap,
	            unsigned long long *d_hist, uint32_t modulo, uint32_t hist_shift, RM *sample = nullptr, uint64_t sample_cap = 0) const {
		if constexpr (W == 1) {
			if (lanes(b) && unrolled) {
				hipLaunchKernelGGL((k_superkmers_lanes_s<17, C>), dim3(grid_lane), dim3(256), lds_lane, h->stream, b.d_packed, b.d_off + lo, b.d_len + lo, n,
				                   b.first_id + lo, k, m, buf, rec_cap, h->d_counters, d_hist, modulo, hist_shift, sample, sample_cap);
				return;
			}
		}
		if (lanes(b) && wide)
			hipLaunchKernelGGL((k_superkmers_lanes<W, true, C>), dim3(grid_lane), dim3(256), lds_lane, h->stream, b.d_packed, b.d_off + lo, b.d_len + lo, n,
			                   b.first_id + lo, k, m, buf, rec_cap, h->d_counters, d_hist, modulo, hist_shift, sample, sample_cap);
		else if (lanes(b))
			hipLaunchKernelGGL((k_superkmers_lanes<W, false, C>), dim3(grid_lane), dim3(256), lds_lane, h->stream, b.d_packed, b.d_off + lo, b.d_len + lo, n,
			                   b.first_id + lo, k, m, buf, rec_cap, h->d_counters, d_hist, modulo, hist_shift, sample, sample_cap);
		else
			hipLaunchKernelGGL((k_superkmers<W, C>), dim3(grid_wave), dim3(256), 0, h->stream, b.d_packed, b.d_off + lo, b.d_len + lo, n,
			                   b.first_id + lo, k, m, buf, rec_cap, h->d_counters, d_hist, modulo, hist_shift, sample, sample_cap);
	}
};

// the aggregation kernel of the sweep: a wavefront per bucket of a three-level partition (k_aggregate), or -- rg_debug_set flags
// bit 21 -- a workgroup per bucket of a two-level one (k_aggregate_wg)
template <int W, bool C>
struct AggLaunch {
	using RM = RecMem<W, C>;
	static constexpr int THREADS = 1024, KS = 4096, RS = (W == 1 ? 1024 : 512);  // 132 / 152 KB of LDS: one workgroup of 16 wavefronts per CU
	static constexpr uint32_t KLIMIT = KS * 5 / 8;
	bool wg = true;
	bool big = false;  // k_aggregate with tables of AGG_SLOTS_BIG k-mers per wavefront (diverse inputs)
	int occ = 1;
	void plan(bool use_wg, bool use_big = false) {
		wg = use_wg;
		big = use_big && !use_wg;
		occ = wg ? 1 : 4;
		if (wg) {
			if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (k_aggregate_wg<W, THREADS, KS, RS, false, C>), THREADS, 0) != hipSuccess || occ < 1) occ = 1;
		} else if (big) {
			if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (k_aggregate<W, false, AGG_SLOTS_BIG, C>), 64 * agg_waves<W>(), 0) != hipSuccess || occ < 1) occ = 2;
		} else if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (k_aggregate<W, false, AGG_SLOTS, C>), 64 * agg_waves<W>(), 0) != hipSuccess || occ < 1) occ = 4;
		(void)hipGetLastError();
	}
	uint32_t per_group() const { return wg ? 1u : agg_waves<W>(); }           // buckets a workgroup works on at a time
	uint64_t key_limit() const { return wg ? KLIMIT : (big ? agg_limit<AGG_SLOTS_BIG>() : AGG_LIMIT); }       // keys a bucket attempt can hold
	uint64_t slack(unsigned grid) const { return wg ? 0 : (uint64_t)grid * agg_waves<W>() * LOG_CHUNK; }  // log entries lost to open chunks
	// occurrences a bucket should hold at most: its distinct k-mers must fit the LDS table at the coverage assemblies are run at
	// (4096: C2 at 2^21 buckets of some 48 distinct k-mers, three levels of 7 bits -- a level of 8 bits costs k_scatter 1.1 ms more
	// than the aggregation gains from buckets half the size: 57.1 against 57.6 ms/step; diverse inputs get their finer partition
	// from the record sample)
	uint64_t occ_per_bucket() const { return wg ? (uint64_t)KS * 24 : 4096; }
	unsigned grid_for_buckets(uint64_t n, int n_cu) const { return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n + per_group() - 1) / per_group(), (uint64_t)n_cu * occ)); }
	AggItems items = {nullptr, nullptr, nullptr, 0, 0, 0, 0, 0};  // k_aggregate makes the Bloom admission's items on the way (sweep_partitioned_t decides)
	void fast(rg_handle *h, unsigned grid, const RM *recs, const unsigned long long *begin, uint32_t nb, int k, Slot *log, uint64_t log_cap,
	          uint32_t flag_unique, uint32_t *retry, int m, uint32_t rbits) const {
		if (wg)
			hipLaunchKernelGGL((k_aggregate_wg<W, THREADS, KS, RS, false, C>), dim3(grid), dim3(THREADS), 0, h->stream, recs, begin, nb, k, log, log_cap, h->d_counters,
			                   flag_unique, (const uint32_t *)nullptr, retry, m, rbits);
		else if (big)
			hipLaunchKernelGGL((k_aggregate<W, false, AGG_SLOTS_BIG, C>), dim3(grid), dim3(64 * agg_waves<W>()), 0, h->stream, recs, begin, nb, k, log, log_cap, h->d_counters,
			                   flag_unique, (const uint32_t *)nullptr, retry, m, rbits, items, (const unsigned long long *)nullptr);
		else
			hipLaunchKernelGGL((k_aggregate<W, false, AGG_SLOTS, C>), dim3(grid), dim3(64 * agg_waves<W>()), 0, h->stream, recs, begin, nb, k, log, log_cap, h->d_counters, flag_unique,
			                   (const uint32_t *)nullptr, retry, m, rbits, items, (const unsigned long long *)nullptr);
	}
	// list_len: the list's length is on the device (nlist = its upper bound); not for the workgroup-per-bucket kernel
	void slow(rg_handle *h, unsigned grid, const RM *recs, const unsigned long long *begin, uint32_t nlist, int k, Slot *log, uint64_t log_cap,
	          uint32_t flag_unique, const uint32_t *list, int m, uint32_t rbits, const unsigned long long *list_len = nullptr) const {
		if (wg)
			hipLaunchKernelGGL((k_aggregate_wg<W, THREADS, KS, RS, true, C>), dim3(grid), dim3(THREADS), 0, h->stream, recs, begin, nlist, k, log, log_cap, h->d_counters,
			                   flag_unique, list, (uint32_t *)nullptr, m, rbits);
		else if (big)
			hipLaunchKernelGGL((k_aggregate<W, true, AGG_SLOTS_BIG, C>), dim3(grid), dim3(64 * agg_waves<W>()), 0, h->stream, recs, begin, nlist, k, log, log_cap, h->d_counters,
			                   flag_unique, list, (uint32_t *)nullptr, m, rbits, items, list_len);
		else
			hipLaunchKernelGGL((k_aggregate<W, true, AGG_SLOTS, C>), dim3(grid), dim3(64 * agg_waves<W>()), 0, h->stream, recs, begin, nlist, k, log, log_cap, h->d_counters, flag_unique,
			                   list, (uint32_t *)nullptr, m, rbits, items, list_len);
	}
};

// top / nested_bits: the sweep over the sample of the records that sizes the real one (see SAMPLE_BITS): the buckets are
// numbered by the nested_bits bits below bit `top` of the bucket number, nothing is sampled or probed, no statistics kept.
template <int W, bool C>
static int sweep_partitioned_t(rg_handle *h, Slot **out_log, uint64_t *out_n, bool *ok, const void *ext = nullptr, uint64_t ext_n = 0,
                               int top = BUCKET_TOP, int nested_bits = 0) {
	using RM = RecMem<W, C>;                          // the records' form in memory: 24 / 40 bytes, or 20 when k <= 31
	constexpr int SCATTER_TILE = scatter_tile<W, C>();
	const bool nested = nested_bits > 0;
	*ok = false;
	h->log_all_unique = false;
	*out_log = nullptr;
	*out_n = 0;
	const int dev = h->cfg.device;
	const int k = h->cfg.k;
	const int m = k < MMER_MAX ? k : MMER_MAX;
	const int w = k - m + 1;
	if (w > lmax_for(W, k)) return RG_OK;  // a record could not hold a whole minimizer run (never for valid k)
	uint64_t occ_bound = 0, rec_bound = 0, total_reads = 0;
	for (const ReadBatch &b : h->reads) {
		if (b.maxlen < (uint32_t)k) continue;
		const uint64_t per = (uint64_t)b.maxlen - k + 1;
		occ_bound += b.n * per;
		double frac = 3.0 / (w + 1);
		if (frac > 1.0) frac = 1.0;
		rec_bound += b.n * ((uint64_t)(per * frac) + (per + 63) / 64 + 2);
		// w <= 2: the bound above is the number of k-mers itself, and a wavefront that cannot fit a step's records (up to 128)
		// into its chunk of REC_CHUNK leaves the rest of the chunk unused
		if (frac >= 1.0) rec_bound += b.n * per / (REC_CHUNK / 128) + REC_CHUNK;
		total_reads += b.n;
		if (b.first_id + b.n >= (1ull << 33) - 1) return RG_OK;  // read ids must fit the record header
	}
	if (ext) {  // a record holds (w+1)/2 k-mers on average, w at most
		occ_bound = std::max<uint64_t>(ext_n, ext_n * (uint64_t)(w + 1) / 2);
		rec_bound = ext_n;
		total_reads = 0;
	}
	if (occ_bound == 0) {
		*ok = true;
		return RG_OK;
	}
	// number of buckets: at most 2048 k-mer occurrences each, so that at the coverage assemblies are run at (>= 10)
	// a bucket's distinct k-mers fit the LDS table of k_aggregate; buckets of more diverse inputs are split further
	// by k_aggregate<SLOW> itself
	AggLaunch<W, C> agg;
	agg.plan((h->dbg & 0x200000) != 0, (h->dbg & 0x4000000) != 0);  // bit 21: a workgroup per bucket of a coarser partition (slower as it stands: DESIGN.md); bit 26: the 512-slot tables (tests)
	int bits = nested ? nested_bits : (int)((h->dbg >> 8) & 0xFF);
	const bool bits_forced = bits != 0;
	if (bits == 0) {
		bits = agg.wg ? 12 : 8;
		while (bits < MAX_BUCKET_BITS && (occ_bound >> bits) > agg.occ_per_bucket()) bits++;
	}
	if (bits < 8) bits = 8;
	if (bits > MAX_BUCKET_BITS) bits = MAX_BUCKET_BITS;
	// The rule above assumes the redundancy of an assembly's reads (a k-mer seen ten times or more).  Reads with errors, or a
	// metagenome's thin species, hold several times the distinct k-mers per occurrence: big inputs are sized from a sample of
	// their records instead (below), which may raise the number of bits.
	const bool sampling = !nested && !bits_forced && !agg.wg && (bits >= 16 || (h->dbg & 0x2000000)) && bits < MAX_BUCKET_BITS;  // bit 25: small inputs too (tests)
	// one to three levels of at most 8 bits each, split evenly: fewer digits per level = longer runs per (tile, digit)
	// (22 bits: 7 + 7 + 8; 8 + 7 + 7 measured 1.5-2.4 ms slower on C2, same box)
	// fbits > 0: the record kernel itself does a first level of fbits bits (k_superkmers_lanes_s<17, C, true>: decided below, before
	// the kernel runs -- the record sample may still raise `bits` afterwards, which then only re-splits the levels behind it)
	int nlev = 0, dbits[4] = {0, 0, 0, 0}, fbits = 0;
	auto split_levels = [&]() {
		const int rest = bits - fbits, first = fbits ? 1 : 0;
		const int more = rest <= 8 ? 1 : (rest <= 16 ? 2 : 3);
		nlev = first + more;
		dbits[0] = dbits[1] = dbits[2] = dbits[3] = 0;
		if (fbits) dbits[0] = fbits;
		for (int l = 0, left = rest; l < more; ++l) {
			dbits[first + l] = left / (more - l);  // the odd bits go to the last levels: the earlier ones also write the digit bytes, in runs as short as their own
			left -= dbits[first + l];
		}
	};
	split_levels();
	uint64_t NBUCK = 1ull << bits;
	const uint64_t NBUCK_MAX = sampling ? (1ull << MAX_BUCKET_BITS) : NBUCK;  // what the arrays are sized for
	// persistent grids sized to what is resident at once, so that no second, nearly empty round of
	// workgroups runs behind the first
	int n_cu = 256;
	(void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
	SuperkmerLaunch<W, C> sk;
	sk.plan(h, n_cu, k, m, total_reads);
	const int grid1 = sk.grid_max();
	uint64_t launches1 = 0;  // every launch of k_superkmers leaves up to one open chunk per wavefront
	for (const ReadBatch &b : h->reads)
		if (b.maxlen >= (uint32_t)k) launches1++;
	uint64_t rec_cap = rec_bound + launches1 * (uint64_t)grid1 * 4 * REC_CHUNK;
	// The first level folded into the record kernel (one-word keys in 20-byte records, the kernel unrolled for k = 31, every
	// batch of reads short enough for it): 2^fbits runs per workgroup of that kernel, each with room for its share of the
	// bound on the records, a quarter more and eight standard deviations (the bound is about twice what C2's reads make, so a
	// digit twice as heavy as the others still fits; a run that overflows all the same sends the sweep round again the
	// classic way).  rg_debug_set bit 7: never; bit 3: on small inputs too (tests).
	const uint64_t SEGMAX = 1ull << 19;  // segments a level may read: 2^(bits - its own digit), up to 2^19 with a folded first level in front of three more
	uint32_t run_cap = 0, n_wg = 0;
	if (W == 1 && C && !ext && !nested && !h->fuse_off && !(h->dbg & 0x80u) && !agg.wg && (bits >= 14 || (h->dbg & 0x8u))) {
		bool all = sk.grid_lane > 0;
		for (const ReadBatch &b : h->reads)
			if (b.maxlen >= (uint32_t)k && !sk.can_fuse(b)) all = false;
		if (all) {
			n_wg = (uint32_t)sk.grid_lane;
			// 5 bits: 32 open runs per workgroup, 128 per CU -- what a CU's cache still merges into whole lines.  C2, record kernel /
			// the levels behind it / step: not folded 10.3 / 16.2 (7+7+7) / 53.0 ms; 3 bits 10.2 / 15.5 (6+6+6) / 54.5; 4 bits 10.6 /
			// 15.7 / 54.9; 5 bits 12.7 / 12.8 (8+8) / 51.7; 6 bits 17.0 / 11.8 (7+8) / 55.5
			fbits = std::min(5, std::max(1, bits / 3));
			while (fbits > 0 && ((uint64_t)n_wg << fbits) > SEGMAX) fbits--;
			// small inputs: runs of a few hundred records leave the next level's tiles mostly empty (C1 whole, 608 records per run: 3.9 ->
			// 6.8 ms) -- a run should be expected to fill two tiles or more
			if (fbits > 0 && !(h->dbg & 0x8u) && (double)rec_bound / (double)((uint64_t)n_wg << fbits) < 4.0 * SCATTER_TILE) fbits = 0;
			if (fbits > 0) {
				const double share = (double)rec_bound / (double)((uint64_t)n_wg << fbits);
				const uint64_t rc64 = ((uint64_t)(share * 1.25 + 8.0 * std::sqrt(share)) + 64 + 63) & ~63ull;
				if (rc64 < (1ull << 31) && ((rc64 * n_wg) << fbits) < (1ull << 40)) {
					run_cap = (uint32_t)rc64;
					rec_cap = ((uint64_t)run_cap * n_wg) << fbits;
					split_levels();
				} else {
					fbits = 0;
				}
			}
		}
	}
	const bool fuse = fbits > 0;
	uint32_t *d_runcount = nullptr;
	unsigned long long *d_segend = nullptr;
	const uint64_t sample_cap = sampling ? (rec_bound >> (SAMPLE_BITS - 2)) + 65536 : 0;  // four times its expected share
	RM *d_sample = nullptr;

	RM *buf[2] = {nullptr, nullptr};
	unsigned long long *d_seg = nullptr, *d_hist = nullptr, *d_begin = nullptr, *d_sums = nullptr;
	uint32_t *d_tile = nullptr;
	uint8_t *d_side = nullptr;  // next-level digit of every record of the current level's output
	bool side_valid = false;
	Slot *log = nullptr;
	auto cleanup = [&]() {
		pool_free(dev, d_side);
		pool_free(dev, d_runcount);
		pool_free(dev, d_segend);
		pool_free(dev, d_sample);
		pool_free(dev, buf[0]);
		pool_free(dev, buf[1]);
		pool_free(dev, d_seg);
		pool_free(dev, d_hist);
		pool_free(dev, d_begin);
		pool_free(dev, d_sums);
		pool_free(dev, d_tile);
	};
#define PTRY(call)                                                                                          \
	do {                                                                                                    \
		hipError_t e_ = (call);                                                                             \
		if (e_ != hipSuccess) {                                                                             \
			cleanup();                                                                                      \
			pool_free(dev, log);                                                                            \
			return fail(h, e_ == hipErrorOutOfMemory ? RG_ERR_NOMEM : RG_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
		}                                                                                                   \
	} while (0)
	if (!ext && pool_alloc(dev, (void **)&buf[0], rec_cap * sizeof(RM)) != hipSuccess) {  // not enough memory: direct sweep
		(void)hipGetLastError();
		{ if (getenv("RG_DEBUG")) fprintf(stderr, "[raygraph] partitioned sweep gives up at line %d (no memory)\n", __LINE__); return RG_OK; }
	}
	PTRY(pool_alloc(dev, (void **)&d_seg, (SEGMAX + 1) * sizeof(unsigned long long)));
	PTRY(pool_alloc(dev, (void **)&d_tile, (SEGMAX + 1) * sizeof(uint32_t)));
	PTRY(pool_alloc(dev, (void **)&d_hist, std::max<uint64_t>(NBUCK_MAX, 256) * sizeof(unsigned long long)));
	PTRY(pool_alloc(dev, (void **)&d_begin, (NBUCK_MAX + 1) * sizeof(unsigned long long)));
	PTRY(pool_alloc(dev, (void **)&d_sums, (NBUCK_MAX / SCAN_PER_BLOCK + 2) * sizeof(unsigned long long)));
	if (sampling && pool_alloc(dev, (void **)&d_sample, sample_cap * sizeof(RM)) != hipSuccess) {
		(void)hipGetLastError();
		d_sample = nullptr;  // no sample, no harm: the rule of thumb above stands
	}
	std::vector<uint32_t> run_count;  // fuse: what the record kernel put into each of its runs
	if (fuse) {
		const uint64_t nruns = (uint64_t)n_wg << fbits;
		PTRY(pool_alloc(dev, (void **)&d_side, rec_cap));
		PTRY(pool_alloc(dev, (void **)&d_runcount, nruns * sizeof(uint32_t)));
		PTRY(pool_alloc(dev, (void **)&d_segend, (SEGMAX + 1) * sizeof(unsigned long long)));
		PTRY(hipMemsetAsync(d_runcount, 0, nruns * sizeof(uint32_t), h->stream));
		run_count.resize(nruns);
	}
	PTRY(hipMemsetAsync(d_hist, 0, 256 * sizeof(unsigned long long), h->stream));
	PTRY(hipMemsetAsync(h->d_counters + PC_REC_CURSOR, 0, 11 * sizeof(unsigned long long), h->stream));
	// 1. reads -> records.  The kernel that sees every record first (the one that cuts them, or for records received from
	// other ranks a histogram pass) counts them by the top 8 bits below `top` -- the first level's digits are the top
	// dbits[0] <= 8 of those, its histogram a fold of this one -- and copies the sample aside.
	uint64_t n_in = ext_n, valid = ext_n;
	int rc = RG_OK;
	const int hist0_shift = top - 8;
	std::vector<unsigned long long> hist0(256);
	if (!ext) {
		for (const ReadBatch &b : h->reads) {
			if (b.maxlen < (uint32_t)k) continue;
			{
				ProfScope ps(h, sk.name(b));
				if (fuse) sk.launch_fused(h, b, k, m, buf[0], d_sample, sample_cap, d_side, d_runcount, run_cap, fbits);
				else sk.launch(h, b, 0, b.n, k, m, buf[0], rec_cap, d_hist, 0u, (uint32_t)hist0_shift, d_sample, sample_cap);
			}
			PTRY(hipGetLastError());
		}
	} else {
		const unsigned long long seg1[2] = {0, n_in};
		const uint64_t ntiles = (n_in + SCATTER_TILE - 1) / SCATTER_TILE;
		const uint32_t tile1[2] = {0, (uint32_t)ntiles};
		PTRY(hipMemcpyAsync(d_seg, seg1, sizeof seg1, hipMemcpyHostToDevice, h->stream));
		PTRY(hipMemcpyAsync(d_tile, tile1, sizeof tile1, hipMemcpyHostToDevice, h->stream));
		PTRY(rg_stream_sync(h));  // seg1 / tile1 live on this stack frame
		if (ntiles) {
			ProfScope ps(h, "k_level_hist");
			hipLaunchKernelGGL((k_level_hist<W, C>), dim3((unsigned)ntiles), dim3(256), 0, h->stream, (const RM *)ext, m, 1u, d_seg, d_tile, hist0_shift, 8, d_hist,
			                   d_sample, sample_cap, h->d_counters);
		}
		PTRY(hipGetLastError());
	}
	// the sample's distinct k-mers, counted behind the kernel that took it and read back with the same round trip
	unsigned long long *d_sset = nullptr;
	if (d_sample && !(h->dbg & 0x80000000u)) {  // rg_debug_set bit 31: count them with a nested sweep over the sample instead (tests)
		const uint64_t sset_cap = next_pow2(4 * (occ_bound >> SAMPLE_BITS) + (1ull << 16));
		if (pool_alloc(dev, (void **)&d_sset, sset_cap * sizeof(unsigned long long)) == hipSuccess) {
			PTRY(hipMemsetAsync(d_sset, 0xFF, sset_cap * sizeof(unsigned long long), h->stream));
			ProfScope ps(h, "k_sample_distinct");
			hipLaunchKernelGGL((k_sample_distinct<W, C>), dim3(grid_for(sample_cap)), dim3(256), 0, h->stream, d_sample, sample_cap, k, d_sset, sset_cap, h->d_counters);
			PTRY(hipGetLastError());
		} else {
			(void)hipGetLastError();
			d_sset = nullptr;
		}
	}
	PTRY(hipMemcpyAsync(hist0.data(), d_hist, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
	if (fuse) PTRY(hipMemcpyAsync(run_count.data(), d_runcount, run_count.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
	rc = sync_counters(h);
	const bool sset_used = d_sset != nullptr;
	pool_free(dev, d_sset);  // (the stream has drained)
	if (rc) {
		cleanup();
		return rc;
	}
	if (!ext) {
		if (h->h_counters[PC_REC_OVF]) {
			cleanup();
			if (fuse) {  // a run sized in advance was too small: once more with the first level as a pass of its own
				if (getenv("RG_DEBUG")) fprintf(stderr, "[raygraph] partitioned sweep: a run of the folded first level overflowed, classic levels instead\n");
				h->fuse_off = true;
				if (hipMemsetAsync(h->d_counters + C_OCC, 0, sizeof(unsigned long long), h->stream) != hipSuccess) return fail(h, RG_ERR_HIP, "hipMemsetAsync failed");  // forget what the attempt counted
				const int rc2 = sweep_partitioned_t<W, C>(h, out_log, out_n, ok, ext, ext_n, top, nested_bits);
				h->fuse_off = false;
				return rc2;
			}
			return RG_OK;
		}
		n_in = std::min<uint64_t>(h->h_counters[PC_REC_CURSOR], rec_cap);
	}
	valid = 0;
	for (auto c : hist0) valid += c;
	if (fuse)
		for (auto c : run_count) valid += c;
	if (ext && valid != ext_n) {
		cleanup();
		return fail(h, RG_ERR_STATE, "received records: %llu of %llu are valid", (unsigned long long)valid, (unsigned long long)ext_n);
	}
	if (valid == 0) {
		cleanup();
		*ok = true;
		return RG_OK;
	}
	// 1b. the sample's own sweep: distinct k-mers in 1 / 2^SAMPLE_BITS of the buckets -> buckets of about AGG_TARGET k-mers
	double sample_distinct = 0;  // its estimate of the input's distinct k-mers (0 = no sample): also sizes the log
	if (d_sample) {
		const uint64_t n_sample = h->h_counters[PC_SAMPLE];
		if (n_sample && n_sample <= sample_cap) {
			bool sok = false;
			unsigned long long sample_keys = 0;
			if (sset_used && rc == RG_OK && h->h_counters[PC_SAMPLE_FAIL] == 0) {
				sample_keys = h->h_counters[PC_SAMPLE_DISTINCT];
				sok = true;
			} else {  // no room for the set, or the set too full: the sample goes through a small sweep of its own
				Slot *slog = nullptr;
				uint64_t sn = 0;
				h->prof_alias = "sample_sweep";
				const int src = sweep_partitioned_t<W, C>(h, &slog, &sn, &sok, d_sample, n_sample, BUCKET_TOP - SAMPLE_BITS, 12);
				h->prof_alias = nullptr;
				pool_free(dev, slog);
				if (src != RG_OK) {
					cleanup();
					return src;
				}
				sample_keys = h->h_counters[PC_LOG_KEYS];
			}
			if (sok) {
				const double distinct = (double)sample_keys * (double)(1u << SAMPLE_BITS);
				sample_distinct = distinct;
				int want = bits;
				while (want < MAX_BUCKET_BITS && distinct / (double)(1ull << want) > (double)AGG_TARGET) want++;
				if (getenv("RG_DEBUG"))
					fprintf(stderr, "[raygraph] record sample: %llu records, %llu distinct k-mers -> %.3g in all, 2^%d buckets (rule of thumb 2^%d)\n",
					        (unsigned long long)n_sample, sample_keys, distinct, want, bits);
				if (want != bits) {
					bits = want;
					split_levels();
					NBUCK = 1ull << bits;
				}
				// even the finest partition leaves buckets of about a hundred k-mers or more: the larger tables (C5's share per GPU
				// at 2^24 buckets holds 168 per bucket and a third of the buckets overflowed the 190 of the small ones)
				if (distinct / (double)NBUCK > (double)AGG_BIG_ABOVE && !(h->dbg & 0x200000)) agg.plan(false, true);
			}
		}
		pool_free(dev, d_sample);
		d_sample = nullptr;
	}
	if (!fuse) {  // the first level's histogram
		std::vector<unsigned long long> folded(256, 0);
		for (int d8 = 0; d8 < 256; ++d8) folded[d8 >> (8 - dbits[0])] += hist0[d8];
		hist0 = folded;
		PTRY(hipMemcpyAsync(d_hist, hist0.data(), 256 * sizeof(unsigned long long), hipMemcpyHostToDevice, h->stream));
	}
	if (ext && nlev > 1) {  // received records: both level buffers are ours; one block (see build_table_ranges)
		PTRY(pool_alloc(dev, (void **)&buf[1], 2 * valid * sizeof(RM)));
		buf[0] = buf[1] + valid;  // (pool_free ignores it: not a block of its own)
	} else {
		PTRY(pool_alloc(dev, (void **)&buf[1], valid * sizeof(RM)));
	}

	// 2. partition levels
	int cur = 0;                                      // buf[cur] holds the level's input (level 0 of received records: ext)
	const RM *level_in = ext ? (const RM *)ext : buf[0];
	std::vector<unsigned long long> seg_begin = {0, n_in};  // host copy of the current segmentation
	std::vector<uint32_t> tile_first;
	int used_bits = 0;
	std::vector<unsigned long long> seg_end;  // (runs only)
	for (int level = 0; level < 4 && dbits[level] > 0; ++level) {
		if (fuse && level == 0) {  // done by the record kernel: its runs are the next level's input
			used_bits += dbits[0];
			const uint32_t nruns = n_wg << fbits;
			seg_begin.resize(nruns + 1);
			seg_end.resize(nruns + 1);
			for (uint32_t r = 0; r <= nruns; ++r) {
				seg_begin[r] = (uint64_t)r * run_cap;
				seg_end[r] = seg_begin[r] + (r < nruns ? run_count[r] : 0);
			}
			side_valid = true;
			continue;
		}
		const bool runs = fuse && level == 1;  // segments = the record kernel's runs, n_wg of them per output segment
		const uint32_t S = (uint32_t)seg_begin.size() - 1;
		const uint32_t seg_div = runs ? n_wg : 1u;
		const int db = dbits[level];
		const int shift = top - used_bits - db;
		tile_first.assign(S + 1, 0);
		uint64_t ntiles = 0;
		for (uint32_t sgi = 0; sgi < S; ++sgi) {
			tile_first[sgi] = (uint32_t)ntiles;
			ntiles += ((runs ? seg_end[sgi] : seg_begin[sgi + 1]) - seg_begin[sgi] + SCATTER_TILE - 1) / SCATTER_TILE;
		}
		tile_first[S] = (uint32_t)ntiles;
		PTRY(hipMemcpyAsync(d_seg, seg_begin.data(), (S + 1) * sizeof(unsigned long long), hipMemcpyHostToDevice, h->stream));
		PTRY(hipMemcpyAsync(d_tile, tile_first.data(), (S + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
		if (runs) PTRY(hipMemcpyAsync(d_segend, seg_end.data(), (S + 1) * sizeof(unsigned long long), hipMemcpyHostToDevice, h->stream));
		const unsigned long long *segend_arg = runs ? d_segend : nullptr;
		const uint64_t nout = (uint64_t)(S / seg_div) << db;  // buckets after this level
		const bool have_next = level < 3 && dbits[level + 1] > 0;
		if (have_next && !d_side && pool_alloc(dev, (void **)&d_side, valid) != hipSuccess) {
			(void)hipGetLastError();  // no room for the side bytes: the next histogram reads the records
			d_side = nullptr;
		}
		if (level > 0 && side_valid) {  // the previous level's scatter left this level's digits, one byte per record
			PTRY(hipMemsetAsync(d_hist, 0, nout * sizeof(unsigned long long), h->stream));
			ProfScope ps(h, "k_level_hist_side");
			// (runs: the record kernel left 8 bucket bits per record, of which this level takes the top db)
			hipLaunchKernelGGL(k_level_hist_side<SCATTER_TILE>, dim3((unsigned)((ntiles + HIST_SIDE_TILES - 1) / HIST_SIDE_TILES)), dim3(256), 0, h->stream, d_side, S, d_seg, d_tile, db, d_hist, (uint32_t)ntiles,
			                   segend_arg, seg_div, runs ? 8 - db : 0);
			PTRY(hipGetLastError());
		} else if (level > 0) {  // level 0's histogram is in place
			PTRY(hipMemsetAsync(d_hist, 0, nout * sizeof(unsigned long long), h->stream));
			ProfScope ps(h, "k_level_hist");
			hipLaunchKernelGGL((k_level_hist<W, C>), dim3((unsigned)ntiles), dim3(256), 0, h->stream, level_in, m, S, d_seg, d_tile, shift, db, d_hist);
			PTRY(hipGetLastError());
		}
		if (!buf[cur ^ 1]) PTRY(pool_alloc(dev, (void **)&buf[cur ^ 1], valid * sizeof(RM)));
		// exclusive scan -> d_begin[nout+1]; the scatter's cursors are a copy of it (reusing d_hist)
		const uint32_t nblk = (uint32_t)((nout + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK);
		hipLaunchKernelGGL(k_scan_local, dim3(nblk), dim3(256), 0, h->stream, d_hist, d_begin, nout, d_sums);
		hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(64), 0, h->stream, d_sums, nblk, d_sums + nblk);
		hipLaunchKernelGGL(k_scan_add, dim3(nblk), dim3(256), 0, h->stream, d_begin, nout, d_sums, d_sums + nblk, d_hist);
		PTRY(hipGetLastError());
		{
			ProfScope ps(h, "k_scatter");
			uint8_t *side_out = (have_next && !(h->dbg & 0x100000)) ? d_side : nullptr;  // bit 20: histograms read the records (tests)
			const int ndb = have_next ? dbits[level + 1] : 0;
			if (nested)
				hipLaunchKernelGGL((k_scatter<W, 1, C>), dim3((unsigned)ntiles), dim3(scatter_threads<W, C>()), 0, h->stream, level_in, m, S, d_seg, d_tile, shift, db, d_hist,
				                   buf[cur ^ 1], 0u, side_out, shift - ndb, (1u << ndb) - 1u, (const unsigned long long *)nullptr, 1u);
			else
				hipLaunchKernelGGL((k_scatter<W, 0, C>), dim3((unsigned)ntiles), dim3(scatter_threads<W, C>()), 0, h->stream, level_in, m, S, d_seg, d_tile, shift, db, d_hist,
				                   buf[cur ^ 1], 0u, side_out, shift - ndb, (1u << ndb) - 1u, segend_arg, seg_div);
			side_valid = side_out != nullptr;
		}
		PTRY(hipGetLastError());
		used_bits += db;
		cur ^= 1;
		level_in = buf[cur];
		if (level < 3 && dbits[level + 1] > 0) {  // the next level needs this segmentation on the host
			seg_begin.resize(nout + 1);
			PTRY(hipMemcpyAsync(seg_begin.data(), d_begin, (nout + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
			PTRY(rg_stream_sync(h));
			if (seg_begin[nout] != valid) {
				cleanup();
				return fail(h, RG_ERR_STATE, "partition level %d lost records: %llu of %llu", level, seg_begin[nout], (unsigned long long)valid);
			}
		}
	}
	// 3. aggregate per bucket into the log, one entry per distinct k-mer.  A quarter of the occurrences is a
	// deterministic size (the pool reuses the block) that only inputs of little redundancy (coverage below 4, or
	// errors in most reads) exceed; for them the aggregation is run again with a larger log.
	// Sub-passes by whole records only help while a bucket holds several minimizers: beyond 2^24 buckets a bucket of a 100-Mbp
	// genome is one minimizer and its error variants, and halving by records until that shows costs more than it saves
	// (C2 with 1 % errors: k_aggregate_retry 29 ms with hash levels from the start, 55 ms behind 8 record levels)
	const uint32_t agg_rbits = (!nested && bits < 24) ? (uint32_t)(24 - bits) : 0u;
	const unsigned agg_grid = agg.grid_for_buckets(NBUCK, n_cu);
	const uint64_t log_slack = agg.slack(agg_grid);
	uint64_t log_cap = std::min<uint64_t>(occ_bound, std::max<uint64_t>(occ_bound / 4, 1ull << 20)) + log_slack;
	// the record sample has counted the distinct k-mers of 1 / 4096 of the buckets already: the log is sized from that and the
	// bucket probe below (1 / 256 of the buckets aggregated ahead of the rest, 0.17 ms and two round trips on C2) is left out
	const bool sized_by_sample = sample_distinct > 0 && !(h->dbg & 0x40000);
	if (sized_by_sample) log_cap = std::min<uint64_t>(occ_bound, std::max<uint64_t>((uint64_t)(sample_distinct * 1.15) + (1ull << 16), 1ull << 20)) + log_slack;
	if ((h->dbg & 64) && !nested) log_cap = LOG_CHUNK;  // tests: the first attempt overflows
	uint32_t *d_retry = nullptr;
	if (pool_alloc(dev, (void **)&d_retry, NBUCK * sizeof(uint32_t)) != hipSuccess) {
		(void)hipGetLastError();
		cleanup();
		{ if (getenv("RG_DEBUG")) fprintf(stderr, "[raygraph] partitioned sweep gives up at line %d (no memory)\n", __LINE__); return RG_OK; }
	}
	// entries may be marked unique unless other ranks aggregate the same keys (entry route, size > 1)
	const uint32_t flag_unique = (h->cfg.size == 1 || ext) ? 1u : 0u;
	// One rank, a filter of 2^28 bits or more (the reference's cap is 2^33: every large input) and a log sized by the record
	// sample: k_aggregate makes the Bloom admission's items while it writes the log, and rg_resolve_admission starts from
	// them instead of reading the log to make them (k_xor_keys: 13 ms on C2 with 1 % errors, 1.1 error-free).  A run that
	// overflows, a log that had to grow, an entry that is not unique: the items are dropped and the admission goes the
	// usual way.  rg_debug_set bit 2: never.
	if (!nested) drop_held_items(h);
	bool with_items = false;
	// ... and, for one-word keys, writes the log in 16-byte entries (key, meta): the item carries the stamp, and the one-pass
	// admission reads nothing else (half the bytes of the log, written here and read there; every other route gets the log
	// expanded first: expand_pending_log).  Such a log serves only with its items: if they turn out unusable the aggregation is
	// done again the plain way.  rg_debug_set bit 27: never.
	bool compact = false;
	uint64_t items_bits = 0;
	if (!nested && !ext && h->cfg.size == 1 && flag_unique && !agg.wg && !(h->dbg & 0x4u) && sized_by_sample && (sample_distinct >= (double)(1u << 21) || (h->dbg & 0x2000000u))) {  // (bit 25: the sample, hence this, on small inputs too: tests)
		uint64_t mb = h->cfg.bloom_bits;
		if (mb == RG_BLOOM_DEFAULT) {  // as rg_resolve_admission will size it (MessageProcessor.cpp:886-899 + VerticesExtractor.cpp:276-320)
			mb = 4ull * ((uint64_t)h->cfg.k * 2 * 2) * h->n_reads;
			if (mb > 1024ull * 1024 * 1024 * 8) mb = 1024ull * 1024 * 1024 * 8;
		}
		if (mb >= (1ull << 28) && !(mb & (mb - 1))) {  // (2^28: what leaves the admission's key partition two levels or more behind this one)
			const uint64_t nruns = (uint64_t)AGG_ITEM_RUNS * agg_grid;
			const double share = (double)(log_cap - log_slack) / (double)nruns;
			const uint64_t rcap = ((uint64_t)(share * 1.25 + 8.0 * std::sqrt(share)) + 64 + 63) & ~63ull;
			if (rcap < (1ull << 31) && pool_alloc(dev, (void **)&h->held.items, nruns * rcap * sizeof(Item16)) == hipSuccess &&
			    pool_alloc(dev, (void **)&h->held.side, nruns * rcap) == hipSuccess && pool_alloc(dev, (void **)&h->held.d_count, nruns * sizeof(uint32_t)) == hipSuccess) {
				int B = 0;
				while ((1ull << B) < mb) B++;
				agg.items = AggItems{h->held.items, h->held.side, h->held.d_count, (uint32_t)rcap, agg_grid, mb - 1, B - 5, 0};
				h->held.n_wg = agg_grid;
				h->held.run_cap = (uint32_t)rcap;
				h->held.count.resize(nruns);
				items_bits = mb;
				with_items = true;
				compact = W == 1 && !(h->dbg & 0x8000000u);
				agg.items.compact = compact ? 1u : 0u;
			} else {
				(void)hipGetLastError();
				drop_held_items(h);
			}
		}
	}
	for (int attempt = 0;; ++attempt) {
		if (with_items && attempt > 0) {  // the log had to grow: the runs were sized for the first one
			with_items = false;
			compact = false;
			agg.items.items = nullptr;
			agg.items.compact = 0;
			drop_held_items(h);
		}
		const size_t log_esz = compact ? sizeof(Ent16) : sizeof(Slot);
		const bool force_probe = (h->dbg & 0x40000) != 0;
		const bool will_probe = !nested && attempt == 0 && !(h->dbg & 64) && !sized_by_sample && (NBUCK >= (1u << 16) || force_probe);
		const uint32_t P = force_probe ? std::max<uint32_t>((uint32_t)NBUCK / 4, 1u) : (uint32_t)NBUCK / 256;
		// the sample runs on a small log of its own (one entry per occurrence of its share would always do; 1.5 x the average
		// share + the chunks left open is what it gets), the real log is allocated once its size is known
		const uint64_t alloc_cap = will_probe ? std::min<uint64_t>(log_cap, (uint64_t)((double)occ_bound * 1.5 * (double)P / (double)NBUCK) + log_slack + (1ull << 16)) : log_cap;
		if (pool_alloc(dev, (void **)&log, alloc_cap * log_esz) != hipSuccess) {
			(void)hipGetLastError();
			log = nullptr;
			pool_free(dev, d_retry);
			cleanup();
			{ if (getenv("RG_DEBUG")) fprintf(stderr, "[raygraph] partitioned sweep gives up at line %d (no memory)\n", __LINE__); return RG_OK; }
		}
		// Before the first attempt a sample of the buckets (the bucket number is a hash: any range is a fair sample) is
		// aggregated alone, overflowing buckets included, and the log sized from what it produced: an input of little
		// redundancy (errors in most reads) then gets its larger log at once instead of losing a whole aggregation to
		// find out.  Costs 1/256 of the aggregation; bit 18 of rg_debug_set runs it on small inputs too and always resizes (tests).
		bool mostly_overflowing = false;  // the sample says that most buckets hold more k-mers than the LDS table: split them all from the start
		if (sized_by_sample && attempt == 0) mostly_overflowing = sample_distinct / (double)NBUCK > 0.9 * (double)agg.key_limit();
		if (will_probe) {
			bool fine = hipMemsetAsync(h->d_counters + PC_LOG_CURSOR, 0, 6 * sizeof(unsigned long long), h->stream) == hipSuccess;
			if (fine) {
				ProfScope ps(h, "k_aggregate_probe");
				agg.fast(h, agg.grid_for_buckets(P, n_cu), level_in, d_begin, P, k, log, alloc_cap, flag_unique, d_retry, m, agg_rbits);
				fine = hipGetLastError() == hipSuccess && sync_counters(h) == RG_OK;
				const uint64_t n_retry = fine ? h->h_counters[PC_AGG_RETRY] : 0;
				mostly_overflowing = fine && 2 * n_retry > P;
				if (fine && n_retry && !h->h_counters[PC_LOG_OVF] && h->h_counters[PC_LOG_CURSOR] + n_retry * agg.key_limit() <= alloc_cap) {
					agg.slow(h, agg.grid_for_buckets(n_retry, n_cu), level_in, d_begin, (uint32_t)n_retry, k, log, alloc_cap, flag_unique, d_retry, m, agg_rbits);
					fine = hipGetLastError() == hipSuccess && sync_counters(h) == RG_OK;
				} else if (n_retry) {
					fine = false;  // the sample alone does not fit: the attempt below finds out and the log doubles
				}
			}
			if (fine && !h->h_counters[PC_LOG_OVF]) {
				const double per_bucket = (double)h->h_counters[PC_LOG_KEYS] / (double)P;
				const uint64_t proj = std::min<uint64_t>(occ_bound, (uint64_t)(per_bucket * (double)NBUCK * 1.10) + (1ull << 16));
				if (getenv("RG_DEBUG"))
					fprintf(stderr, "[raygraph] log probe: %llu keys in %u of %llu buckets -> %llu entries expected, log holds %llu\n",
					        (unsigned long long)h->h_counters[PC_LOG_KEYS], P, (unsigned long long)NBUCK, (unsigned long long)proj,
					        (unsigned long long)(alloc_cap - log_slack));
				log_cap = std::max<uint64_t>(proj, 1ull << 20) + log_slack;
			} else {
				(void)hipGetLastError();
			}
			pool_free(dev, log);  // the sample's log; the real one has the size the sample asks for (or, if the sample failed, the default)
			log = nullptr;
			if (pool_alloc(dev, (void **)&log, log_cap * log_esz) != hipSuccess) {
				(void)hipGetLastError();
				pool_free(dev, d_retry);
				cleanup();
				{ if (getenv("RG_DEBUG")) fprintf(stderr, "[raygraph] partitioned sweep gives up at line %d (no memory)\n", __LINE__); return RG_OK; }
			}
		}
		bool too_small = false;
		hipError_t e0 = hipMemsetAsync(h->d_counters + PC_LOG_CURSOR, 0, 6 * sizeof(unsigned long long), h->stream);
		if (e0 == hipSuccess) e0 = hipMemsetAsync(h->d_counters + PC_LOG_MULTI, 0, sizeof(unsigned long long), h->stream);
		if (e0 == hipSuccess && with_items) e0 = hipMemsetAsync(h->d_counters + PC_ITEM_OVF, 0, sizeof(unsigned long long), h->stream);
		if (e0 == hipSuccess && with_items) e0 = hipMemsetAsync(h->held.d_count, 0, h->held.count.size() * sizeof(uint32_t), h->stream);
		if (e0 == hipSuccess) e0 = hipMemsetAsync(h->d_counters + PC_CUT_ANCHORED, 0, 2 * sizeof(unsigned long long), h->stream);
		if (e0 == hipSuccess && mostly_overflowing) {
			// an attempt at whole buckets would be thrown away for most of them (C2 with 1 % errors: 47 ms): every bucket goes
			// through the splitting kernel at once
			ProfScope ps(h, "k_aggregate_split");
			agg.slow(h, agg_grid, level_in, d_begin, (uint32_t)NBUCK, k, log, log_cap, flag_unique, (const uint32_t *)nullptr, m, agg_rbits);
			e0 = hipGetLastError();
		} else if (e0 == hipSuccess) {
			ProfScope ps(h, "k_aggregate");
			agg.fast(h, agg_grid, level_in, d_begin, (uint32_t)NBUCK, k, log, log_cap, flag_unique, d_retry, m, agg_rbits);
			e0 = hipGetLastError();
		}
		if (e0 == hipSuccess && !agg.wg && !mostly_overflowing) {
			// the buckets whose k-mers did not fit the LDS table, split further: the launch follows at once and reads the length of
			// the list from the counter the first one left (no wait of the host in between; a log the first launch has filled
			// stops it at its first chunk)
			ProfScope ps(h, "k_aggregate_retry");
			agg.slow(h, agg.grid_for_buckets(NBUCK, n_cu), level_in, d_begin, (uint32_t)NBUCK, k, log, log_cap, flag_unique, d_retry, m, agg_rbits, h->d_counters + PC_AGG_RETRY);
		} else if (e0 == hipSuccess && !mostly_overflowing) {
			rc = sync_counters(h);
			const uint64_t n_retry = rc ? 0 : h->h_counters[PC_AGG_RETRY];
			// every bucket left over holds more k-mers than the LDS table: if even that many cannot fit, skip the launch
			if (!rc && n_retry && h->h_counters[PC_LOG_CURSOR] + n_retry * agg.key_limit() > log_cap) too_small = true;
			if (n_retry && !too_small && !h->h_counters[PC_LOG_OVF]) {  // the buckets whose k-mers did not fit the LDS table: split further
				ProfScope ps(h, "k_aggregate_retry");
				agg.slow(h, agg.grid_for_buckets(n_retry, n_cu), level_in, d_begin, (uint32_t)n_retry, k, log, log_cap, flag_unique, d_retry, m, agg_rbits);
			}
		}
		hipError_t e = (e0 != hipSuccess) ? e0 : hipGetLastError();
		if (e == hipSuccess && !rc && with_items) e = hipMemcpyAsync(h->held.count.data(), h->held.d_count, h->held.count.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream);
		if (e == hipSuccess && !rc) rc = sync_counters(h);
		if (e != hipSuccess || rc) {
			pool_free(dev, d_retry);
			cleanup();
			drop_held_items(h);
			pool_free(dev, log);
			return e != hipSuccess ? fail(h, RG_ERR_HIP, "k_aggregate: %s", hipGetErrorString(e)) : rc;
		}
		if (agg.wg && h->h_counters[PC_LDS_SPILL]) {  // a share of a bucket that no split could make fit: count with the direct sweep instead
			pool_free(dev, d_retry);
			cleanup();
			pool_free(dev, log);
			return RG_OK;
		}
		if (!h->h_counters[PC_LOG_OVF] && !too_small) {
			if (compact) {  // usable only if the items are (the conditions below, where they are kept)
				uint64_t total = 0;
				for (auto c : h->held.count) total += c;
				const uint64_t n = h->h_counters[PC_LOG_CURSOR];
				const bool fine = flag_unique && h->h_counters[PC_NONUNIQUE] == 0 && h->h_counters[PC_LDS_SPILL] == 0 && h->h_counters[PC_ITEM_OVF] == 0 &&
				                  total == h->h_counters[PC_LOG_KEYS] && n > 0 && n < (1ull << 32);
				if (!fine) {
					if (getenv("RG_DEBUG")) fprintf(stderr, "[raygraph] partitioned sweep: the compact log is of no use without its items, aggregating again\n");
					pool_free(dev, log);
					log = nullptr;
					continue;  // (the next attempt drops the items and writes full entries; the log keeps its size)
				}
			}
			break;
		}
		pool_free(dev, log);
		log = nullptr;
		if (log_cap >= occ_bound + log_slack) {  // cannot happen: one entry per occurrence always fits
			pool_free(dev, d_retry);
			cleanup();
			return RG_OK;
		}
		// twice the size, then one slot per occurrence (releasing and mapping tens of GiB is what makes this slow: no more than needed)
		const uint64_t doubled = (h->dbg & 64) ? 0 : 2 * (log_cap - log_slack);
		log_cap = ((attempt == 0 && doubled) ? std::min<uint64_t>(occ_bound, doubled) : occ_bound) + log_slack;
	}
	pool_free(dev, d_retry);
	cleanup();
	*out_log = log;
	*out_n = h->h_counters[PC_LOG_CURSOR];
	*ok = true;
	if (nested) return RG_OK;  // the caller reads PC_LOG_KEYS
	h->log_all_unique = flag_unique && h->h_counters[PC_NONUNIQUE] == 0 && h->h_counters[PC_LDS_SPILL] == 0;
	h->log_keys = h->h_counters[PC_LOG_KEYS];
	h->log_compact = compact;
	h->log_multi = h->h_counters[PC_LOG_MULTI];
	h->stats.compact_log = compact ? 1 : 0;
	if (with_items) {
		uint64_t total = 0;
		for (auto c : h->held.count) total += c;
		if (h->log_all_unique && h->h_counters[PC_ITEM_OVF] == 0 && total == h->log_keys && *out_n < (1ull << 32)) {
			h->held.valid = true;
			h->held.log = log;
			h->held.m_bits = items_bits;
			h->held.total = total;
		} else {
			if (getenv("RG_DEBUG")) fprintf(stderr, "[raygraph] the items k_aggregate made are dropped (%llu of %llu keys, %llu runs overflowed)\n", (unsigned long long)total, (unsigned long long)h->log_keys, (unsigned long long)h->h_counters[PC_ITEM_OVF]);
			drop_held_items(h);
			if (compact) {  // (checked before the loop was left)
				pool_free(dev, log);
				*out_log = nullptr;
				return fail(h, RG_ERR_STATE, "a compact log lost its items");
			}
		}
	}
	h->stats.superkmer_records = valid;
	h->stats.partition_levels = (uint64_t)((dbits[0] > 0) + (dbits[1] > 0) + (dbits[2] > 0) + (dbits[3] > 0));
	h->stats.folded_level_bits = (uint64_t)fbits;
	h->stats.log_entries = *out_n;
	if (getenv("RG_DEBUG"))
		fprintf(stderr, "[raygraph] partitioned sweep: %llu reads, %llu occurrences, 2^%d buckets, %llu record slots (%llu valid, cap %llu), log %llu entries (cap %llu), %llu LDS spills, %llu bucket retries\n",
		        (unsigned long long)total_reads, (unsigned long long)h->h_counters[C_OCC], bits, (unsigned long long)n_in, (unsigned long long)valid,
		        (unsigned long long)rec_cap, (unsigned long long)*out_n, (unsigned long long)log_cap, (unsigned long long)h->h_counters[PC_LDS_SPILL],
		        (unsigned long long)h->h_counters[PC_AGG_RETRY]);
	if (getenv("RG_DEBUG") && (h->h_counters[PC_CUT_ANCHORED] || h->h_counters[PC_CUT_EXPANDED]))
		fprintf(stderr, "[raygraph] records cut by a read end: %llu added to an anchor's k-mers, %llu expanded\n", (unsigned long long)h->h_counters[PC_CUT_ANCHORED],
		        (unsigned long long)h->h_counters[PC_CUT_EXPANDED]);
#undef PTRY
	return RG_OK;
}

static int sweep_partitioned(rg_handle *h, Slot **out_log, uint64_t *out_n, bool *ok, const void *ext = nullptr, uint64_t ext_n = 0) {
	if (h->W == 2) return sweep_partitioned_t<2, false>(h, out_log, out_n, ok, ext, ext_n);
	return rec_compact(h) ? sweep_partitioned_t<1, true>(h, out_log, out_n, ok, ext, ext_n) : sweep_partitioned_t<1, false>(h, out_log, out_n, ok, ext, ext_n);
}

// ---- record exchange between ranks: stage 1 on the sender (reads -> records, counted per destination) ----
template <int W, bool C>
static int records_local_t(rg_handle *h, uint32_t part, uint32_t nparts) {
	using RM = RecMem<W, C>;
	const int dev = h->cfg.device;
	const int k = h->cfg.k;
	const int m = k < MMER_MAX ? k : MMER_MAX;
	const int w = k - m + 1;
	const uint32_t size = (uint32_t)h->cfg.size;
	if (w > lmax_for(W, k)) return fail(h, RG_ERR_STATE, "k-mer length %d cannot be cut into records", k);
	if (size > 256) return fail(h, RG_ERR_STATE, "the record exchange supports at most 256 ranks (use rg_count_local + rg_export_owner_buckets)");
	// part p of P takes reads [n*p/P, n*(p+1)/P) of every batch
	auto slice = [&](const ReadBatch &b, uint64_t &lo, uint64_t &hi) {
		lo = b.n / nparts * part + std::min<uint64_t>(part, b.n % nparts);
		hi = lo + b.n / nparts + (part < b.n % nparts ? 1 : 0);
	};
	uint64_t rec_bound = 0, total_reads = 0, launches1 = 0;
	for (const ReadBatch &b : h->reads) {
		if (b.maxlen < (uint32_t)k) continue;
		uint64_t lo, hi;
		slice(b, lo, hi);
		const uint64_t per = (uint64_t)b.maxlen - k + 1;
		rec_bound += (hi - lo) * per;  // exact upper bound: a record holds at least one k-mer
		total_reads += hi - lo;
		launches1++;
		if (b.first_id + b.n >= (1ull << 33) - 1) return fail(h, RG_ERR_CAPACITY, "read ids above 2^33 do not fit a record");
	}
	h->rec_counts.assign(size, 0);
	h->rec_slots = 0;
	if (rec_bound == 0) return RG_OK;
	// the expected number of records is 2/(w+1) per k-mer; three times that has never been seen to overflow, and if it
	// does the exact bound is used in a second attempt
	int n_cu = 256;
	(void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
	SuperkmerLaunch<W, C> sk;
	sk.plan(h, n_cu, k, m, total_reads);
	const int grid1 = sk.grid_max();
	unsigned long long *d_hist = nullptr;
	HIP_TRY(h, pool_alloc(dev, (void **)&d_hist, 256 * sizeof(unsigned long long)));
	for (int attempt = 0; attempt < 2; ++attempt) {
		double frac = attempt ? 1.0 : std::min(1.0, 3.0 / (w + 1));
		// (frac == 1: one record per k-mer at most, plus what wavefronts leave unused at the end of their chunks)
		const uint64_t rec_cap = (uint64_t)(rec_bound * frac) + (frac >= 1.0 ? rec_bound / (REC_CHUNK / 128) : 0) + total_reads * 3 + launches1 * (uint64_t)grid1 * 4 * REC_CHUNK;
		RM *buf = nullptr;
		if (pool_alloc(dev, (void **)&buf, rec_cap * sizeof(RM)) != hipSuccess) {
			(void)hipGetLastError();
			pool_free(dev, d_hist);
			return fail(h, RG_ERR_NOMEM, "out of device memory for %llu records", (unsigned long long)rec_cap);
		}
		hipError_t e = hipMemsetAsync(d_hist, 0, 256 * sizeof(unsigned long long), h->stream);
		if (e == hipSuccess) e = hipMemsetAsync(h->d_counters + PC_REC_CURSOR, 0, 8 * sizeof(unsigned long long), h->stream);
		if (e == hipSuccess) e = hipMemsetAsync(h->d_counters + C_OCC, 0, sizeof(unsigned long long), h->stream);
		for (const ReadBatch &b : h->reads) {
			if (b.maxlen < (uint32_t)k || e != hipSuccess) continue;
			uint64_t lo, hi;
			slice(b, lo, hi);
			if (hi == lo) continue;
			ProfScope ps(h, sk.name(b));
			sk.launch(h, b, lo, hi - lo, k, m, buf, rec_cap, d_hist, size, 16u);
			e = hipGetLastError();
		}
		std::vector<unsigned long long> hist(256);
		if (e == hipSuccess) e = hipMemcpyAsync(hist.data(), d_hist, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream);
		int rc = (e == hipSuccess) ? sync_counters(h) : fail(h, RG_ERR_HIP, "k_superkmers: %s", hipGetErrorString(e));
		if (rc) {
			pool_free(dev, buf);
			pool_free(dev, d_hist);
			return rc;
		}
		if (h->h_counters[PC_REC_OVF]) {
			pool_free(dev, buf);
			continue;
		}
		h->rec_buf = buf;
		h->rec_slots = std::min<uint64_t>(h->h_counters[PC_REC_CURSOR], rec_cap);
		for (uint32_t r = 0; r < size; ++r) h->rec_counts[r] = hist[r];
		pool_free(dev, d_hist);
		return RG_OK;
	}
	pool_free(dev, d_hist);
	return fail(h, RG_ERR_CAPACITY, "record buffer overflow");
}

// stage 2 on the sender: one scatter pass by destination rank straight into the caller's send buffer
template <int W, bool C>
static int export_records_t(rg_handle *h, void *d_out) {
	using RM = RecMem<W, C>;
	constexpr int SCATTER_TILE = scatter_tile<W, C>();
	const int dev = h->cfg.device;
	const int k = h->cfg.k;
	const int m = k < MMER_MAX ? k : MMER_MAX;
	const uint32_t size = (uint32_t)h->cfg.size;
	const uint64_t n_in = h->rec_slots;
	if (n_in == 0) return RG_OK;
	unsigned long long *d_seg = nullptr, *d_cur = nullptr;
	uint32_t *d_tile = nullptr;
	hipError_t e = pool_alloc(dev, (void **)&d_seg, 2 * sizeof(unsigned long long));
	if (e == hipSuccess) e = pool_alloc(dev, (void **)&d_tile, 2 * sizeof(uint32_t));
	if (e == hipSuccess) e = pool_alloc(dev, (void **)&d_cur, 256 * sizeof(unsigned long long));
	const unsigned long long seg[2] = {0, n_in};
	const uint64_t ntiles = (n_in + SCATTER_TILE - 1) / SCATTER_TILE;
	const uint32_t tile_first[2] = {0, (uint32_t)ntiles};
	std::vector<unsigned long long> cursors(256, 0);
	unsigned long long run = 0;
	for (uint32_t r = 0; r < size; ++r) {
		cursors[r] = run;
		run += h->rec_counts[r];
	}
	if (e == hipSuccess) e = hipMemcpyAsync(d_seg, seg, sizeof seg, hipMemcpyHostToDevice, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(d_tile, tile_first, sizeof tile_first, hipMemcpyHostToDevice, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(d_cur, cursors.data(), 256 * sizeof(unsigned long long), hipMemcpyHostToDevice, h->stream);
	if (e == hipSuccess) {
		ProfScope ps(h, "k_scatter");
		hipLaunchKernelGGL((k_scatter<W, 0, C>), dim3((unsigned)ntiles), dim3(scatter_threads<W, C>()), 0, h->stream, (const RM *)h->rec_buf, m, 1u, d_seg, d_tile, 0, 8, d_cur,
		                   (RM *)d_out, size, (uint8_t *)nullptr, 0, 0u);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = rg_stream_sync(h);
	pool_free(dev, d_seg);
	pool_free(dev, d_tile);
	pool_free(dev, d_cur);
	pool_free(dev, h->rec_buf);
	h->rec_buf = nullptr;
	h->rec_slots = 0;
	if (e != hipSuccess) return fail(h, RG_ERR_HIP, "record export: %s", hipGetErrorString(e));
	return RG_OK;
}

// ---- Bloom admission through region partitioning (rg_bloom.h) -----------------------------------
// *ok = false: not applicable or a region overflowed; the caller then uses the dense stamp method.
// T: the slots whose keys go through the filter -- the vertex table, or the log of the sweep (same 32-byte layout)
static int bloom_partitioned(rg_handle *h, Table &T, uint64_t m_bits, uint64_t keys, bool *ok) {
	*ok = false;
	const int dev = h->cfg.device;
	const uint64_t R = (m_bits + REGION_BITS - 1) / REGION_BITS;
	if (R > 65536 || T.cap >= (1ull << 31) || keys == 0) return RG_OK;
	const int nlevels = R > 256 ? 2 : 1;
	const int shift1 = nlevels == 2 ? 56 : 48;
	const uint64_t NBUCK = nlevels == 2 ? 65536 : 256;
	int n_cu = 256;
	(void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
	const int grid_items = (int)std::min<uint64_t>((uint64_t)n_cu * 8, (T.cap + 255) / 256);
	const uint64_t m_inv = (m_bits & (m_bits - 1)) ? (uint64_t)((((unsigned __int128)1) << 64) / m_bits) : 0;  // bloom_mod
	// dense key array (the sweep's log: every chunk full) -> items made straight into their first-level ranges; sparse
	// (a vertex table at load <= 0.7) -> items streamed out densely first, then one more partition level
	const bool fused = (h->dbg & 0x20000) || keys * 5 >= T.cap * 4;  // bit 17 forces it (tests)
	const uint64_t item_cap = fused ? 8 * keys : 8 * (keys + (uint64_t)grid_items * 4 * ITEM_CHUNK_KEYS);
	unsigned long long *A = nullptr, *B = nullptr, *d_seg = nullptr, *d_hist = nullptr, *d_begin = nullptr, *d_sums = nullptr;
	uint32_t *d_tile = nullptr, *d_lost = nullptr;
	uint8_t *d_side = nullptr;
	auto cleanup = [&]() {
		pool_free(dev, A);
		pool_free(dev, B);
		pool_free(dev, d_side);
		pool_free(dev, d_seg);
		pool_free(dev, d_hist);
		pool_free(dev, d_begin);
		pool_free(dev, d_sums);
		pool_free(dev, d_tile);
		pool_free(dev, d_lost);
	};
#define BTRY(call)                                                                                          \
	do {                                                                                                    \
		hipError_t e_ = (call);                                                                             \
		if (e_ != hipSuccess) {                                                                             \
			cleanup();                                                                                      \
			return fail(h, e_ == hipErrorOutOfMemory ? RG_ERR_NOMEM : RG_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
		}                                                                                                   \
	} while (0)
	if (pool_alloc(dev, (void **)&A, item_cap * 8) != hipSuccess || pool_alloc(dev, (void **)&B, 8 * keys * 8) != hipSuccess) {
		(void)hipGetLastError();
		cleanup();
		return RG_OK;
	}
	BTRY(pool_alloc(dev, (void **)&d_seg, 257 * sizeof(unsigned long long)));
	BTRY(pool_alloc(dev, (void **)&d_tile, 257 * sizeof(uint32_t)));
	BTRY(pool_alloc(dev, (void **)&d_hist, NBUCK * sizeof(unsigned long long)));
	BTRY(pool_alloc(dev, (void **)&d_begin, (NBUCK + 1) * sizeof(unsigned long long)));
	BTRY(pool_alloc(dev, (void **)&d_sums, (NBUCK / SCAN_PER_BLOCK + 2) * sizeof(unsigned long long)));
	const uint64_t lost_words = T.cap / 4 + 1;
	BTRY(pool_alloc(dev, (void **)&d_lost, lost_words * sizeof(uint32_t)));
	BTRY(hipMemsetAsync(d_lost, 0, lost_words * sizeof(uint32_t), h->stream));
	BTRY(hipMemsetAsync(d_hist, 0, 256 * sizeof(unsigned long long), h->stream));
	BTRY(hipMemsetAsync(h->d_counters + BC_ITEM_CURSOR, 0, 4 * sizeof(unsigned long long), h->stream));
	unsigned long long *bufs[2] = {A, B};
	int cur = 0, first_level = 0;
	std::vector<unsigned long long> seg_begin = {0, 0};
	std::vector<uint32_t> tile_first;
	if (fused) {
		// level 0: the digit histogram of all items first, then the items are produced straight into their ranges
		{
			ProfScope ps(h, "k_bloom_hist");
			if (h->W == 1) hipLaunchKernelGGL(k_bloom_hist<1>, dim3(grid_items), dim3(256), 0, h->stream, T.slots, T.cap, m_bits, m_inv, shift1, d_hist);
			else hipLaunchKernelGGL(k_bloom_hist<2>, dim3(grid_items), dim3(256), 0, h->stream, T.slots, T.cap, m_bits, m_inv, shift1, d_hist);
			BTRY(hipGetLastError());
		}
		hipLaunchKernelGGL(k_scan_local, dim3(1), dim3(256), 0, h->stream, d_hist, d_begin, (uint64_t)256, d_sums);
		hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(64), 0, h->stream, d_sums, 1u, d_sums + 1);
		hipLaunchKernelGGL(k_scan_add, dim3(1), dim3(256), 0, h->stream, d_begin, (uint64_t)256, d_sums, d_sums + 1, d_hist);
		{
			ProfScope ps(h, "k_bloom_scatter_items");
			const unsigned grid = (unsigned)((T.cap + ITEM_TILE_SLOTS - 1) / ITEM_TILE_SLOTS);
			// the second level's digits, one byte per item, for its histogram (rg_debug_set bit 20: it reads the items)
			if (nlevels > 1 && !(h->dbg & 0x100000) && pool_alloc(dev, (void **)&d_side, 8 * keys) != hipSuccess) {
				(void)hipGetLastError();
				d_side = nullptr;
			}
			if (h->W == 1) hipLaunchKernelGGL(k_bloom_scatter_items<1>, dim3(grid), dim3(SCATTER_THREADS), 0, h->stream, T.slots, T.cap, m_bits, m_inv, shift1, d_hist, bufs[1], d_side, 48);
			else hipLaunchKernelGGL(k_bloom_scatter_items<2>, dim3(grid), dim3(SCATTER_THREADS), 0, h->stream, T.slots, T.cap, m_bits, m_inv, shift1, d_hist, bufs[1], d_side, 48);
			BTRY(hipGetLastError());
		}
		cur = 1;
		first_level = 1;
		if (nlevels > 1) {
			seg_begin.resize(257);
			BTRY(hipMemcpyAsync(seg_begin.data(), d_begin, 257 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
			BTRY(rg_stream_sync(h));
			if (seg_begin[256] != 8 * keys) {
				cleanup();
				return fail(h, RG_ERR_STATE, "Bloom partition lost items: %llu of %llu", seg_begin[256], (unsigned long long)(8 * keys));
			}
		}
	} else {
		{
			ProfScope ps(h, "k_bloom_items");
			if (h->W == 1) hipLaunchKernelGGL(k_bloom_items<1>, dim3(grid_items), dim3(256), 0, h->stream, T.slots, T.cap, m_bits, A, item_cap, shift1, h->d_counters, d_hist);
			else hipLaunchKernelGGL(k_bloom_items<2>, dim3(grid_items), dim3(256), 0, h->stream, T.slots, T.cap, m_bits, A, item_cap, shift1, h->d_counters, d_hist);
			BTRY(hipGetLastError());
		}
		const int rc0 = sync_counters(h);
		if (rc0) {
			cleanup();
			return rc0;
		}
		if (h->h_counters[BC_ITEM_OVF]) {
			if (getenv("RG_DEBUG")) fprintf(stderr, "[raygraph] Bloom partition: item buffer overflow, using dense stamps\n");
			cleanup();
			return RG_OK;
		}
		seg_begin[1] = std::min<uint64_t>(h->h_counters[BC_ITEM_CURSOR], item_cap);
	}
	for (int level = first_level; level < nlevels; ++level) {
		const uint32_t S = (uint32_t)seg_begin.size() - 1;
		const int shift = level == 0 ? shift1 : 48;
		tile_first.assign(S + 1, 0);
		uint64_t ntiles = 0;
		for (uint32_t sgi = 0; sgi < S; ++sgi) {
			tile_first[sgi] = (uint32_t)ntiles;
			ntiles += (seg_begin[sgi + 1] - seg_begin[sgi] + SCATTER64_TILE - 1) / SCATTER64_TILE;
		}
		tile_first[S] = (uint32_t)ntiles;
		BTRY(hipMemcpyAsync(d_seg, seg_begin.data(), (S + 1) * sizeof(unsigned long long), hipMemcpyHostToDevice, h->stream));
		BTRY(hipMemcpyAsync(d_tile, tile_first.data(), (S + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
		const uint64_t nout = (uint64_t)S << 8;
		ProfScope ps(h, "k_scatter64");
		if (level > 0) {
			BTRY(hipMemsetAsync(d_hist, 0, nout * sizeof(unsigned long long), h->stream));
			if (level == 1 && d_side) hipLaunchKernelGGL(k_level_hist_side<SCATTER64_TILE>, dim3((unsigned)((ntiles + HIST_SIDE_TILES - 1) / HIST_SIDE_TILES)), dim3(256), 0, h->stream, d_side, S, d_seg, d_tile, 8, d_hist, (uint32_t)ntiles);
			else hipLaunchKernelGGL(k_hist64, dim3((unsigned)ntiles), dim3(256), 0, h->stream, bufs[cur], S, d_seg, d_tile, shift, d_hist);
		}
		const uint32_t nblk = (uint32_t)((nout + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK);
		hipLaunchKernelGGL(k_scan_local, dim3(nblk), dim3(256), 0, h->stream, d_hist, d_begin, nout, d_sums);
		hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(64), 0, h->stream, d_sums, nblk, d_sums + nblk);
		hipLaunchKernelGGL(k_scan_add, dim3(nblk), dim3(256), 0, h->stream, d_begin, nout, d_sums, d_sums + nblk, d_hist);
		hipLaunchKernelGGL(k_scatter64, dim3((unsigned)ntiles), dim3(SCATTER_THREADS), 0, h->stream, bufs[cur], S, d_seg, d_tile, shift, d_hist, bufs[cur ^ 1]);
		BTRY(hipGetLastError());
		cur ^= 1;
		if (level + 1 < nlevels) {
			seg_begin.resize(nout + 1);
			BTRY(hipMemcpyAsync(seg_begin.data(), d_begin, (nout + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
			BTRY(rg_stream_sync(h));
			if (seg_begin[nout] != 8 * keys) {
				cleanup();
				return fail(h, RG_ERR_STATE, "Bloom partition lost items: %llu of %llu", seg_begin[nout], (unsigned long long)(8 * keys));
			}
		}
	}
	BTRY(hipMemsetAsync(h->d_counters + C_SETBITS, 0, sizeof(unsigned long long), h->stream));
	{
		ProfScope ps(h, "k_bloom_region");
		// expected contested bits of a region: 2^17 * P(a bit has two or more touchers) at fill lambda
		const double lambda = 8.0 * (double)keys / (double)m_bits;
		const double contested = (double)REGION_BITS * (1.0 - std::exp(-lambda) * (1.0 + lambda));
		const bool big = contested > 0.8 * REGION_MULTI_CAP;
		auto launch = [&](auto kernel) {
			int occ = 1;
			if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, 512, 0) != hipSuccess || occ < 1) occ = 1;
			(void)hipGetLastError();
			const unsigned grid = (unsigned)std::min<uint64_t>(NBUCK, (uint64_t)n_cu * occ);
			hipLaunchKernelGGL(kernel, dim3(grid), dim3(512), 0, h->stream, bufs[cur], d_begin, (uint32_t)NBUCK, T.slots, d_lost, h->d_counters, (int)C_SETBITS, bufs[cur ^ 1]);
		};
		if (h->W == 1) {
			if (big) launch(k_bloom_region<1, REGION_MULTI_CAP_BIG>);
			else launch(k_bloom_region<1, REGION_MULTI_CAP>);
		} else {
			if (big) launch(k_bloom_region<2, REGION_MULTI_CAP_BIG>);
			else launch(k_bloom_region<2, REGION_MULTI_CAP>);
		}
		BTRY(hipGetLastError());
	}
	int rc = sync_counters(h);
	if (rc) {
		cleanup();
		return rc;
	}
	if (getenv("RG_DEBUG"))
		fprintf(stderr, "[raygraph] Bloom partition: %llu keys, %llu bits, %llu regions in %d level(s), %llu regions too contested\n",
		        (unsigned long long)keys, (unsigned long long)m_bits, (unsigned long long)R, nlevels, (unsigned long long)h->h_counters[BC_REGION_OVF]);
	if (h->h_counters[BC_REGION_OVF]) {  // a region had more contested bits than LDS can rank: dense method
		(void)hipMemsetAsync(h->d_counters + C_SETBITS, 0, sizeof(unsigned long long), h->stream);
		cleanup();
		return RG_OK;
	}
	{
		ProfScope ps(h, "k_admit");
		const int grid = grid_for(T.cap);
		if (h->W == 1) hipLaunchKernelGGL(k_admit_lost<1>, dim3(grid), dim3(BLOCK), 0, h->stream, T.slots, T.cap, d_lost, h->d_counters, h->d_hist_dense, h->d_hist_ovf);
		else hipLaunchKernelGGL(k_admit_lost<2>, dim3(grid), dim3(BLOCK), 0, h->stream, T.slots, T.cap, d_lost, h->d_counters, h->d_hist_dense, h->d_hist_ovf);
		BTRY(hipGetLastError());
	}
	BTRY(rg_stream_sync(h));
	cleanup();
	*ok = true;
#undef BTRY
	return RG_OK;
}

// ---- Bloom admission for power-of-two filters at any fill (rg_bloomx.h) ------------------------------------
// *ok = false: not applicable (filter size not a power of two, no memory); the caller then uses another method.
// keep_lost != nullptr: the admission kernel is not run; the marks (one byte per slot of T, 8 = false positive) are handed to
// the caller, who admits while building the table and frees them with pool_free.
// counted: the first level's ranges come from a histogram pass (k_xor_hist) instead of being sized in advance.
// The pending log in its full form, if it is a compact one (AggItems::compact): for every route but the one-pass admission.
// with_stamps: the arrival stamps come back from the items k_aggregate made beside the log, which must still be held (they
// stay valid for the expanded log: same positions); without, the stamps are INF_STAMP -- enough once the admission is decided.
static int expand_pending_log(rg_handle *h, bool with_stamps) {
	if (!h->pending_log || !h->pending_compact) return RG_OK;
	const int dev = h->cfg.device;
	const bool items = h->held.valid && h->held.log == h->pending_log;
	if (with_stamps && !items) return fail(h, RG_ERR_STATE, "a compact log without its items");
	Slot *full = nullptr;
	if (pool_alloc(dev, (void **)&full, h->pending_n * sizeof(Slot)) != hipSuccess) {
		(void)hipGetLastError();
		return fail(h, RG_ERR_NOMEM, "out of device memory expanding the log");
	}
	{
		ProfScope ps(h, "k_expand_log");
		hipLaunchKernelGGL(k_expand_entries, dim3((unsigned)std::min<uint64_t>((h->pending_n + 255) / 256, 1u << 16)), dim3(256), 0, h->stream, (const Ent16 *)h->pending_log, h->pending_n, full);
		if (with_stamps)
			hipLaunchKernelGGL(k_expand_stamps, dim3(AGG_ITEM_RUNS * h->held.n_wg), dim3(256), 0, h->stream, h->held.items, h->held.d_count, h->held.run_cap, full, h->pending_n);
	}
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) e = rg_stream_sync(h);
	if (e != hipSuccess) {
		pool_free(dev, full);
		return fail(h, RG_ERR_HIP, "expanding the log: %s", hipGetErrorString(e));
	}
	pool_free(dev, h->pending_log);
	if (items) h->held.log = full;
	h->pending_log = full;
	h->pending_compact = false;
	h->stats.compact_log = 2;
	return RG_OK;
}

static int bloom_xor(rg_handle *h, Table &T, uint64_t m_bits, uint64_t keys, bool *ok, uint8_t **keep_lost = nullptr, bool counted = false) {
	*ok = false;
	if (h->dbg & 0x10000000u) counted = true;  // rg_debug_set bit 28 (tests)
	const int dev = h->cfg.device;
	if (m_bits == 0 || (m_bits & (m_bits - 1)) || T.cap >= (1ull << 32) || keys == 0) return RG_OK;
	int B = 0;
	while ((1ull << B) < m_bits) B++;
	// region size: 2^14 bits, one workgroup of 1024 threads per CU (C2 with 1 % errors: k_xor_region 58 ms; 2^13 bits x 2
	// workgroups of 512: 73 ms; 2^12 x 4 of 256: 77 ms)
	int rmax = 14;
	// Dense filters: a region's eight streams stay in registers between the two passes while each holds at most
	// 1024 x XOR_SPER keys; beyond that the streams would be read twice (C5's share per GPU: 5370 keys per stream at 2^14
	// bits), so the regions are halved instead -- the workgroup stays at 1024 threads, one per CU
	bool dense_small = false;
	while (rmax > 12 && B > rmax && (double)keys / (double)(1ull << (B - rmax)) > 0.85 * 1024 * XOR_SPER) {
		rmax--;
		dense_small = true;
	}
	const int R = std::min(B, rmax);
	const int PB = B - R;  // partition bits
	if (PB > 24) return RG_OK;
	const uint64_t P = 1ull << PB;
	int c1 = 0, c2 = 0, c3 = PB;
	if (PB > 16) {
		c1 = PB / 3;
		c2 = (PB - c1) / 2;
		c3 = PB - c1 - c2;
	} else if (PB > 8) {
		c1 = PB / 2;
		c3 = PB - c1;
	}
	// the items of this very log and filter, made by k_aggregate while it wrote the log (AggItems): the first level (5 bits) is
	// done, in runs per workgroup of that kernel
	const bool from_items = h->held.valid && h->held.log == T.slots && h->held.m_bits == m_bits && h->held.total == keys && !counted && PB - 5 > 8 && PB - 5 <= 16;
	if (h->pending_compact && T.slots == h->pending_log && !(from_items && keep_lost)) {  // this route reads the log, or admits in place
		const int rcx = expand_pending_log(h, true);
		if (rcx) return rcx;
		T.slots = h->pending_log;
		return bloom_xor(h, T, m_bits, keys, ok, keep_lost, counted);
	}
	if (!from_items) drop_held_items(h);  // (whatever is held belongs to another log or filter)
	rg_handle::HeldItems held;            // owned here from now on
	h->stats.items_from_sweep = from_items ? 1 : 0;
	if (from_items) {
		held = h->held;
		h->held = rg_handle::HeldItems();
		c1 = 5;
		c2 = (PB - 5) / 2;
		c3 = PB - 5 - c2;
	}
	const uint32_t item_runs = from_items ? AGG_ITEM_RUNS * held.n_wg : 0;
	const uint64_t m_mask = m_bits - 1;
	int n_cu = 256;
	(void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
	const uint64_t npos_cap = keys + 64 * P;  // every partition is padded to whole wavefronts
	Item16 *A = nullptr, *Bb = nullptr;
	unsigned long long *pk = nullptr, *planes = nullptr, *d_seg = nullptr, *d_hist = nullptr, *d_begin = nullptr, *d_sums = nullptr;
	uint32_t *ix = nullptr, *d_tile = nullptr;
	uint8_t *d_side = nullptr, *d_lost = nullptr;
	unsigned long long *d_lim = nullptr;  // ranges sized in advance: their ends for k_xor_keys, then what was filled of each
	// The first level's digit is the top c1 bits of a hash of distinct keys: every range gets its share of the keys plus eight
	// standard deviations and a little, and the histogram pass over the key array is not needed (C5's share per GPU: 14.5 ms).
	// A range that overflows all the same is noticed and the counted way taken.
	const uint64_t S1 = 1ull << c1;
	const uint64_t slot1 = (keys / S1 + (uint64_t)(8.0 * std::sqrt((double)keys / (double)S1)) + 4096 + 63) & ~63ull;
	const uint64_t cap1 = from_items ? (uint64_t)item_runs * held.run_cap : (counted ? keys : S1 * slot1);  // items the first level's output holds
	Item16 *Bb_own = nullptr;  // from_items: the middle level's output is a block of its own (A is k_aggregate's)
	auto cleanup = [&]() {
		pool_free(dev, d_lim);
		pool_free(dev, Bb_own);
		pool_free(dev, held.d_count);
		if (from_items && (void *)d_side == (void *)held.side) d_side = nullptr;
		pool_free(dev, held.side);
		pool_free(dev, A);  // Bb is its second half
		if ((void *)pk != (void *)A) {
			pool_free(dev, pk);
			pool_free(dev, ix);
		}
		pool_free(dev, planes);
		pool_free(dev, d_seg);
		pool_free(dev, d_hist);
		pool_free(dev, d_begin);
		pool_free(dev, d_sums);
		pool_free(dev, d_tile);
		pool_free(dev, d_side);
		pool_free(dev, d_lost);
	};
	const uint64_t plane_words = (npos_cap + 63) / 64;
	const uint64_t NB = std::max<uint64_t>(P, 256);
	const uint64_t SEGMAX = 65536;
	// with a middle level the last one reads Bb and A is free by then: the packed words and slot indices (12 B per position)
	// take its place (16 B per key) when they fit -- at C5's share per GPU (3.5e9 keys) that is 42 GB less at the peak
	const bool alias = c2 && npos_cap * 12 + 256 <= keys * sizeof(Item16);
	hipError_t e = hipSuccess;
	if (from_items) {
		A = held.items;
		held.items = nullptr;
		e = pool_alloc(dev, (void **)&Bb_own, keys * sizeof(Item16));
		Bb = Bb_own;
	} else {
		e = pool_alloc(dev, (void **)&A, (cap1 + (c2 ? keys : 0)) * sizeof(Item16));  // both item buffers in one block (see build_table_ranges)
		if (e == hipSuccess && c2) Bb = A + cap1;
	}
	if (e == hipSuccess) e = pool_alloc(dev, (void **)&d_lim, std::max<uint64_t>(256, (uint64_t)item_runs + 1) * sizeof(unsigned long long));
	if (alias) {
		pk = (unsigned long long *)A;
		ix = (uint32_t *)((uint8_t *)A + ((npos_cap * 8 + 255) & ~255ull));
	} else {
		if (e == hipSuccess) e = pool_alloc(dev, (void **)&pk, npos_cap * 8);
		if (e == hipSuccess) e = pool_alloc(dev, (void **)&ix, npos_cap * 4);
	}
	if (e == hipSuccess) e = pool_alloc(dev, (void **)&planes, 8 * plane_words * 8);
	if (from_items) d_side = held.side;
	else if (e == hipSuccess) e = pool_alloc(dev, (void **)&d_side, cap1);
	if (e == hipSuccess) e = pool_alloc(dev, (void **)&d_lost, T.cap + 4);
	if (e == hipSuccess) e = pool_alloc(dev, (void **)&d_seg, (SEGMAX + 1) * sizeof(unsigned long long));
	if (e == hipSuccess) e = pool_alloc(dev, (void **)&d_tile, (SEGMAX + 1) * sizeof(uint32_t));
	if (e == hipSuccess) e = pool_alloc(dev, (void **)&d_hist, NB * sizeof(unsigned long long));
	if (e == hipSuccess) e = pool_alloc(dev, (void **)&d_begin, (NB + 1) * sizeof(unsigned long long));
	if (e == hipSuccess) e = pool_alloc(dev, (void **)&d_sums, (NB / SCAN_PER_BLOCK + 2) * sizeof(unsigned long long));
	if (e != hipSuccess) {
		(void)hipGetLastError();
		cleanup();
		if (h->pending_compact && T.slots == h->pending_log) return fail(h, RG_ERR_NOMEM, "out of device memory for the admission of a compact log");  // (its items went with the cleanup)
		return RG_OK;
	}
#define XTRY(call)                                                                                          \
	do {                                                                                                    \
		hipError_t e_ = (call);                                                                             \
		if (e_ != hipSuccess) {                                                                             \
			cleanup();                                                                                      \
			return fail(h, e_ == hipErrorOutOfMemory ? RG_ERR_NOMEM : RG_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
		}                                                                                                   \
	} while (0)
	auto scan = [&](uint64_t n) {  // d_hist[n] -> d_begin[n + 1], d_hist = copy of the begins (the scatter's cursors)
		const uint32_t nblk = (uint32_t)((n + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK);
		hipLaunchKernelGGL(k_scan_local, dim3(nblk), dim3(256), 0, h->stream, d_hist, d_begin, n, d_sums);
		hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(64), 0, h->stream, d_sums, nblk, d_sums + nblk);
		hipLaunchKernelGGL(k_scan_add, dim3(nblk), dim3(256), 0, h->stream, d_begin, n, d_sums, d_sums + nblk, d_hist);
	};
	if (!alias) XTRY(hipMemsetAsync(pk, 0xFF, npos_cap * 8, h->stream));
	XTRY(hipMemsetAsync(d_lost, 0, T.cap + 4, h->stream));
	XTRY(hipMemsetAsync(d_hist, 0, 256 * sizeof(unsigned long long), h->stream));
	// level 1: items produced straight into the ranges of the top c1 bits -- ranges sized in advance, or (counted) from a histogram pass
	std::vector<unsigned long long> seg_begin, seg_fill;  // seg_fill: where the filling of the level-1 ranges ended (sized in advance only)
	if (from_items) {  // done by k_aggregate: its runs are the ranges, filled as far as it counted
		seg_begin.resize((size_t)item_runs + 1);
		seg_fill.resize(item_runs);
		for (uint32_t r = 0; r <= item_runs; ++r) seg_begin[r] = (uint64_t)r * held.run_cap;
		for (uint32_t r = 0; r < item_runs; ++r) seg_fill[r] = seg_begin[r] + held.count[r];
		XTRY(hipMemcpyAsync(d_lim, seg_fill.data(), (size_t)item_runs * sizeof(unsigned long long), hipMemcpyHostToDevice, h->stream));
	} else if (counted) {
		const int grid_scan = (int)std::min<uint64_t>((uint64_t)n_cu * 8, (T.cap + 255) / 256);
		ProfScope ps(h, "k_xor_hist");
		if (h->W == 1) hipLaunchKernelGGL(k_xor_hist<1>, dim3(grid_scan), dim3(256), 0, h->stream, T.slots, T.cap, m_mask, B - c1, (1u << c1) - 1u, d_hist);
		else hipLaunchKernelGGL(k_xor_hist<2>, dim3(grid_scan), dim3(256), 0, h->stream, T.slots, T.cap, m_mask, B - c1, (1u << c1) - 1u, d_hist);
		XTRY(hipGetLastError());
		scan(S1);
	} else {
		seg_begin.resize(S1 + 1);
		seg_fill.resize(S1);
		for (uint64_t d = 0; d <= S1; ++d) seg_begin[d] = d * slot1;
		XTRY(hipMemcpyAsync(d_hist, seg_begin.data(), S1 * sizeof(unsigned long long), hipMemcpyHostToDevice, h->stream));      // the cursors
		XTRY(hipMemcpyAsync(d_lim, seg_begin.data() + 1, S1 * sizeof(unsigned long long), hipMemcpyHostToDevice, h->stream));  // the ends
		XTRY(hipMemsetAsync(h->d_counters + XC_PRESIZE_OVF, 0, sizeof(unsigned long long), h->stream));
	}
	const int nd = c2 ? c2 : c3;                 // bits of the level after the first
	const int nshift = B - c1 - nd;
	if (!from_items) {
		ProfScope ps(h, "k_xor_keys");
		const unsigned grid = (unsigned)((T.cap + XOR_TILE - 1) / XOR_TILE);
		const unsigned long long *lim = counted ? nullptr : d_lim;
		if (h->W == 1) hipLaunchKernelGGL(k_xor_keys<1>, dim3(grid), dim3(XOR_THREADS), 0, h->stream, T.slots, T.cap, m_mask, B - c1, (1u << c1) - 1u, d_hist, A, d_side, nshift, (1u << nd) - 1u, lim, h->d_counters);
		else hipLaunchKernelGGL(k_xor_keys<2>, dim3(grid), dim3(XOR_THREADS), 0, h->stream, T.slots, T.cap, m_mask, B - c1, (1u << c1) - 1u, d_hist, A, d_side, nshift, (1u << nd) - 1u, lim, h->d_counters);
		XTRY(hipGetLastError());
	}
	if (!counted && !from_items) {
		XTRY(hipMemcpyAsync(seg_fill.data(), d_hist, S1 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
		const int rcs = sync_counters(h);
		if (rcs) {
			cleanup();
			return rcs;
		}
		if (h->h_counters[XC_PRESIZE_OVF]) {  // a range sized in advance was too small (never seen): count them
			cleanup();
			return bloom_xor(h, T, m_bits, keys, ok, keep_lost, true);
		}
		XTRY(hipMemcpyAsync(d_lim, seg_fill.data(), S1 * sizeof(unsigned long long), hipMemcpyHostToDevice, h->stream));
	}
	std::vector<uint32_t> tile_first;
	const Item16 *level_in = A;
	int used = c1;
	for (int level = 1; level <= (c2 ? 2 : 1); ++level) {
		const bool final_level = level == (c2 ? 2 : 1);
		const int db = final_level ? c3 : c2;
		const bool runs = from_items && level == 1;  // the segments are k_aggregate's runs, held.n_wg of them per output segment
		const uint64_t S64 = runs ? item_runs : 1ull << used;
		const uint32_t seg_div = runs ? held.n_wg : 1u;
		const bool gaps = level == 1 && !counted;  // the segments are the ranges sized in advance: [seg_begin[s], seg_fill[s])
		uint64_t held = 0;
		if (!gaps) {
			seg_begin.resize(S64 + 1);
			XTRY(hipMemcpyAsync(seg_begin.data(), d_begin, (S64 + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
			XTRY(rg_stream_sync(h));
			held = seg_begin[S64];
		} else {
			for (uint64_t d = 0; d < S64; ++d) held += seg_fill[d] - seg_begin[d];
		}
		if (held != keys) {
			cleanup();
			return fail(h, RG_ERR_STATE, "Bloom key partition holds %llu of %llu keys", (unsigned long long)held, (unsigned long long)keys);
		}
		const uint32_t S = (uint32_t)S64;
		const unsigned long long *d_segend = gaps ? d_lim : nullptr;
		tile_first.assign(S + 1, 0);
		uint64_t ntiles = 0;
		for (uint32_t sgi = 0; sgi < S; ++sgi) {
			tile_first[sgi] = (uint32_t)ntiles;
			ntiles += ((gaps ? seg_fill[sgi] : seg_begin[sgi + 1]) - seg_begin[sgi] + XOR_TILE - 1) / XOR_TILE;
		}
		tile_first[S] = (uint32_t)ntiles;
		XTRY(hipMemcpyAsync(d_seg, seg_begin.data(), (S + 1) * sizeof(unsigned long long), hipMemcpyHostToDevice, h->stream));
		XTRY(hipMemcpyAsync(d_tile, tile_first.data(), (S + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
		const uint64_t nout = (uint64_t)(S / seg_div) << db;
		XTRY(hipMemsetAsync(d_hist, 0, nout * sizeof(unsigned long long), h->stream));
		{
			ProfScope ps(h, "k_level_hist_side");
			// (runs: k_aggregate left 8 bits of o per item, of which this level takes the top db)
			hipLaunchKernelGGL(k_level_hist_side<XOR_TILE>, dim3((unsigned)((ntiles + HIST_SIDE_TILES - 1) / HIST_SIDE_TILES)), dim3(256), 0, h->stream, d_side, S, d_seg, d_tile, db, d_hist, (uint32_t)ntiles, d_segend,
			                   seg_div, runs ? 8 - db : 0);
			XTRY(hipGetLastError());
		}
		if (final_level) hipLaunchKernelGGL(k_round64, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, h->stream, d_hist, nout);
		if (final_level && alias) XTRY(hipMemsetAsync(pk, 0xFF, npos_cap * 8, h->stream));  // A has been read by the middle level
		scan(nout);
		XTRY(hipGetLastError());
		{
			ProfScope ps(h, "k_scatter16");
			const int shift = B - used - db;
			if (final_level)
				hipLaunchKernelGGL((k_scatter16<true>), dim3((unsigned)ntiles), dim3(XOR_THREADS), 0, h->stream, level_in, S, d_seg, d_tile, shift, db, d_hist, (Item16 *)nullptr,
				                   (uint8_t *)nullptr, 0, 0u, pk, ix, (1ull << R) - 1ull, d_segend, seg_div);
			else
				hipLaunchKernelGGL((k_scatter16<false>), dim3((unsigned)ntiles), dim3(XOR_THREADS), 0, h->stream, level_in, S, d_seg, d_tile, shift, db, d_hist, Bb, d_side,
				                   shift - c3, (1u << c3) - 1u, (unsigned long long *)nullptr, (uint32_t *)nullptr, 0ull, d_segend, seg_div);
			XTRY(hipGetLastError());
		}
		used += db;
		level_in = Bb;
	}
	// regions
	XorConsts K;
	static const uint64_t MAGIC[8] = {0xe70b369c4c19f0f9ULL, 0xbeb7f38b993441a2ULL, 0x3149cd9246ca7995ULL, 0x2ef5b2c17d479ee8ULL,
	                                  0xfb8daceab90fe233ULL, 0x20fa74e37d497859ULL, 0x9007d0caef749698ULL, 0x4e4100a5605ef967ULL};  // BloomFilter.cpp:72-79
	for (int i = 0; i < 8; ++i) {
		K.c[i] = (uint32_t)((MAGIC[i] & m_mask) >> R);
		K.low[i] = (uint32_t)(MAGIC[i] & ((1ull << R) - 1ull));
	}
	XTRY(hipMemsetAsync(h->d_counters + C_SETBITS, 0, sizeof(unsigned long long), h->stream));
	// sparse filters (a stream of a region fits the workgroup with room to spare): the pipelined kernel
	bool distinct_bits = true;  // no two hashes of a key can name the same bit
	for (int i = 0; i < 8; ++i)
		for (int j = i + 1; j < 8; ++j)
			if ((MAGIC[i] & m_mask) == (MAGIC[j] & m_mask)) distinct_bits = false;
	const bool sparse = rmax == 14 && keys / P <= 512 && !(h->dbg & 0x20000000u);  // rg_debug_set bit 29: the dense kernel on a sparse filter (tests)
	if (!sparse && dense_small) {
		ProfScope ps(h, "k_xor_region");
		if (rmax == 13)
			hipLaunchKernelGGL((k_xor_region<13, 1024>), dim3((unsigned)std::min<uint64_t>(P, (uint64_t)n_cu)), dim3(1024), 0, h->stream, pk, d_begin, (uint32_t)P, R, K, planes, plane_words, h->d_counters, (int)C_SETBITS, distinct_bits ? 1 : 0);
		else
			hipLaunchKernelGGL((k_xor_region<12, 1024>), dim3((unsigned)std::min<uint64_t>(P, (uint64_t)n_cu)), dim3(1024), 0, h->stream, pk, d_begin, (uint32_t)P, R, K, planes, plane_words, h->d_counters, (int)C_SETBITS, distinct_bits ? 1 : 0);
		XTRY(hipGetLastError());
	} else
	if (sparse) {
		ProfScope ps(h, "k_xor_region");
		// a thread holds one key per stream: streams of up to 256 keys on average take 512 threads (fewer idle lanes, cheaper
		// barriers: 4.96 -> 4.74 ms on C2, 190 keys per stream; 256 threads: 4.88), fuller ones 1024
		if (keys / P <= 256)
			hipLaunchKernelGGL((k_xor_region_sparse<14, 512>), dim3((unsigned)std::min<uint64_t>(P, (uint64_t)n_cu)), dim3(512), 0, h->stream, pk, d_begin, (uint32_t)P, R, K, planes,
			                   plane_words, h->d_counters, (int)C_SETBITS, distinct_bits ? 1 : 0);
		else
			hipLaunchKernelGGL((k_xor_region_sparse<14, 1024>), dim3((unsigned)std::min<uint64_t>(P, (uint64_t)n_cu)), dim3(1024), 0, h->stream, pk, d_begin, (uint32_t)P, R, K, planes,
			                   plane_words, h->d_counters, (int)C_SETBITS, distinct_bits ? 1 : 0);
		XTRY(hipGetLastError());
	} else {
		ProfScope ps(h, "k_xor_region");
		if (rmax == 14)
			hipLaunchKernelGGL((k_xor_region<14, 1024>), dim3((unsigned)std::min<uint64_t>(P, (uint64_t)n_cu)), dim3(1024), 0, h->stream, pk, d_begin, (uint32_t)P, R, K, planes, plane_words, h->d_counters, (int)C_SETBITS, distinct_bits ? 1 : 0);
		else if (rmax == 13)
			hipLaunchKernelGGL((k_xor_region<13, 512>), dim3((unsigned)std::min<uint64_t>(P, (uint64_t)n_cu * 2)), dim3(512), 0, h->stream, pk, d_begin, (uint32_t)P, R, K, planes, plane_words, h->d_counters, (int)C_SETBITS, distinct_bits ? 1 : 0);
		else
			hipLaunchKernelGGL((k_xor_region<12, 256>), dim3((unsigned)std::min<uint64_t>(P, (uint64_t)n_cu * 4)), dim3(256), 0, h->stream, pk, d_begin, (uint32_t)P, R, K, planes, plane_words, h->d_counters, (int)C_SETBITS, distinct_bits ? 1 : 0);
		XTRY(hipGetLastError());
	}
	{
		ProfScope ps(h, "k_xor_fp");
		XTRY(hipMemsetAsync(h->d_counters + C_FP_COUNT, 0, sizeof(unsigned long long), h->stream));
		hipLaunchKernelGGL(k_xor_fp, dim3(grid_for(npos_cap)), dim3(256), 0, h->stream, pk, ix, planes, plane_words, npos_cap, d_lost, h->d_counters + C_FP_COUNT);
		XTRY(hipGetLastError());
	}
	if (!keep_lost) {
		ProfScope ps(h, "k_admit");
		const int grid = grid_for(T.cap);
		if (h->W == 1) hipLaunchKernelGGL(k_admit_lost<1>, dim3(grid), dim3(BLOCK), 0, h->stream, T.slots, T.cap, (const uint32_t *)d_lost, h->d_counters, h->d_hist_dense, h->d_hist_ovf);
		else hipLaunchKernelGGL(k_admit_lost<2>, dim3(grid), dim3(BLOCK), 0, h->stream, T.slots, T.cap, (const uint32_t *)d_lost, h->d_counters, h->d_hist_dense, h->d_hist_ovf);
		XTRY(hipGetLastError());
	}
	{  // (with the counters: the false positives counted by k_xor_fp bound what the admission stores)
		const int rcf = sync_counters(h);
		if (rcf) {
			cleanup();
			return rcf;
		}
	}
	if (keep_lost) {
		*keep_lost = d_lost;
		d_lost = nullptr;
	}
	if (getenv("RG_DEBUG"))
		fprintf(stderr, "[raygraph] Bloom by key partitions: %llu keys, 2^%d bits, %llu partitions (levels %d+%d+%d bits), regions of 2^%d bits\n",
		        (unsigned long long)keys, B, (unsigned long long)P, c1, c2, c3, R);
	cleanup();
	*ok = true;
#undef XTRY
	return RG_OK;
}

// rg_merge_entries, first call into an empty partition: the received entries are copied into a log of the library's own
// and those that do not carry the unique mark (or hold no key) are counted; none = the copy can be admitted as a log
__global__ __launch_bounds__(256) void k_adopt_entries(const Slot *__restrict__ in, uint64_t n, Slot *__restrict__ out,
                                                       unsigned long long *__restrict__ counters) {
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	uint32_t bad = 0, multi = 0;
	for (; i < n; i += stride) {
		const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(&in[i].k0), b = *reinterpret_cast<const ulonglong2 *>(&in[i].stamp);
		if (a.x == EMPTY || !(b.y & LOG_UNIQUE)) bad++;
		else if ((b.y & COUNT_MASK & ~LOG_UNIQUE) >= 2) multi++;  // (what admission stores whatever the filter says: sizes the table build)
		if (out) {  // no copy: the caller's buffer itself becomes the log (rg_adopt_entries)
			*reinterpret_cast<ulonglong2 *>(&out[i].k0) = a;
			*reinterpret_cast<ulonglong2 *>(&out[i].stamp) = b;
		}
	}
	if (bad) atomicAdd(&counters[C_ADOPT_BAD], (unsigned long long)bad);
	for (int d = 32; d > 0; d >>= 1) multi += __shfl_down(multi, d);
	if ((threadIdx.x & 63) == 0 && multi) atomicAdd(&counters[C_LOG_MULTI], (unsigned long long)multi);
}

// size == 1 with a pending log and a caller that wants the table of all candidates before admission (the owner
// exchange calls): the classic merge of the whole log
static int materialize_pending_log(rg_handle *h) {
	if (!h->pending_log) return RG_OK;
	if (int rcx = expand_pending_log(h, true)) return rcx;
	Table &T = h->tab;
	int rc = table_reserve(h, T, h->pending_n, C_FILL);
	if (rc) return rc;
	{
		ProfScope ps(h, "k_merge");
		if (h->W == 1) hipLaunchKernelGGL(k_merge<1>, dim3(grid_for(h->pending_n)), dim3(BLOCK), 0, h->stream, h->pending_log, h->pending_n, T.slots, T.cap, h->d_counters);
		else hipLaunchKernelGGL(k_merge<2>, dim3(grid_for(h->pending_n)), dim3(BLOCK), 0, h->stream, h->pending_log, h->pending_n, T.slots, T.cap, h->d_counters);
	}
	if (hipGetLastError() != hipSuccess) return fail(h, RG_ERR_HIP, "k_merge launch failed");
	rc = sync_counters(h);
	pool_free(h->cfg.device, h->pending_log);  // (a borrowed buffer is not the pool's: nothing happens)
	h->pending_log = nullptr;
	h->pending_borrowed = false;
	h->pending_n = 0;
	if (rc) return rc;
	T.fill_known = h->h_counters[C_FILL];
	T.fill_pending = 0;
	return RG_OK;
}

// reads the device histogram (dense bins + overflow list) back and caches it in the handle
// staged: the dense bins are in h->hist_stage and the counters on the host already (build_table_ranges, one-pass admission)
static int collect_histogram(rg_handle *h, const unsigned long long *dense, const uint32_t *ovf, bool staged = false) {
	std::vector<unsigned long long> hd_own;
	hipError_t e = hipSuccess;
	if (!staged) {
		hd_own.resize(HIST_DENSE);
		e = hipMemcpyAsync(hd_own.data(), dense, HIST_DENSE * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream);
		if (e != hipSuccess) return fail(h, RG_ERR_HIP, "histogram: %s", hipGetErrorString(e));
		int rc = sync_counters(h);
		if (rc) return rc;
	}
	const std::vector<unsigned long long> &hd = staged ? h->hist_stage : hd_own;
	const uint64_t novf = h->h_counters[C_OVF];
	std::vector<uint32_t> ho;
	if (novf > HIST_OVF_CAP) return fail(h, RG_ERR_CAPACITY, "more than %u vertices with coverage >= %u", HIST_OVF_CAP, HIST_DENSE);
	if (novf) {
		ho.resize(novf);
		e = hipMemcpy(ho.data(), ovf, novf * sizeof(uint32_t), hipMemcpyDeviceToHost);
		if (e != hipSuccess) return fail(h, RG_ERR_HIP, "histogram: %s", hipGetErrorString(e));
		std::sort(ho.begin(), ho.end());
	}
	h->hist_cache.clear();
	for (uint32_t c = 0; c < HIST_DENSE; ++c)
		if (hd[c]) h->hist_cache.emplace_back(c, hd[c]);
	for (size_t i = 0; i < ho.size();) {
		size_t e2 = i;
		while (e2 < ho.size() && ho[e2] == ho[i]) e2++;
		h->hist_cache.emplace_back(ho[i], e2 - i);
		i = e2;
	}
	h->hist_valid = true;
	return RG_OK;
}

// ---- the vertex table from the admitted entries of a log, range by range (rg_table.h) ---------------------
// *done = false: not applicable (too few entries, no memory, overflow list full); the caller then inserts the classic way.
// lost != nullptr: the log is as the sweep wrote it, lost[i] is entry i's mark and first_hist the admitted entries per top byte
// of the slot hash (k_admit_lost<W, false>): the first partition level applies the admission rule to what it reads.
// lost == nullptr: the log was admitted in place.
// fused: the first level is the admission's only pass over the log (k_scatter_entries with AdmitStats) -- `retained` is then an
// upper bound on the stored vertices (entries seen twice or more + false positives), the level writes into ranges sized in
// advance from it and counts the vertices, after which the table gets its real capacity.  *presize_failed: a range was too
// small; nothing was built, the statistics the pass gathered are void and the caller counts first, the classic way.
static int build_table_ranges(rg_handle *h, const Slot *log, uint64_t n_log, uint64_t retained, bool *done, const uint8_t *lost = nullptr,
                              const unsigned long long *first_hist = nullptr, bool fused = false, bool *presize_failed = nullptr, bool compact_log = false) {
	// compact_log: 16-byte entries (AggItems::compact), read by the fused first level only
	*done = false;
	if (compact_log && (!fused || h->W != 1)) return fail(h, RG_ERR_STATE, "a compact log on a route that does not read one");
	if (presize_failed) *presize_failed = false;
	const int dev = h->cfg.device;
	Table &T = h->tab;
	if (T.slots || retained == 0 || n_log == 0) return RG_OK;
	uint64_t cap = (uint64_t)((double)retained / 0.7) + 1;
	if (cap < 4096) cap = 4096;
	if (h->cfg.table_slots > cap) cap = h->cfg.table_slots;
	// 2^bits ranges of the slot hash, each at most RANGE_SPAN_MAX slots; the first level always takes the top 8 bits (what the
	// admission pass counted before the capacity was known), the rest is split over one or two more
	int bits = 8;
	while (bits < 25 && (cap >> bits) >= RANGE_SPAN_MAX) bits++;
	if (bits > 24) return RG_OK;
	// ranges of up to 2560 slots leave room for two workgroups per CU, whose zero / fill / write-out phases then overlap
	// (C5's share per GPU: 3147 slots, one workgroup per CU, 2.8 TB/s against C2's 3.6): one more bit where that adds no level
	if ((cap >> bits) >= 2560 && bits != 8 && bits != 16 && bits < 24) bits++;
	const uint64_t NR = 1ull << bits;
	const int rem = bits - 8;
	const int nlev = 1 + (rem > 0 ? 1 : 0) + (rem > 8 ? 1 : 0);
	int dbits[3] = {8, 0, 0};
	if (nlev == 2) dbits[1] = rem;
	if (nlev == 3) {
		dbits[1] = rem / 2;
		dbits[2] = rem - dbits[1];
	}
	const uint64_t NBUCK = 1ull << bits;
	if (fused && nlev < 2) {  // one level: its output must be the ranges themselves, without gaps
		if (presize_failed) *presize_failed = true;
		return RG_OK;
	}
	Slot *tab = nullptr, *ovf = nullptr;
	uint8_t *buf[2] = {nullptr, nullptr};  // admitted entries between the levels: Ent<W>, 16 bytes for one-word keys
	const size_t esz = h->W == 1 ? sizeof(Ent<1>) : sizeof(Ent<2>);
	const uint64_t TILE = h->W == 1 ? ent_tile<1>() : ent_tile<2>();
	unsigned long long *d_seg = nullptr, *d_hist = nullptr, *d_begin = nullptr, *d_sums = nullptr;
	uint32_t *d_tile = nullptr;
	uint8_t *d_side = nullptr;
	const uint64_t ovf_cap = retained / 16 + 4096;
	// fused: 256 ranges of the first level, an even share of the bound each plus eight standard deviations and a little
	const uint64_t slot0 = fused ? ((retained / 256 + (uint64_t)(8.0 * std::sqrt((double)retained / 256.0)) + 4096 + 63) & ~63ull) : 0;
	const uint64_t cap0 = fused ? 256 * slot0 : retained;  // entries the first level's output holds
	unsigned long long *d_lim = nullptr, *d_segend = nullptr, *d_spread = nullptr;
	auto cleanup = [&]() {
		pool_free(dev, d_lim);
		pool_free(dev, d_segend);
		pool_free(dev, d_spread);
		pool_free(dev, buf[0]);  // buf[1] is its second half
		pool_free(dev, ovf);
		pool_free(dev, d_seg);
		pool_free(dev, d_hist);
		pool_free(dev, d_begin);
		pool_free(dev, d_sums);
		pool_free(dev, d_tile);
		pool_free(dev, d_side);
	};
	const uint64_t SEGMAX = 65536;
	// (fused: the table is allocated once the first level has counted the vertices)
	hipError_t e = fused ? hipSuccess : pool_alloc(dev, (void **)&tab, cap * sizeof(Slot));
	// the two entry buffers are one block: a few large blocks of the same sizes every time are what the pool can hand out
	// again when the device is nearly full (C5's share per GPU), where many middling ones go back to the driver and return
	if (e == hipSuccess) e = pool_alloc(dev, (void **)&buf[0], (cap0 + (nlev > 1 ? retained : 0)) * esz);
	if (e == hipSuccess && nlev > 1) buf[1] = buf[0] + cap0 * esz;
	if (e == hipSuccess && fused) e = pool_alloc(dev, (void **)&d_lim, 256 * sizeof(unsigned long long));
	if (e == hipSuccess && fused) e = pool_alloc(dev, (void **)&d_segend, 257 * sizeof(unsigned long long));
	if (e == hipSuccess) e = pool_alloc(dev, (void **)&ovf, ovf_cap * sizeof(Slot));
	if (e == hipSuccess) e = pool_alloc(dev, (void **)&d_seg, (SEGMAX + 1) * sizeof(unsigned long long));
	if (e == hipSuccess) e = pool_alloc(dev, (void **)&d_tile, (SEGMAX + 1) * sizeof(uint32_t));
	if (e == hipSuccess) e = pool_alloc(dev, (void **)&d_hist, NBUCK * sizeof(unsigned long long));
	if (e == hipSuccess) e = pool_alloc(dev, (void **)&d_begin, (NBUCK + 1) * sizeof(unsigned long long));
	if (e == hipSuccess) e = pool_alloc(dev, (void **)&d_sums, (NBUCK / SCAN_PER_BLOCK + 2) * sizeof(unsigned long long));
	if (e == hipSuccess && nlev > 1) e = pool_alloc(dev, (void **)&d_side, cap0);
	if (e != hipSuccess) {  // not enough memory for the partition buffers: the classic insert needs none
		(void)hipGetLastError();
		cleanup();
		pool_free(dev, tab);
		return RG_OK;
	}
#define TTRY(call)                                                                                          \
	do {                                                                                                    \
		hipError_t e_ = (call);                                                                             \
		if (e_ != hipSuccess) {                                                                             \
			cleanup();                                                                                      \
			pool_free(dev, tab);                                                                            \
			return fail(h, e_ == hipErrorOutOfMemory ? RG_ERR_NOMEM : RG_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
		}                                                                                                   \
	} while (0)
	TTRY(hipMemsetAsync(h->d_counters + TC_OVF_CURSOR, 0, 2 * sizeof(unsigned long long), h->stream));
	TTRY(hipMemsetAsync(h->d_counters + C_FILL, 0, sizeof(unsigned long long), h->stream));
	std::vector<unsigned long long> seg_begin = {0, n_log}, seg_end;  // seg_end: where the ranges sized in advance were filled to (fused, second level)
	std::vector<uint32_t> tile_first;
	const void *level_in = log;
	int used_bits = 0, cur = 0;
	uint64_t stored = retained;  // fused: what the first level counted
	AdmitStats no_stats = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0};
	for (int level = 0; level < nlev; ++level) {
		const uint32_t S = (uint32_t)seg_begin.size() - 1;
		const int db = dbits[level];
		const int shift = bits - used_bits - db;
		const bool gaps = fused && level == 1;
		tile_first.assign(S + 1, 0);
		uint64_t ntiles = 0;
		for (uint32_t sgi = 0; sgi < S; ++sgi) {
			tile_first[sgi] = (uint32_t)ntiles;
			ntiles += ((gaps ? seg_end[sgi] : seg_begin[sgi + 1]) - seg_begin[sgi] + TILE - 1) / TILE;
		}
		tile_first[S] = (uint32_t)ntiles;
		TTRY(hipMemcpyAsync(d_seg, seg_begin.data(), (S + 1) * sizeof(unsigned long long), hipMemcpyHostToDevice, h->stream));
		TTRY(hipMemcpyAsync(d_tile, tile_first.data(), (S + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
		if (gaps) TTRY(hipMemcpyAsync(d_segend, seg_end.data(), S * sizeof(unsigned long long), hipMemcpyHostToDevice, h->stream));
		const unsigned long long *segend_arg = gaps ? d_segend : nullptr;
		const uint64_t nout = (uint64_t)S << db;
		TTRY(hipMemsetAsync(d_hist, 0, nout * sizeof(unsigned long long), h->stream));
		std::vector<unsigned long long> starts;  // fused first level: where its ranges start
		if (level == 0 && fused) {
			starts.resize(257);
			for (int d = 0; d <= 256; ++d) starts[d] = (uint64_t)d * slot0;
			TTRY(hipMemcpyAsync(d_hist, starts.data(), 256 * sizeof(unsigned long long), hipMemcpyHostToDevice, h->stream));      // the cursors
			TTRY(hipMemcpyAsync(d_lim, starts.data() + 1, 256 * sizeof(unsigned long long), hipMemcpyHostToDevice, h->stream));  // the ends
			TTRY(hipMemsetAsync(h->d_counters + C_ENT_PRESIZE, 0, sizeof(unsigned long long), h->stream));
		} else if (level == 0 && first_hist) {
			TTRY(hipMemcpyAsync(d_hist, first_hist, 256 * sizeof(unsigned long long), hipMemcpyDeviceToDevice, h->stream));
		} else if (level == 0) {
			ProfScope ps(h, "k_entry_hist");
			const int grid = (int)std::min<uint64_t>(2048, (n_log + 255) / 256);
			if (h->W == 1) hipLaunchKernelGGL(k_entry_hist<1>, dim3(grid), dim3(256), 0, h->stream, log, n_log, bits, shift, (1u << db) - 1u, d_hist);
			else hipLaunchKernelGGL(k_entry_hist<2>, dim3(grid), dim3(256), 0, h->stream, log, n_log, bits, shift, (1u << db) - 1u, d_hist);
		} else {
			ProfScope ps(h, "k_level_hist_side");
			if (h->W == 1) hipLaunchKernelGGL(k_level_hist_side<ent_tile<1>()>, dim3((unsigned)((ntiles + HIST_SIDE_TILES - 1) / HIST_SIDE_TILES)), dim3(256), 0, h->stream, d_side, S, d_seg, d_tile, db, d_hist, (uint32_t)ntiles, segend_arg);
			else hipLaunchKernelGGL(k_level_hist_side<ent_tile<2>()>, dim3((unsigned)((ntiles + HIST_SIDE_TILES - 1) / HIST_SIDE_TILES)), dim3(256), 0, h->stream, d_side, S, d_seg, d_tile, db, d_hist, (uint32_t)ntiles, segend_arg);
		}
		TTRY(hipGetLastError());
		if (!(level == 0 && fused)) {  // (the fused first level's cursors are its ranges' starts)
			const uint32_t nblk = (uint32_t)((nout + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK);
			hipLaunchKernelGGL(k_scan_local, dim3(nblk), dim3(256), 0, h->stream, d_hist, d_begin, nout, d_sums);
			hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(64), 0, h->stream, d_sums, nblk, d_sums + nblk);
			hipLaunchKernelGGL(k_scan_add, dim3(nblk), dim3(256), 0, h->stream, d_begin, nout, d_sums, d_sums + nblk, d_hist);
			TTRY(hipGetLastError());
		}
		AdmitStats st = no_stats;
		if (level == 0 && fused) {
			TTRY(pool_alloc(dev, (void **)&d_spread, (size_t)ADMIT_SPREAD * 258 * sizeof(unsigned long long)));
			TTRY(hipMemsetAsync(d_spread, 0, (size_t)ADMIT_SPREAD * 258 * sizeof(unsigned long long), h->stream));
			st = AdmitStats{h->d_counters, h->d_hist_dense, h->d_hist_ovf, d_lim, d_spread, HIST_DENSE, HIST_OVF_CAP, (int)C_RETAINED, (int)C_INSERTIONS, (int)C_OVF, (int)C_ENT_PRESIZE, compact_log ? 1u : 0u};
		}
		{
			ProfScope ps(h, "k_scatter_entries");
			const bool have_next = level + 1 < nlev;
			const int ndb = have_next ? dbits[level + 1] : 0;
			uint8_t *side_out = have_next ? d_side : nullptr;
			uint8_t *out = buf[cur];
			if (h->W == 1) {
				if (level == 0) hipLaunchKernelGGL((k_scatter_entries<1, true>), dim3((unsigned)ntiles), dim3(ENT_THREADS), 0, h->stream, (const Slot *)level_in, S, d_seg, d_tile, bits, shift, db, d_hist, (Ent<1> *)out, side_out, shift - ndb, (1u << ndb) - 1u, lost, st);
				else hipLaunchKernelGGL((k_scatter_entries<1, false>), dim3((unsigned)ntiles), dim3(ENT_THREADS), 0, h->stream, (const Ent<1> *)level_in, S, d_seg, d_tile, bits, shift, db, d_hist, (Ent<1> *)out, side_out, shift - ndb, (1u << ndb) - 1u, (const uint8_t *)nullptr, no_stats, segend_arg);
			} else {
				if (level == 0) hipLaunchKernelGGL((k_scatter_entries<2, true>), dim3((unsigned)ntiles), dim3(ENT_THREADS), 0, h->stream, (const Slot *)level_in, S, d_seg, d_tile, bits, shift, db, d_hist, (Ent<2> *)out, side_out, shift - ndb, (1u << ndb) - 1u, lost, st);
				else hipLaunchKernelGGL((k_scatter_entries<2, false>), dim3((unsigned)ntiles), dim3(ENT_THREADS), 0, h->stream, (const Ent<2> *)level_in, S, d_seg, d_tile, bits, shift, db, d_hist, (Ent<2> *)out, side_out, shift - ndb, (1u << ndb) - 1u, (const uint8_t *)nullptr, no_stats, segend_arg);
			}
		}
		TTRY(hipGetLastError());
		if (level == 0 && fused) hipLaunchKernelGGL(k_fold_admit_stats, dim3(1), dim3(320), 0, h->stream, st);
		used_bits += db;
		level_in = buf[cur];
		cur ^= 1;
		if (level == 0 && fused) {  // what the pass counted: the vertices stored, and how far each range was filled
			seg_end.resize(257);
			TTRY(hipMemcpyAsync(seg_end.data(), d_hist, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
			const int rcs = sync_counters(h);
			if (rcs) {
				cleanup();
				return rcs;
			}
			if (h->h_counters[C_ENT_PRESIZE]) {  // a range sized in advance was too small: the caller counts first
				cleanup();
				if (presize_failed) *presize_failed = true;
				return RG_OK;
			}
			stored = h->h_counters[C_RETAINED];
			seg_begin = starts;
			uint64_t held = 0;
			for (int d = 0; d < 256; ++d) held += seg_end[d] - seg_begin[d];
			if (held != stored || stored > retained) {
				cleanup();
				return fail(h, RG_ERR_STATE, "the admission pass stored %llu vertices, its ranges hold %llu (bound %llu)", (unsigned long long)stored, (unsigned long long)held, (unsigned long long)retained);
			}
			if (stored == 0) {  // nothing admitted: an empty table, made by the caller
				cleanup();
				return RG_OK;
			}
			cap = (uint64_t)((double)stored / 0.7) + 1;
			if (cap < 4096) cap = 4096;
			if (h->cfg.table_slots > cap) cap = h->cfg.table_slots;
			TTRY(pool_alloc(dev, (void **)&tab, cap * sizeof(Slot)));
		} else if (level + 1 < nlev) {
			seg_begin.resize(nout + 1);
			TTRY(hipMemcpyAsync(seg_begin.data(), d_begin, (nout + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
			TTRY(rg_stream_sync(h));
			if (seg_begin[nout] != stored) {
				cleanup();
				pool_free(dev, tab);
				return fail(h, RG_ERR_STATE, "table partition level %d holds %llu of %llu admitted entries", level, seg_begin[nout], (unsigned long long)stored);
			}
		}
	}
	{
		ProfScope ps(h, "k_build_ranges");
		int n_cu = 256;
		(void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
		const size_t lds = (size_t)((cap >> bits) + 2) * sizeof(Slot);
		const void *fn = h->W == 1 ? reinterpret_cast<const void *>(&k_build_ranges<1>) : reinterpret_cast<const void *>(&k_build_ranges<2>);
		TTRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		int occ = 1;
		if ((h->W == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_build_ranges<1>, 1024, lds)
		               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_build_ranges<2>, 1024, lds)) != hipSuccess || occ < 1)
			occ = 1;
		const unsigned grid = (unsigned)std::min<uint64_t>(NR, (uint64_t)n_cu * occ);
		if (h->W == 1) hipLaunchKernelGGL(k_build_ranges<1>, dim3(grid), dim3(1024), lds, h->stream, (const Ent<1> *)level_in, d_begin, (uint32_t)NR, bits, tab, cap, ovf, ovf_cap, h->d_counters);
		else hipLaunchKernelGGL(k_build_ranges<2>, dim3(grid), dim3(1024), lds, h->stream, (const Ent<2> *)level_in, d_begin, (uint32_t)NR, bits, tab, cap, ovf, ovf_cap, h->d_counters);
	}
	TTRY(hipGetLastError());
	if (fused && h->d_hist_dense) {  // the admission's histogram is complete since the first level: it travels with this wait
		h->hist_stage.resize(HIST_DENSE);
		TTRY(hipMemcpyAsync(h->hist_stage.data(), h->d_hist_dense, HIST_DENSE * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
	}
	int rc = sync_counters(h);
	if (rc) {
		cleanup();
		pool_free(dev, tab);
		return rc;
	}
	h->hist_staged = fused && h->d_hist_dense;
	const uint64_t n_ovf = h->h_counters[TC_OVF_CURSOR];
	if (h->h_counters[TC_OVF_LOST]) {  // more entries left their range than the list holds (never seen): build the classic way
		cleanup();
		pool_free(dev, tab);
		TTRY(hipMemsetAsync(h->d_counters + C_FILL, 0, sizeof(unsigned long long), h->stream));
		return RG_OK;
	}
	if (n_ovf) {
		ProfScope ps(h, "k_insert_admitted");
		if (h->W == 1) hipLaunchKernelGGL(k_insert_admitted<1>, dim3(grid_for(n_ovf)), dim3(BLOCK), 0, h->stream, ovf, n_ovf, tab, cap, h->d_counters);
		else hipLaunchKernelGGL(k_insert_admitted<2>, dim3(grid_for(n_ovf)), dim3(BLOCK), 0, h->stream, ovf, n_ovf, tab, cap, h->d_counters);
		TTRY(hipGetLastError());
		rc = sync_counters(h);
		if (rc) {
			cleanup();
			pool_free(dev, tab);
			return rc;
		}
	}
	if (getenv("RG_DEBUG"))
		fprintf(stderr, "[raygraph] table of %llu slots built from %llu admitted entries in %llu ranges of up to %u slots (%d partition levels), %llu entries left their range\n",
		        (unsigned long long)cap, (unsigned long long)retained, (unsigned long long)NR, (unsigned)(cap >> bits) + 1u, nlev, (unsigned long long)n_ovf);
	cleanup();
	T.slots = tab;
	T.cap = cap;
	T.fill_known = h->h_counters[C_FILL];
	T.fill_pending = 0;
	*done = true;
#undef TTRY
	return RG_OK;
}

// a temporary pool block that goes back to the pool however the scope is left (HIP_TRY returns early)
struct PoolBlock {
	int dev;
	void *p = nullptr;
	explicit PoolBlock(int d) : dev(d) {}
	PoolBlock(const PoolBlock &) = delete;
	PoolBlock &operator=(const PoolBlock &) = delete;
	~PoolBlock() {
		if (p) pool_free(dev, p);
	}
	hipError_t alloc(size_t bytes) { return pool_alloc(dev, &p, bytes); }
	void *release() {
		void *q = p;
		p = nullptr;
		return q;
	}
	template <class T>
	T *as() const {
		return (T *)p;
	}
};

// ---- hooks for rg_comm.cpp (rg_internal.h) ---------------------------------------------------------------------
#include "rg_internal.h"
namespace rgi {
int fail(rg_handle *h, int code, const char *fmt, ...) {
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	return ::fail(h, code, "%s", buf);
}
hipStream_t stream_of(rg_handle *h) { return h->stream; }
int device_of(const rg_handle *h) { return h->cfg.device; }
int rank_of(const rg_handle *h) { return h->cfg.rank; }
int size_of(const rg_handle *h) { return h->cfg.size; }
int kmer_words_of(const rg_handle *h) { return h->W; }
uint64_t reads_of(const rg_handle *h) { return h->n_reads; }
bool has_local(const rg_handle *h) { return h->local.slots != nullptr; }
hipError_t pool_get(int device, void **p, uint64_t bytes) { return pool_alloc(device, p, bytes); }
void pool_put(int device, void *p) { pool_free(device, p); }
int adopt_pool_block(rg_handle *h, void *d_entries, uint64_t n_entries, bool *adopted) {
	*adopted = false;
	const int rc = rg_adopt_entries(h, d_entries, n_entries);
	if (rc == RG_OK) *adopted = n_entries > 0 && h->pending_log == (Slot *)d_entries;
	return rc;
}
}  // namespace rgi

extern "C" {

const char *rg_version(void) { return "raygraph-mi355x 0.1 (gfx950)"; }

const char *rg_last_error(const rg_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int rg_create(const rg_config *cfg, rg_handle **out) {
	if (!cfg || !out) return fail(nullptr, RG_ERR_ARG, "null argument");
	*out = nullptr;
	int W = cfg->kmer_words;
	if (W == 0) W = cfg->k <= 31 ? 1 : 2;
	if (W != 1 && W != 2) return fail(nullptr, RG_ERR_ARG, "kmer_words must be 0, 1 or 2");
	if (cfg->k < 1 || (cfg->k % 2) == 0 || cfg->k > 32 * W - 1)
		return fail(nullptr, RG_ERR_ARG, "k=%d must be odd and within 1..%d for %d-word k-mers", cfg->k, 32 * W - 1, W);
	if (cfg->size < 1 || cfg->rank < 0 || cfg->rank >= cfg->size) return fail(nullptr, RG_ERR_ARG, "bad rank/size %d/%d", cfg->rank, cfg->size);
	if (cfg->reserved0 != 0) return fail(nullptr, RG_ERR_ARG, "rg_config.reserved0 must be 0");
	int ndev = 0;
	hipError_t e = hipGetDeviceCount(&ndev);
	if (e != hipSuccess || ndev <= 0) return fail(nullptr, RG_ERR_HIP, "no HIP device available (%s); this library has no CPU path", hipGetErrorString(e));
	if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, RG_ERR_ARG, "device %d out of range (have %d)", cfg->device, ndev);
	e = hipSetDevice(cfg->device);
	if (e != hipSuccess) return fail(nullptr, RG_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
	rg_handle *h = new rg_handle();
	h->cfg = *cfg;
	h->W = W;
	if (cfg->stream) h->stream = (hipStream_t)cfg->stream;
	else {
		e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
		if (e != hipSuccess) {
			delete h;
			return fail(nullptr, RG_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
		}
		h->own_stream = true;
	}
	if (pool_alloc(h->cfg.device, (void **)&h->d_counters, sizeof(unsigned long long) * C_NUM) != hipSuccess ||
	    hipHostMalloc((void **)&h->h_counters, sizeof(unsigned long long) * C_NUM) != hipSuccess ||
	    pool_alloc(h->cfg.device, (void **)&h->d_owner, sizeof(unsigned long long) * 2 * (size_t)cfg->size) != hipSuccess) {
		rg_destroy(h);
		return fail(nullptr, RG_ERR_NOMEM, "allocating counters failed");
	}
	(void)hipMemsetAsync(h->d_counters, 0, sizeof(unsigned long long) * C_NUM, h->stream);
	if (cfg->device < 64) (void)noalloc_init(cfg->device);  // set-up of rg_pack_reads_noalloc (it checks the result itself)
	(void)hipGetLastError();
	h->owner_counts.assign((size_t)cfg->size, 0);
	*out = h;
	return RG_OK;
}

int rg_destroy(rg_handle *h) {
	if (!h) return RG_OK;
	(void)hipSetDevice(h->cfg.device);
	if (h->stream) (void)rg_stream_sync(h);
	for (auto &kv : h->prof)
		for (auto &p : kv.second) {
			(void)hipEventDestroy(p.first);
			(void)hipEventDestroy(p.second);
		}
	for (auto &b : h->reads)
		if (b.owned) {
			pool_free(h->cfg.device, (void *)b.d_packed);
			pool_free(h->cfg.device, (void *)b.d_off);
			pool_free(h->cfg.device, (void *)b.d_len);
		}
	table_free(h, h->tab);
	table_free(h, h->local);
	drop_held_items(h);
	if (h->d_counters) pool_free(h->cfg.device, h->d_counters);
	if (h->h_counters) (void)hipHostFree(h->h_counters);
	if (h->d_owner) pool_free(h->cfg.device, h->d_owner);
	if (h->d_purge_origin) pool_free(h->cfg.device, h->d_purge_origin);
	if (h->d_marker_qidx) pool_free(h->cfg.device, h->d_marker_qidx);
	if (h->d_marker_kprefix) pool_free(h->cfg.device, h->d_marker_kprefix);
	if (h->rec_buf) pool_free(h->cfg.device, h->rec_buf);
	if (h->pending_log) pool_free(h->cfg.device, h->pending_log);
	for (int i = 0; i < 2; ++i)
		if (h->export_stage[i]) (void)hipHostFree(h->export_stage[i]);
	if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
	delete h;
	return RG_OK;
}

int rg_add_reads(rg_handle *h, const uint8_t *packed, uint64_t packed_bytes, const uint64_t *byte_offsets,
                 const uint32_t *lengths, uint64_t n_reads) {
	if (!h) return RG_ERR_ARG;
	if (h->phase != PH_CREATED) return fail(h, RG_ERR_STATE, "reads must be added before counting");
	if (n_reads == 0) return RG_OK;
	if (!packed || !byte_offsets || !lengths) return fail(h, RG_ERR_ARG, "null read arrays");
	uint32_t maxlen = 0;
	for (uint64_t i = 0; i < n_reads; ++i) {
		if (lengths[i] > 65535u) return fail(h, RG_ERR_ARG, "read %llu longer than 65535", (unsigned long long)i);
		if (byte_offsets[i] + (lengths[i] + 3) / 4 > packed_bytes) return fail(h, RG_ERR_ARG, "read %llu outside the packed buffer", (unsigned long long)i);
		maxlen = std::max(maxlen, lengths[i]);
	}
	HIP_TRY(h, hipSetDevice(h->cfg.device));
	ReadBatch b;
	uint64_t padded = ((packed_bytes + 7) / 8) * 8 + 16;
	PoolBlock gp(h->cfg.device), gof(h->cfg.device), gl(h->cfg.device);
	HIP_TRY(h, gp.alloc(padded));
	HIP_TRY(h, gof.alloc(n_reads * sizeof(uint64_t)));
	HIP_TRY(h, gl.alloc(n_reads * sizeof(uint32_t)));
	HIP_TRY(h, hipMemsetAsync(gp.as<uint8_t>() + (padded - 24), 0, 24, h->stream));
	HIP_TRY(h, hipMemcpyAsync(gp.p, packed, packed_bytes, hipMemcpyHostToDevice, h->stream));
	HIP_TRY(h, hipMemcpyAsync(gof.p, byte_offsets, n_reads * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
	HIP_TRY(h, hipMemcpyAsync(gl.p, lengths, n_reads * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
	HIP_TRY(h, rg_stream_sync(h));
	b.d_packed = (const uint64_t *)gp.release();  // the handle owns them from here (rg_destroy frees owned batches)
	b.d_off = (const uint64_t *)gof.release();
	b.d_len = (const uint32_t *)gl.release();
	b.n = n_reads;
	b.first_id = h->cfg.first_read_id + h->n_reads;
	b.maxlen = maxlen;
	b.owned = true;
	h->reads.push_back(b);
	h->n_reads += n_reads;
	return RG_OK;
}

int rg_add_reads_device(rg_handle *h, const void *d_packed, uint64_t packed_bytes, const void *d_byte_offsets,
                        const void *d_lengths, uint64_t n_reads, uint32_t max_length) {
	if (!h) return RG_ERR_ARG;
	if (h->phase != PH_CREATED) return fail(h, RG_ERR_STATE, "reads must be added before counting");
	if (n_reads == 0) return RG_OK;
	if (!d_packed || !d_byte_offsets || !d_lengths) return fail(h, RG_ERR_ARG, "null read arrays");
	if (((uintptr_t)d_packed & 7) != 0) return fail(h, RG_ERR_ARG, "device packed buffer must be 8-byte aligned");
	if (max_length > 65535u) return fail(h, RG_ERR_ARG, "max_length above 65535");
	(void)packed_bytes;
	ReadBatch b;
	b.d_packed = (const uint64_t *)d_packed;
	b.d_off = (const uint64_t *)d_byte_offsets;
	b.d_len = (const uint32_t *)d_lengths;
	b.n = n_reads;
	b.first_id = h->cfg.first_read_id + h->n_reads;
	b.maxlen = max_length;
	b.owned = false;
	h->reads.push_back(b);
	h->n_reads += n_reads;
	return RG_OK;
}

int rg_count_local(rg_handle *h) {
	if (h) h->hist_valid = h->owner_counts_valid = false;
	if (!h) return RG_ERR_ARG;
	if (h->phase != PH_CREATED) return fail(h, RG_ERR_STATE, "rg_count_local called twice");
	HIP_TRY(h, hipSetDevice(h->cfg.device));
	const bool single = h->cfg.size == 1;
	Table &T = single ? h->tab : h->local;
	const int fill_counter = single ? C_FILL : C_FILL_LOCAL;
	const int k = h->cfg.k;
	// choose the sweep: partitioned (super-k-mers + LDS aggregation) for real inputs, direct for small
	// ones; rg_debug_set bit 0 forces the partitioned sweep, bit 1 the direct one (tests)
	{
		uint64_t occ_bound = 0;
		for (const ReadBatch &b : h->reads)
			if (b.maxlen >= (uint32_t)k) occ_bound += b.n * ((uint64_t)b.maxlen - k + 1);
		const uint32_t flags = (uint32_t)h->dbg & 0xFFu;
		if (!(flags & 2u) && ((flags & 1u) || occ_bound >= (1ull << 24))) {
			Slot *log = nullptr;
			uint64_t n_log = 0;
			bool ok = false;
			int rc = sweep_partitioned(h, &log, &n_log, &ok);
			if (rc) return rc;
			if (ok) {
				h->stats.sweep_partitioned = 1;
				if (single && h->log_all_unique && n_log && n_log < (1ull << 32)) {
					// admission first, insertion afterwards (rg_resolve_admission): only stored vertices reach the table
					h->pending_log = log;
					h->pending_compact = h->log_compact;
					h->pending_n = n_log;
					h->pending_keys = h->log_keys;
					h->pending_multi = h->log_multi;  // (the sweep read the counters after its last launch: no wait for them here)
				} else if (h->log_compact) {
					pool_free(h->cfg.device, log);
					return fail(h, RG_ERR_STATE, "a compact log that cannot be kept");  // (the sweep makes one only where it is)
				} else if (single) {
					rc = table_reserve(h, T, n_log ? n_log : 1, fill_counter);
					if (rc == RG_OK && n_log) {
						ProfScope ps(h, "k_merge");
						if (h->W == 1) hipLaunchKernelGGL(k_merge<1>, dim3(grid_for(n_log)), dim3(BLOCK), 0, h->stream, log, n_log, T.slots, T.cap, h->d_counters);
						else hipLaunchKernelGGL(k_merge<2>, dim3(grid_for(n_log)), dim3(BLOCK), 0, h->stream, log, n_log, T.slots, T.cap, h->d_counters);
						if (hipGetLastError() != hipSuccess) rc = fail(h, RG_ERR_HIP, "k_merge launch failed");
					}
					if (rc == RG_OK) rc = sync_counters(h);
					pool_free(h->cfg.device, log);
					if (rc) return rc;
				} else {
					h->local.slots = log;  // the log plays the local table's role: a flat array of entries
					h->local.cap = n_log;
					rc = sync_counters(h);
					if (rc) return rc;
				}
				T.fill_known = single ? h->h_counters[fill_counter] : n_log;
				T.fill_pending = 0;
				h->stats.reads = h->n_reads;
				h->stats.occurrences = h->h_counters[C_OCC];
				h->phase = PH_COUNTED_LOCAL;
				return RG_OK;
			}
			// an estimate was exceeded: forget what the attempt counted and sweep directly
			HIP_TRY(h, hipMemsetAsync(h->d_counters + C_OCC, 0, sizeof(unsigned long long), h->stream));
		}
	}
	const uint64_t LAUNCH_OCC = 1ull << 27;  // k-mer occurrences per launch (bounds table growth)
	for (const ReadBatch &b : h->reads) {
		if (b.maxlen < (uint32_t)k) continue;
		const uint64_t per_read = (uint64_t)b.maxlen - (uint64_t)k + 1;
		uint64_t reads_per_launch = std::max<uint64_t>(TILE_READS, LAUNCH_OCC / per_read);
		reads_per_launch = (reads_per_launch / TILE_READS) * TILE_READS;
		for (uint64_t r0 = 0; r0 < b.n; r0 += reads_per_launch) {
			uint64_t r1 = std::min(b.n, r0 + reads_per_launch);
			uint64_t bound = (r1 - r0) * per_read;
			int rc = table_reserve(h, T, bound, fill_counter);
			if (rc) return rc;
			uint64_t ntiles = (r1 - r0 + TILE_READS - 1) / TILE_READS;
			int grid = (int)std::min<uint64_t>(ntiles, 256 * 8);
			ProfScope ps(h, "k_count");
			if (h->W == 1)
				hipLaunchKernelGGL(k_count<1>, dim3(grid), dim3(BLOCK), 0, h->stream, b.d_packed, b.d_off, b.d_len, r0, r1,
				                   b.first_id + r0, k, T.slots, T.cap, h->d_counters, fill_counter);
			else
				hipLaunchKernelGGL(k_count<2>, dim3(grid), dim3(BLOCK), 0, h->stream, b.d_packed, b.d_off, b.d_len, r0, r1,
				                   b.first_id + r0, k, T.slots, T.cap, h->d_counters, fill_counter);
			HIP_TRY(h, hipGetLastError());
			T.fill_pending += bound;
		}
	}
	if (!T.slots) {  // no read long enough: keep an empty table so later phases work
		int rc = table_reserve(h, T, 1, fill_counter);
		if (rc) return rc;
	}
	int rc = sync_counters(h);
	if (rc) return rc;
	T.fill_known = h->h_counters[fill_counter];
	T.fill_pending = 0;
	h->stats.reads = h->n_reads;
	h->stats.occurrences = h->h_counters[C_OCC];
	h->phase = PH_COUNTED_LOCAL;
	return RG_OK;
}

int rg_records_local_part(rg_handle *h, uint32_t part, uint32_t nparts) {
	if (!h) return RG_ERR_ARG;
	if (nparts == 0 || part >= nparts) return fail(h, RG_ERR_ARG, "part %u of %u", part, nparts);
	if (h->phase != PH_CREATED) return fail(h, RG_ERR_STATE, "rg_records_local called after counting started");
	if (h->rec_buf) return fail(h, RG_ERR_STATE, "the records of the previous part have not been exported");
	HIP_TRY(h, hipSetDevice(h->cfg.device));
	int rc = h->W == 2 ? records_local_t<2, false>(h, part, nparts) : (rec_compact(h) ? records_local_t<1, true>(h, part, nparts) : records_local_t<1, false>(h, part, nparts));
	if (rc) return rc;
	if (part == 0) h->stats.occurrences = 0;
	h->stats.reads = h->n_reads;
	h->stats.occurrences += h->rec_slots ? h->h_counters[C_OCC] : 0;
	return RG_OK;
}

int rg_records_local(rg_handle *h) { return rg_records_local_part(h, 0, 1); }

int rg_record_counts(rg_handle *h, uint64_t *counts, uint32_t *record_bytes) {
	if (!h || !counts) return RG_ERR_ARG;
	if (h->rec_counts.size() != (size_t)h->cfg.size) return fail(h, RG_ERR_STATE, "rg_record_counts needs rg_records_local first");
	for (int r = 0; r < h->cfg.size; ++r) counts[r] = h->rec_counts[r];
	if (record_bytes) *record_bytes = rec_compact(h) ? (uint32_t)sizeof(Rec20) : (uint32_t)(8 * (2 * h->W + 1));
	return RG_OK;
}

int rg_export_records(rg_handle *h, void *d_out, uint64_t capacity_records) {
	if (!h) return RG_ERR_ARG;
	if (h->rec_counts.size() != (size_t)h->cfg.size) return fail(h, RG_ERR_STATE, "rg_export_records needs rg_records_local first");
	uint64_t total = 0;
	for (uint64_t c : h->rec_counts) total += c;
	if (total > capacity_records) return fail(h, RG_ERR_CAPACITY, "record buffer holds %llu records, need %llu", (unsigned long long)capacity_records, (unsigned long long)total);
	if (total && !d_out) return fail(h, RG_ERR_ARG, "null record buffer");
	HIP_TRY(h, hipSetDevice(h->cfg.device));
	if (h->W == 2) return export_records_t<2, false>(h, d_out);
	return rec_compact(h) ? export_records_t<1, true>(h, d_out) : export_records_t<1, false>(h, d_out);
}

int rg_count_records(rg_handle *h, const void *d_records, uint64_t n_records) {
	if (h) h->hist_valid = h->owner_counts_valid = false;
	if (!h || (n_records && !d_records)) return RG_ERR_ARG;
	if (h->phase != PH_CREATED) return fail(h, RG_ERR_STATE, "rg_count_records called after counting started");
	HIP_TRY(h, hipSetDevice(h->cfg.device));
	Slot *log = nullptr;
	uint64_t n_log = 0;
	bool ok = false;
	const uint32_t flags = (uint32_t)h->dbg & 0xFFu;
	if (!(flags & 2u)) {
		int rc = sweep_partitioned(h, &log, &n_log, &ok, d_records ? d_records : (const void *)h /* non-null, never read */, n_records);
		if (rc) return rc;
	}
	if (ok) {
		h->stats.sweep_partitioned = 1;
		h->local.slots = log;  // the log plays the local table's role: a flat array of entries
		h->local.cap = n_log;
		h->local.fill_known = n_log;
		h->local.fill_pending = 0;
	} else {
		// an estimate was exceeded (or the direct route was asked for): one table upsert per occurrence
		const int w = h->cfg.k - std::min(h->cfg.k, MMER_MAX) + 1;
		int rc = table_reserve(h, h->local, std::max<uint64_t>(1, n_records * (uint64_t)w), C_FILL_LOCAL);
		if (rc) return rc;
		if (n_records) {
			ProfScope ps(h, "k_count_records");
			if (h->W == 1 && rec_compact(h)) hipLaunchKernelGGL((k_count_records<1, true>), dim3(grid_for(n_records)), dim3(256), 0, h->stream, (const Rec20 *)d_records, n_records, h->cfg.k, h->local.slots, h->local.cap, h->d_counters, (int)C_FILL_LOCAL);
			else if (h->W == 1) hipLaunchKernelGGL(k_count_records<1>, dim3(grid_for(n_records)), dim3(256), 0, h->stream, (const Rec<1> *)d_records, n_records, h->cfg.k, h->local.slots, h->local.cap, h->d_counters, (int)C_FILL_LOCAL);
			else hipLaunchKernelGGL(k_count_records<2>, dim3(grid_for(n_records)), dim3(256), 0, h->stream, (const Rec<2> *)d_records, n_records, h->cfg.k, h->local.slots, h->local.cap, h->d_counters, (int)C_FILL_LOCAL);
			HIP_TRY(h, hipGetLastError());
		}
		rc = sync_counters(h);
		if (rc) return rc;
		h->local.fill_known = h->h_counters[C_FILL_LOCAL];
		h->local.fill_pending = 0;
	}
	h->phase = PH_COUNTED_LOCAL;
	return RG_OK;
}

int rg_owner_counts(rg_handle *h, uint64_t *counts) {
	if (!h || !counts) return RG_ERR_ARG;
	if (h->phase != PH_COUNTED_LOCAL) return fail(h, RG_ERR_STATE, "rg_owner_counts needs rg_count_local first");
	HIP_TRY(h, hipSetDevice(h->cfg.device));
	if (int rcm = materialize_pending_log(h)) return rcm;
	const uint32_t size = (uint32_t)h->cfg.size;
	Table &T = h->local.slots ? h->local : h->tab;  // size == 1 after rg_count_local: the table itself
	HIP_TRY(h, hipMemsetAsync(h->d_owner, 0, sizeof(unsigned long long) * 2 * size, h->stream));
	{
		ProfScope ps(h, "k_owner_counts");
		if (h->W == 1) hipLaunchKernelGGL(k_owner_counts<1>, dim3(grid_for(T.cap)), dim3(BLOCK), size * sizeof(uint32_t), h->stream, T.slots, T.cap, size, h->d_owner);
		else hipLaunchKernelGGL(k_owner_counts<2>, dim3(grid_for(T.cap)), dim3(BLOCK), size * sizeof(uint32_t), h->stream, T.slots, T.cap, size, h->d_owner);
	}
	HIP_TRY(h, hipGetLastError());
	std::vector<unsigned long long> tmp(size);
	HIP_TRY(h, hipMemcpyAsync(tmp.data(), h->d_owner, sizeof(unsigned long long) * size, hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(h, rg_stream_sync(h));
	for (uint32_t r = 0; r < size; ++r) h->owner_counts[r] = counts[r] = tmp[r];
	h->owner_counts_valid = true;
	return RG_OK;
}

int rg_export_owner_buckets(rg_handle *h, void *d_out, uint64_t capacity_entries) {
	if (!h) return RG_ERR_ARG;
	if (h->phase != PH_COUNTED_LOCAL) return fail(h, RG_ERR_STATE, "rg_export_owner_buckets needs rg_count_local first");
	HIP_TRY(h, hipSetDevice(h->cfg.device));
	if (int rcm = materialize_pending_log(h)) return rcm;
	const uint32_t size = (uint32_t)h->cfg.size;
	Table &T = h->local.slots ? h->local : h->tab;
	if (!h->owner_counts_valid) return fail(h, RG_ERR_STATE, "rg_export_owner_buckets needs rg_owner_counts first");
	std::vector<unsigned long long> cursors(size);
	uint64_t total = 0;
	for (uint32_t r = 0; r < size; ++r) {
		cursors[r] = total;
		total += h->owner_counts[r];
	}
	if (total > capacity_entries) return fail(h, RG_ERR_CAPACITY, "bucket buffer holds %llu entries, need %llu", (unsigned long long)capacity_entries, (unsigned long long)total);
	if (total == 0) return RG_OK;
	if (!d_out) return fail(h, RG_ERR_ARG, "null bucket buffer");
	HIP_TRY(h, hipMemcpyAsync(h->d_owner + size, cursors.data(), sizeof(unsigned long long) * size, hipMemcpyHostToDevice, h->stream));
	{
		ProfScope ps(h, "k_owner_scatter");
		if (h->W == 1) hipLaunchKernelGGL(k_owner_scatter<1>, dim3(grid_for(T.cap)), dim3(BLOCK), size * 12, h->stream, T.slots, T.cap, size, h->d_owner + size, (Slot *)d_out);
		else hipLaunchKernelGGL(k_owner_scatter<2>, dim3(grid_for(T.cap)), dim3(BLOCK), size * 12, h->stream, T.slots, T.cap, size, h->d_owner + size, (Slot *)d_out);
	}
	HIP_TRY(h, hipGetLastError());
	HIP_TRY(h, rg_stream_sync(h));
	return RG_OK;
}

int rg_merge_entries(rg_handle *h, const void *d_entries, uint64_t n_entries) {
	if (h) drop_held_items(h);  // (items k_aggregate made belong to the log of rg_count_local alone)
	if (h) h->hist_valid = false;
	if (!h) return RG_ERR_ARG;
	if (h->phase != PH_COUNTED_LOCAL && h->phase != PH_MERGING) return fail(h, RG_ERR_STATE, "rg_merge_entries needs rg_count_local first");
	// one rank: only after rg_count_records (a rank that sent its records to itself); after rg_count_local the table already is the partition
	if (h->cfg.size == 1 && h->phase == PH_COUNTED_LOCAL && !h->local.slots)
		return fail(h, RG_ERR_STATE, "rg_merge_entries is for size > 1 (the local table is already the partition)");
	HIP_TRY(h, hipSetDevice(h->cfg.device));
	h->owner_counts_valid = false;
	if (h->phase == PH_COUNTED_LOCAL) {
		table_free(h, h->local);  // the buckets hold everything the local table knew
		h->phase = PH_MERGING;
	}
	if (n_entries == 0) return RG_OK;
	if (!d_entries) return fail(h, RG_ERR_ARG, "null entries");
	// Record route: every key was aggregated by exactly one rank, so what the owner receives is a log of distinct
	// k-mers like the one a single rank makes for itself (the entries carry the unique mark).  If the whole partition
	// arrives in one call, it is kept as that log: admission runs on it and only admitted vertices reach the table
	// (rg_resolve_admission).  A further call, or entries without the mark, take the classic merge.
	if (!h->tab.slots && !h->pending_log && !h->merge_started && n_entries < (1ull << 32) && !(h->dbg & 0x80000)) {
		h->merge_started = true;
		Slot *copy = nullptr;
		Slot first;  // entries of the entry route carry no mark: one look spares them the copy
		HIP_TRY(h, hipMemcpyAsync(&first, d_entries, sizeof(Slot), hipMemcpyDeviceToHost, h->stream));
		HIP_TRY(h, rg_stream_sync(h));
		if ((first.meta & LOG_UNIQUE) && pool_alloc(h->cfg.device, (void **)&copy, n_entries * sizeof(Slot)) == hipSuccess) {
			hipError_t e = hipMemsetAsync(h->d_counters + C_ADOPT_BAD, 0, 2 * sizeof(unsigned long long), h->stream);  // (and C_LOG_MULTI behind it)
			if (e == hipSuccess) {
				ProfScope ps(h, "k_adopt_entries");
				hipLaunchKernelGGL(k_adopt_entries, dim3(grid_for(n_entries)), dim3(BLOCK), 0, h->stream, (const Slot *)d_entries, n_entries, copy, h->d_counters);
				e = hipGetLastError();
			}
			int rc0 = e == hipSuccess ? sync_counters(h) : fail(h, RG_ERR_HIP, "k_adopt_entries: %s", hipGetErrorString(e));
			if (rc0) {
				pool_free(h->cfg.device, copy);
				return rc0;
			}
			if (h->h_counters[C_ADOPT_BAD] == 0) {
				h->pending_log = copy;
				h->pending_n = h->pending_keys = n_entries;
				h->pending_multi = h->h_counters[C_LOG_MULTI];
				if (getenv("RG_DEBUG")) fprintf(stderr, "[raygraph] rank %d: %llu received entries kept as a log of distinct k-mers\n", h->cfg.rank, (unsigned long long)n_entries);
				return RG_OK;  // synchronised: the caller may reuse d_entries
			}
			pool_free(h->cfg.device, copy);
		} else {
			(void)hipGetLastError();
		}
	}
	h->merge_started = true;
	if (int rcm = materialize_pending_log(h)) return rcm;  // a second call: what was kept as a log goes to the table first
	int rc = table_reserve(h, h->tab, n_entries, C_FILL);
	if (rc) return rc;
	ProfScope ps(h, "k_merge");
	if (h->W == 1) hipLaunchKernelGGL(k_merge<1>, dim3(grid_for(n_entries)), dim3(BLOCK), 0, h->stream, (const Slot *)d_entries, n_entries, h->tab.slots, h->tab.cap, h->d_counters);
	else hipLaunchKernelGGL(k_merge<2>, dim3(grid_for(n_entries)), dim3(BLOCK), 0, h->stream, (const Slot *)d_entries, n_entries, h->tab.slots, h->tab.cap, h->d_counters);
	HIP_TRY(h, hipGetLastError());
	h->tab.fill_pending += n_entries;
	HIP_TRY(h, rg_stream_sync(h));  // the caller may now reuse d_entries
	return RG_OK;
}

int rg_release_local(rg_handle *h) {
	if (h) drop_held_items(h);  // (items k_aggregate made belong to the log of rg_count_local alone)
	if (!h) return RG_ERR_ARG;
	if (h->phase == PH_MERGING) return RG_OK;  // already released (by this call or by the first rg_merge_entries)
	if (h->phase != PH_COUNTED_LOCAL) return fail(h, RG_ERR_STATE, "rg_release_local needs rg_count_local / rg_count_records first");
	if (h->cfg.size == 1 && !h->local.slots) return fail(h, RG_ERR_STATE, "rg_release_local: on one rank the local table is the partition");
	HIP_TRY(h, hipSetDevice(h->cfg.device));
	HIP_TRY(h, rg_stream_sync(h));
	h->owner_counts_valid = false;
	table_free(h, h->local);
	h->phase = PH_MERGING;
	return RG_OK;
}

int rg_adopt_entries(rg_handle *h, void *d_entries, uint64_t n_entries) {
	if (h) drop_held_items(h);  // (items k_aggregate made belong to the log of rg_count_local alone)
	if (h) h->hist_valid = false;
	if (!h) return RG_ERR_ARG;
	if (h->phase != PH_COUNTED_LOCAL && h->phase != PH_MERGING) return fail(h, RG_ERR_STATE, "rg_adopt_entries needs rg_count_local first");
	if (n_entries && !d_entries) return fail(h, RG_ERR_ARG, "null entries");
	HIP_TRY(h, hipSetDevice(h->cfg.device));
	// in place only where rg_merge_entries would keep a copy as the log: first and only delivery, every entry a distinct key
	bool in_place = n_entries > 0 && !h->tab.slots && !h->pending_log && !h->merge_started && n_entries < (1ull << 32) && !(h->dbg & 0x80000) &&
	                (h->cfg.size > 1 || h->local.slots || h->phase == PH_MERGING);
	if (in_place) {  // entries of the entry route carry no mark: one look at the first spares them a pass over the buffer
		Slot first;
		HIP_TRY(h, hipMemcpyAsync(&first, d_entries, sizeof(Slot), hipMemcpyDeviceToHost, h->stream));
		HIP_TRY(h, rg_stream_sync(h));
		in_place = (first.meta & LOG_UNIQUE) != 0;
	}
	if (in_place) {
		hipError_t e = hipMemsetAsync(h->d_counters + C_ADOPT_BAD, 0, 2 * sizeof(unsigned long long), h->stream);  // (and C_LOG_MULTI behind it)
		if (e == hipSuccess) {
			ProfScope ps(h, "k_adopt_entries");
			hipLaunchKernelGGL(k_adopt_entries, dim3(grid_for(n_entries)), dim3(BLOCK), 0, h->stream, (const Slot *)d_entries, n_entries, (Slot *)nullptr, h->d_counters);
			e = hipGetLastError();
		}
		const int rc0 = e == hipSuccess ? sync_counters(h) : fail(h, RG_ERR_HIP, "k_adopt_entries: %s", hipGetErrorString(e));
		if (rc0) return rc0;
		in_place = h->h_counters[C_ADOPT_BAD] == 0;
	}
	if (!in_place) return rg_merge_entries(h, d_entries, n_entries);  // copies what it needs: the buffer is the caller's again on return
	h->owner_counts_valid = false;
	if (h->phase == PH_COUNTED_LOCAL) {
		table_free(h, h->local);
		h->phase = PH_MERGING;
	}
	h->merge_started = true;
	h->pending_log = (Slot *)d_entries;  // pool_free ignores pointers the pool did not hand out: the buffer stays the caller's
	h->pending_borrowed = true;
	h->pending_n = h->pending_keys = n_entries;
	h->pending_multi = h->h_counters[C_LOG_MULTI];
	if (getenv("RG_DEBUG")) fprintf(stderr, "[raygraph] rank %d: %llu received entries admitted in the caller's buffer\n", h->cfg.rank, (unsigned long long)n_entries);
	return RG_OK;
}

int rg_resolve_admission(rg_handle *h, uint64_t local_reads_for_bloom) {
	if (h) h->hist_valid = false;
	if (!h) return RG_ERR_ARG;
	if (h->cfg.size == 1 && h->phase != PH_MERGING && !(h->phase == PH_COUNTED_LOCAL && h->local.slots)) {
		if (h->phase != PH_COUNTED_LOCAL) return fail(h, RG_ERR_STATE, "rg_resolve_admission needs rg_count_local first");
	} else {
		if (h->phase != PH_MERGING && h->phase != PH_COUNTED_LOCAL) return fail(h, RG_ERR_STATE, "rg_resolve_admission needs rg_merge_entries first");
		if (h->phase == PH_COUNTED_LOCAL) table_free(h, h->local);
	}
	HIP_TRY(h, hipSetDevice(h->cfg.device));
	// Admission runs on the slots that hold the candidate keys: the vertex table, or -- one rank, partitioned sweep,
	// every key once in the log -- the log itself, whose admitted entries are inserted afterwards.
	const bool from_log = h->pending_log != nullptr;
	Table logview;
	logview.slots = h->pending_log;
	logview.cap = h->pending_n;
	Table &T = from_log ? logview : h->tab;
	if (!T.slots) {
		int rc = table_reserve(h, T, 1, C_FILL);
		if (rc) return rc;
	}
	// the admission kernel also gathers the coverage histogram and moves the edge bits to their final field
	HIP_TRY(h, pool_alloc(h->cfg.device, (void **)&h->d_hist_dense, HIST_DENSE * sizeof(unsigned long long)));
	if (pool_alloc(h->cfg.device, (void **)&h->d_hist_ovf, HIST_OVF_CAP * sizeof(uint32_t)) != hipSuccess) {
		pool_free(h->cfg.device, h->d_hist_dense);
		h->d_hist_dense = nullptr;
		return fail(h, RG_ERR_NOMEM, "rg_resolve_admission: out of device memory");
	}
	struct HistGuard {
		rg_handle *h;
		~HistGuard() {
			pool_free(h->cfg.device, h->d_hist_dense);
			pool_free(h->cfg.device, h->d_hist_ovf);
			h->d_hist_dense = nullptr;
			h->d_hist_ovf = nullptr;
		}
	} hist_guard{h};
	HIP_TRY(h, hipMemsetAsync(h->d_hist_dense, 0, HIST_DENSE * sizeof(unsigned long long), h->stream));
	HIP_TRY(h, hipMemsetAsync(h->d_counters + C_OVF, 0, sizeof(unsigned long long), h->stream));
	// MessageProcessor.cpp:886-899 + VerticesExtractor.cpp:276-320
	uint64_t m_bits = h->cfg.bloom_bits;
	if (m_bits == RG_BLOOM_DEFAULT) {
		uint64_t reads = local_reads_for_bloom == ~(uint64_t)0 ? h->n_reads : local_reads_for_bloom;
		m_bits = 4ull * ((uint64_t)h->cfg.k * 2 * 2) * reads;
		const uint64_t maximumBits = 1024ull * 1024 * 1024 * 8;
		if (m_bits > maximumBits) m_bits = maximumBits;
	}
	h->stats.bloom_bits = m_bits;
	const int grid = grid_for(T.cap);
	bool resolved = false;
	uint8_t *lost_kept = nullptr;             // marks of a Bloom resolution whose admission is deferred to the table build
	unsigned long long *d_first_hist = nullptr;
	// a compact log (AggItems::compact) is read by bloom_xor's start from the held items and by the one-pass admission; whatever
	// else is about to run gets the full form first
	auto full_log = [&](bool with_stamps) -> int {
		const int rcx = expand_pending_log(h, with_stamps);
		logview.slots = h->pending_log;
		return rcx;
	};
	if (m_bits == 0 && h->pending_compact)
		if (int rcx = full_log(true)) return rcx;
	if (m_bits > 0) {
		// keys in the table decide the method: region partitioning when the filter stays sparse
		// (8 touches per key), dense first-touch stamps otherwise.  rg_debug_set bit 4 forces the
		// first, bit 5 the second (tests).
		int rc0 = from_log ? RG_OK : sync_counters(h);  // (a log's keys were counted by whoever made it)
		if (rc0) return rc0;
		const uint64_t keys = from_log ? h->pending_keys : h->h_counters[C_FILL];
		const uint32_t flags = (uint32_t)h->dbg & 0xFFu;
		const double lambda = 8.0 * (double)keys / (double)m_bits;
		// worth it while a region's contested bits (expected: 2^17 * P(two or more touchers)) fit the LDS rank table
		const double contested = (double)REGION_BITS * (1.0 - std::exp(-lambda) * (1.0 + lambda));
		const bool regions_fit = contested <= 0.9 * REGION_MULTI_CAP_BIG;
		// filters of 2^B bits too full for the regions (or rg_debug_set bit 24: tests): one item per key, partitioned by the
		// key's origin, eight streams per region (rg_bloomx.h)
		const bool regions_cheap = contested <= 0.4 * REGION_MULTI_CAP;  // beyond: stamp gathers and several rounds per region (C2 with 0.2 % errors: 125 ms against 45)
		// ... and, since its regions are software-pipelined (k_xor_region_sparse), for sparse ones too: C2 error-free 12.1 ms
		// against the regions' 15.4.  The (bit, key) items of rg_bloom.h remain for filters whose size is not a power of two.
		(void)regions_cheap;
		if (!(flags & 32u) && !(flags & 16u) && ((h->dbg & 0x1000000) || keys >= (1ull << 20))) {
			// admission on a log whose table will be built range by range: the marks are kept, nothing is written into the log
			const bool defer = from_log && !(h->dbg & 0x800000) && (keys >= (1ull << 20) || (h->dbg & 0x400000));
			rc0 = bloom_xor(h, T, m_bits, keys, &resolved, defer ? &lost_kept : nullptr);
			if (rc0) return rc0;
			if (resolved) h->stats.bloom_partitioned = 2;
		}
		if (!resolved && h->pending_compact)
			if (int rcx = full_log(true)) return rcx;
		if (!resolved && !(flags & 32u) && ((flags & 16u) || (regions_fit && keys >= (1ull << 20)))) {
			rc0 = bloom_partitioned(h, T, m_bits, keys, &resolved);
			if (rc0) return rc0;
			h->stats.bloom_partitioned = resolved ? 1 : 0;
		}
	}
	if (m_bits > 0 && !resolved) {
		unsigned long long *bitstamp = nullptr;
		HIP_TRY(h, pool_alloc(h->cfg.device, (void **)&bitstamp, m_bits * sizeof(unsigned long long)));
		hipError_t e;
		{
			ProfScope ps(h, "bloom_stamp_fill");
			e = hipMemsetAsync(bitstamp, 0xFF, m_bits * sizeof(unsigned long long), h->stream);
		}
		if (e == hipSuccess) {
			{
				ProfScope ps(h, "k_bloom_touch");
				if (h->W == 1) hipLaunchKernelGGL(k_bloom_touch<1>, dim3(grid), dim3(BLOCK), 0, h->stream, T.slots, T.cap, m_bits, bitstamp, h->d_counters);
				else hipLaunchKernelGGL(k_bloom_touch<2>, dim3(grid), dim3(BLOCK), 0, h->stream, T.slots, T.cap, m_bits, bitstamp, h->d_counters);
			}
			{
				ProfScope ps(h, "k_admit");
				if (h->W == 1) hipLaunchKernelGGL((k_admit<1, true>), dim3(grid), dim3(BLOCK), 0, h->stream, T.slots, T.cap, m_bits, bitstamp, h->d_counters, h->d_hist_dense, h->d_hist_ovf);
				else hipLaunchKernelGGL((k_admit<2, true>), dim3(grid), dim3(BLOCK), 0, h->stream, T.slots, T.cap, m_bits, bitstamp, h->d_counters, h->d_hist_dense, h->d_hist_ovf);
			}
			e = hipGetLastError();
		}
		hipError_t e2 = rg_stream_sync(h);
		pool_free(h->cfg.device, bitstamp);  // the reference destroys its filter here too (MessageProcessor.cpp:1010)
		if (e != hipSuccess) return fail(h, RG_ERR_HIP, "Bloom resolution failed: %s", hipGetErrorString(e));
		if (e2 != hipSuccess) return fail(h, RG_ERR_HIP, "Bloom resolution failed: %s", hipGetErrorString(e2));
	} else if (m_bits == 0) {
		if (h->W == 1) hipLaunchKernelGGL((k_admit<1, false>), dim3(grid), dim3(BLOCK), 0, h->stream, T.slots, T.cap, 0ull, (const unsigned long long *)nullptr, h->d_counters, h->d_hist_dense, h->d_hist_ovf);
		else hipLaunchKernelGGL((k_admit<2, false>), dim3(grid), dim3(BLOCK), 0, h->stream, T.slots, T.cap, 0ull, (const unsigned long long *)nullptr, h->d_counters, h->d_hist_dense, h->d_hist_ovf);
		HIP_TRY(h, hipGetLastError());
	}
	// One pass instead of two (round 3): the first level of the table partition applies the admission rule, gathers the
	// statistics and the histogram and writes into ranges sized in advance -- possible when the stored vertices can be bounded
	// before it: entries seen twice or more (counted by whoever made the log) + false positives (counted by k_xor_fp).
	bool fused_built = false, fused_tried = false;
	h->hist_staged = false;
	auto drop_marks = [&]() {
		pool_free(h->cfg.device, lost_kept);
		lost_kept = nullptr;
	};
	if (lost_kept && from_log && h->pending_multi != ~0ull && !(h->dbg & 0x800000)) {
		const uint64_t bound = std::min<uint64_t>(h->pending_keys, h->pending_multi + h->h_counters[C_FP_COUNT]);
		if (bound >= (1ull << 20) || (h->dbg & 0x400000)) {
			bool presize_failed = false;
			fused_tried = true;
			int rcb = build_table_ranges(h, h->pending_log, h->pending_n, std::max<uint64_t>(bound, 1), &fused_built, lost_kept, nullptr, true, &presize_failed, h->pending_compact);
			if (rcb) {
				pool_free(h->cfg.device, lost_kept);
				return rcb;
			}
			(void)presize_failed;
			if (!fused_built) {
				// nothing was built (a range too small, a table of one level, nothing admitted, no memory): what the pass may have gathered is void, count first
				hipError_t ez = hipMemsetAsync(h->d_hist_dense, 0, HIST_DENSE * sizeof(unsigned long long), h->stream);
				if (ez == hipSuccess) ez = hipMemsetAsync(h->d_counters + C_RETAINED, 0, 2 * sizeof(unsigned long long), h->stream);  // (and C_INSERTIONS)
				if (ez == hipSuccess) ez = hipMemsetAsync(h->d_counters + C_OVF, 0, sizeof(unsigned long long), h->stream);
				if (ez != hipSuccess) {
					pool_free(h->cfg.device, lost_kept);
					return fail(h, RG_ERR_HIP, "admission: %s", hipGetErrorString(ez));
				}
				fused_tried = false;
				h->hist_staged = false;
			}
		}
	}
	if (h->pending_compact && !fused_built) {  // the admission is decided: the routes below read full entries, whose stamps no longer matter
		if (int rcx = full_log(false)) {
			drop_marks();
			return rcx;
		}
	}
	if (lost_kept && !fused_tried) {  // the counting pass of the admission: statistics, histogram, first level of the table partition; the log stays as it is
		hipError_t e = pool_alloc(h->cfg.device, (void **)&d_first_hist, 256 * sizeof(unsigned long long));
		if (e == hipSuccess) e = hipMemsetAsync(d_first_hist, 0, 256 * sizeof(unsigned long long), h->stream);
		if (e == hipSuccess) {
			ProfScope ps(h, "k_admit_count");
			if (h->W == 1) hipLaunchKernelGGL((k_admit_lost<1, false>), dim3(grid), dim3(BLOCK), 0, h->stream, T.slots, T.cap, (const uint32_t *)lost_kept, h->d_counters, h->d_hist_dense, h->d_hist_ovf, d_first_hist);
			else hipLaunchKernelGGL((k_admit_lost<2, false>), dim3(grid), dim3(BLOCK), 0, h->stream, T.slots, T.cap, (const uint32_t *)lost_kept, h->d_counters, h->d_hist_dense, h->d_hist_ovf, d_first_hist);
			e = hipGetLastError();
		}
		if (e != hipSuccess) {
			pool_free(h->cfg.device, lost_kept);
			pool_free(h->cfg.device, d_first_hist);
			return fail(h, RG_ERR_HIP, "admission: %s", hipGetErrorString(e));
		}
	}
	// (built by the one-pass admission: the histogram and the counters came with the build's last wait, and nothing has been
	// launched since -- none of the three waits below is needed)
	const bool all_read = fused_built && h->hist_staged;
	int rc = collect_histogram(h, h->d_hist_dense, h->d_hist_ovf, all_read);
	h->hist_staged = false;
	if (rc) {
		pool_free(h->cfg.device, lost_kept);
		pool_free(h->cfg.device, d_first_hist);
		return rc;
	}
	uint64_t candidates = h->h_counters[C_FILL];
	if (from_log) {
		candidates = h->pending_keys;
		const uint64_t retained = h->h_counters[C_RETAINED];
		// the table range by range, without global atomics (rg_debug_set bit 22 forces it on small inputs, bit 23 never: tests)
		bool built = fused_built;
		if (!built && !fused_tried && !(h->dbg & 0x800000) && (retained >= (1ull << 20) || (h->dbg & 0x400000)))
			rc = build_table_ranges(h, h->pending_log, h->pending_n, retained, &built, lost_kept, d_first_hist);
		if (rc == RG_OK && !built && lost_kept) {  // the classic insert wants the log admitted in place: write what the counting pass left out
			ProfScope ps(h, "k_admit");
			if (h->W == 1) hipLaunchKernelGGL((k_admit_lost<1, true, false>), dim3(grid), dim3(BLOCK), 0, h->stream, T.slots, T.cap, (const uint32_t *)lost_kept, h->d_counters, (unsigned long long *)nullptr, (uint32_t *)nullptr);
			else hipLaunchKernelGGL((k_admit_lost<2, true, false>), dim3(grid), dim3(BLOCK), 0, h->stream, T.slots, T.cap, (const uint32_t *)lost_kept, h->d_counters, (unsigned long long *)nullptr, (uint32_t *)nullptr);
			if (hipGetLastError() != hipSuccess) rc = fail(h, RG_ERR_HIP, "k_admit_lost launch failed");
		}
		if (lost_kept) {
			if (!all_read) (void)rg_stream_sync(h);
			pool_free(h->cfg.device, lost_kept);
			pool_free(h->cfg.device, d_first_hist);
			lost_kept = nullptr;
			d_first_hist = nullptr;
		}
		if (rc == RG_OK && !built) rc = table_reserve(h, h->tab, std::max<uint64_t>(1, retained), C_FILL);
		if (rc == RG_OK && !built) {
			ProfScope ps(h, "k_insert_admitted");
			if (h->W == 1) hipLaunchKernelGGL(k_insert_admitted<1>, dim3(grid_for(h->pending_n)), dim3(BLOCK), 0, h->stream, h->pending_log, h->pending_n, h->tab.slots, h->tab.cap, h->d_counters);
			else hipLaunchKernelGGL(k_insert_admitted<2>, dim3(grid_for(h->pending_n)), dim3(BLOCK), 0, h->stream, h->pending_log, h->pending_n, h->tab.slots, h->tab.cap, h->d_counters);
			if (hipGetLastError() != hipSuccess) rc = fail(h, RG_ERR_HIP, "k_insert_admitted launch failed");
		}
		if (rc == RG_OK && !all_read) rc = sync_counters(h);
		pool_free(h->cfg.device, h->pending_log);  // (a borrowed buffer is not the pool's: nothing happens)
		h->pending_log = nullptr;
		h->pending_compact = false;
		h->pending_multi = ~0ull;
		h->pending_borrowed = false;
		h->pending_n = 0;
		if (rc) return rc;
		if (h->h_counters[C_FILL] != retained) return fail(h, RG_ERR_STATE, "inserted %llu of %llu admitted vertices", (unsigned long long)h->h_counters[C_FILL], (unsigned long long)retained);
	}
	h->tab.fill_known = h->h_counters[C_FILL];
	h->tab.fill_pending = 0;
	h->stats.candidates = candidates;
	h->stats.stored_pairs = h->h_counters[C_RETAINED];
	h->stats.bloom_insertions = h->h_counters[C_INSERTIONS];
	h->stats.bloom_set_bits = h->h_counters[C_SETBITS];
	h->stats.table_slots = h->tab.cap;
	h->phase = PH_RESOLVED;
	return RG_OK;
}

int rg_count_kmers(rg_handle *h) {
	if (!h) return RG_ERR_ARG;
	if (h->cfg.size != 1) return fail(h, RG_ERR_STATE, "rg_count_kmers is the size==1 shortcut; use the decomposed calls with size > 1");
	int rc = rg_count_local(h);
	if (rc) return rc;
	return rg_resolve_admission(h, ~(uint64_t)0);
}

int rg_histogram(rg_handle *h, uint32_t *coverage, uint64_t *count, uint64_t cap, uint64_t *n) {
	if (!h || !n) return RG_ERR_ARG;
	if (h->phase < PH_RESOLVED) return fail(h, RG_ERR_STATE, "rg_histogram needs the k-mer count first");
	HIP_TRY(h, hipSetDevice(h->cfg.device));
	if (!h->hist_valid) {  // normally cached by rg_resolve_admission, which gathers it while writing the coverages
		unsigned long long *dense = nullptr;
		uint32_t *ovf = nullptr;
		HIP_TRY(h, pool_alloc(h->cfg.device, (void **)&dense, HIST_DENSE * sizeof(unsigned long long)));
		if (pool_alloc(h->cfg.device, (void **)&ovf, HIST_OVF_CAP * sizeof(uint32_t)) != hipSuccess) {
			pool_free(h->cfg.device, dense);
			return fail(h, RG_ERR_NOMEM, "rg_histogram: out of device memory");
		}
		hipError_t e = hipMemsetAsync(dense, 0, HIST_DENSE * sizeof(unsigned long long), h->stream);
		if (e == hipSuccess) e = hipMemsetAsync(h->d_counters + C_OVF, 0, sizeof(unsigned long long), h->stream);
		if (e == hipSuccess) {
			ProfScope ps(h, "k_histogram");
			hipLaunchKernelGGL(k_histogram, dim3(grid_for(h->tab.cap)), dim3(BLOCK), 0, h->stream, h->tab.slots, h->tab.cap, dense, ovf, h->d_counters);
			e = hipGetLastError();
		}
		int rc = (e == hipSuccess) ? collect_histogram(h, dense, ovf) : fail(h, RG_ERR_HIP, "rg_histogram: %s", hipGetErrorString(e));
		pool_free(h->cfg.device, dense);
		pool_free(h->cfg.device, ovf);
		if (rc) return rc;
	}
	uint64_t m = 0;
	for (const auto &cp : h->hist_cache) {
		if (m < cap && coverage && count) {
			coverage[m] = cp.first;
			count[m] = 2 * cp.second;  // GridTableIterator visits the key and its reverse complement
		}
		m++;
	}
	*n = m;
	return RG_OK;
}

int rg_add_edges(rg_handle *h) {
	if (!h) return RG_ERR_ARG;
	if (h->phase != PH_RESOLVED) return fail(h, RG_ERR_STATE, "rg_add_edges needs the k-mer count first (and runs once)");
	// The edge bits were gathered by the sweep and put in place by the admission kernel; what pass 2 of the
	// reference adds (VerticesExtractor.cpp:40-274) is already there, this call makes it visible to readers.
	h->phase = PH_EDGES;
	return RG_OK;
}

int rg_purge_edges(rg_handle *h) {
	if (!h) return RG_ERR_ARG;
	if (h->phase != PH_EDGES) return fail(h, RG_ERR_STATE, "rg_purge_edges needs rg_add_edges first (and runs once)");
	if (h->cfg.size != 1) return fail(h, RG_ERR_STATE, "rg_purge_edges supports size == 1 only in this version");
	HIP_TRY(h, hipSetDevice(h->cfg.device));
	if (h->W == 1) hipLaunchKernelGGL(k_purge<1>, dim3(grid_for(h->tab.cap)), dim3(BLOCK), 0, h->stream, h->tab.slots, h->tab.cap, h->cfg.k);
	else hipLaunchKernelGGL(k_purge<2>, dim3(grid_for(h->tab.cap)), dim3(BLOCK), 0, h->stream, h->tab.slots, h->tab.cap, h->cfg.k);
	HIP_TRY(h, hipGetLastError());
	HIP_TRY(h, rg_stream_sync(h));
	h->phase = PH_PURGED;
	return RG_OK;
}

int rg_purge_query_counts(rg_handle *h, uint64_t *counts) {
	if (!h || !counts) return RG_ERR_ARG;
	if (h->phase != PH_EDGES) return fail(h, RG_ERR_STATE, "rg_purge_query_counts needs rg_add_edges first");
	HIP_TRY(h, hipSetDevice(h->cfg.device));
	const uint32_t size = (uint32_t)h->cfg.size;
	HIP_TRY(h, hipMemsetAsync(h->d_owner, 0, sizeof(unsigned long long) * 2 * size, h->stream));
	{
		ProfScope ps(h, "k_purge_queries");
		if (h->W == 1) hipLaunchKernelGGL((k_purge_queries<1, false>), dim3(grid_for(h->tab.cap)), dim3(BLOCK), 0, h->stream, h->tab.slots, h->tab.cap, h->cfg.k, size, h->d_owner, (uint64_t *)nullptr, (uint64_t *)nullptr);
		else hipLaunchKernelGGL((k_purge_queries<2, false>), dim3(grid_for(h->tab.cap)), dim3(BLOCK), 0, h->stream, h->tab.slots, h->tab.cap, h->cfg.k, size, h->d_owner, (uint64_t *)nullptr, (uint64_t *)nullptr);
	}
	HIP_TRY(h, hipGetLastError());
	std::vector<unsigned long long> tmp(size);
	HIP_TRY(h, hipMemcpyAsync(tmp.data(), h->d_owner, sizeof(unsigned long long) * size, hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(h, rg_stream_sync(h));
	h->purge_counts.assign(tmp.begin(), tmp.end());
	h->purge_total = 0;
	for (uint32_t r = 0; r < size; ++r) {
		counts[r] = tmp[r];
		h->purge_total += tmp[r];
	}
	return RG_OK;
}

int rg_purge_export_queries(rg_handle *h, void *d_queries, uint64_t cap_queries) {
	if (!h || !d_queries) return RG_ERR_ARG;
	if (h->phase != PH_EDGES || h->purge_counts.size() != (size_t)h->cfg.size)
		return fail(h, RG_ERR_STATE, "rg_purge_export_queries needs rg_purge_query_counts first");
	if (cap_queries < h->purge_total) return fail(h, RG_ERR_CAPACITY, "query buffer holds %llu keys, %llu needed", (unsigned long long)cap_queries, (unsigned long long)h->purge_total);
	HIP_TRY(h, hipSetDevice(h->cfg.device));
	const uint32_t size = (uint32_t)h->cfg.size;
	if (h->d_purge_origin) {
		pool_free(h->cfg.device, h->d_purge_origin);
		h->d_purge_origin = nullptr;
	}
	if (pool_alloc(h->cfg.device, (void **)&h->d_purge_origin, sizeof(uint64_t) * std::max<uint64_t>(1, h->purge_total)) != hipSuccess) {
		(void)hipGetLastError();
		return fail(h, RG_ERR_NOMEM, "rg_purge_export_queries: out of device memory");
	}
	std::vector<unsigned long long> cursors(size);
	unsigned long long run = 0;
	for (uint32_t r = 0; r < size; ++r) {
		cursors[r] = run;
		run += h->purge_counts[r];
	}
	HIP_TRY(h, hipMemcpyAsync(h->d_owner + size, cursors.data(), sizeof(unsigned long long) * size, hipMemcpyHostToDevice, h->stream));
	{
		ProfScope ps(h, "k_purge_queries");
		if (h->W == 1) hipLaunchKernelGGL((k_purge_queries<1, true>), dim3(grid_for(h->tab.cap)), dim3(BLOCK), 0, h->stream, h->tab.slots, h->tab.cap, h->cfg.k, size, h->d_owner + size, (uint64_t *)d_queries, h->d_purge_origin);
		else hipLaunchKernelGGL((k_purge_queries<2, true>), dim3(grid_for(h->tab.cap)), dim3(BLOCK), 0, h->stream, h->tab.slots, h->tab.cap, h->cfg.k, size, h->d_owner + size, (uint64_t *)d_queries, h->d_purge_origin);
	}
	HIP_TRY(h, hipGetLastError());
	HIP_TRY(h, rg_stream_sync(h));
	return RG_OK;
}

int rg_purge_answer(rg_handle *h, const void *d_queries, uint64_t n, uint8_t *d_answers) {
	if (!h || (n && (!d_queries || !d_answers))) return RG_ERR_ARG;
	if (h->phase < PH_EDGES) return fail(h, RG_ERR_STATE, "rg_purge_answer needs rg_add_edges first");
	if (n == 0) return RG_OK;
	HIP_TRY(h, hipSetDevice(h->cfg.device));
	{
		ProfScope ps(h, "k_purge_answer");
		if (h->W == 1) hipLaunchKernelGGL(k_purge_answer<1>, dim3(grid_for(n)), dim3(BLOCK), 0, h->stream, h->tab.slots, h->tab.cap, (const uint64_t *)d_queries, n, d_answers);
		else hipLaunchKernelGGL(k_purge_answer<2>, dim3(grid_for(n)), dim3(BLOCK), 0, h->stream, h->tab.slots, h->tab.cap, (const uint64_t *)d_queries, n, d_answers);
	}
	HIP_TRY(h, hipGetLastError());
	HIP_TRY(h, rg_stream_sync(h));
	return RG_OK;
}

int rg_purge_apply(rg_handle *h, const uint8_t *d_answers, uint64_t n) {
	if (!h || (n && !d_answers)) return RG_ERR_ARG;
	if (h->phase != PH_EDGES || !h->d_purge_origin) return fail(h, RG_ERR_STATE, "rg_purge_apply needs rg_purge_export_queries first");
	if (n != h->purge_total) return fail(h, RG_ERR_ARG, "rg_purge_apply: %llu answers for %llu queries", (unsigned long long)n, (unsigned long long)h->purge_total);
	HIP_TRY(h, hipSetDevice(h->cfg.device));
	if (n) {
		ProfScope ps(h, "k_purge_apply");
		hipLaunchKernelGGL(k_purge_apply, dim3(grid_for(n)), dim3(BLOCK), 0, h->stream, h->tab.slots, h->d_purge_origin, d_answers, n);
		HIP_TRY(h, hipGetLastError());
	}
	HIP_TRY(h, rg_stream_sync(h));
	pool_free(h->cfg.device, h->d_purge_origin);
	h->d_purge_origin = nullptr;
	h->purge_counts.clear();
	h->phase = PH_PURGED;
	return RG_OK;
}

int rg_size(rg_handle *h, uint64_t *size) {
	if (!h || !size) return RG_ERR_ARG;
	if (h->phase < PH_RESOLVED) return fail(h, RG_ERR_STATE, "rg_size needs the k-mer count first");
	*size = 2 * h->stats.stored_pairs;
	return RG_OK;
}

// the stored vertices as Vertex::dump records in a device buffer (slot order, deterministic); *d_out is a pool block the caller frees
static int export_to_device(rg_handle *h, uint8_t **d_out_p) {
	*d_out_p = nullptr;
	const uint64_t n = h->stats.stored_pairs;
	const uint64_t rec = (uint64_t)h->W * 8 + 5;
	const uint64_t cap = h->tab.cap;
	uint64_t nchunks = std::min<uint64_t>(2048, (cap + BLOCK - 1) / BLOCK);
	uint64_t chunk = (cap + nchunks - 1) / nchunks;
	nchunks = (cap + chunk - 1) / chunk;
	PoolBlock g_chunk(h->cfg.device), g_out(h->cfg.device);
	HIP_TRY(h, g_chunk.alloc(nchunks * sizeof(unsigned long long)));
	HIP_TRY(h, g_out.alloc(n * rec));
	unsigned long long *d_chunk = g_chunk.as<unsigned long long>();
	uint8_t *d_out = g_out.as<uint8_t>();
	hipLaunchKernelGGL(k_export_count, dim3((int)nchunks), dim3(BLOCK), 0, h->stream, h->tab.slots, cap, chunk, d_chunk);
	HIP_TRY(h, hipGetLastError());
	std::vector<unsigned long long> hc(nchunks);
	HIP_TRY(h, hipMemcpyAsync(hc.data(), d_chunk, nchunks * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(h, rg_stream_sync(h));
	unsigned long long run = 0;
	for (auto &c : hc) {
		unsigned long long t = c;
		c = run;
		run += t;
	}
	if (run != n) return fail(h, RG_ERR_STATE, "export found %llu stored pairs, expected %llu", run, (unsigned long long)n);
	HIP_TRY(h, hipMemcpyAsync(d_chunk, hc.data(), nchunks * sizeof(unsigned long long), hipMemcpyHostToDevice, h->stream));
	if (h->W == 1) hipLaunchKernelGGL(k_export_write<1>, dim3((int)nchunks), dim3(BLOCK), 0, h->stream, h->tab.slots, cap, chunk, d_chunk, d_out, h->phase >= PH_EDGES ? 0xFFu : 0u);
	else hipLaunchKernelGGL(k_export_write<2>, dim3((int)nchunks), dim3(BLOCK), 0, h->stream, h->tab.slots, cap, chunk, d_chunk, d_out, h->phase >= PH_EDGES ? 0xFFu : 0u);
	HIP_TRY(h, hipGetLastError());
	HIP_TRY(h, rg_stream_sync(h));  // d_chunk goes back to the pool with this scope
	*d_out_p = (uint8_t *)g_out.release();
	return RG_OK;
}

int rg_export(rg_handle *h, void *host_buf, uint64_t cap_records, uint64_t *n_records) {
	if (!h || !n_records) return RG_ERR_ARG;
	if (h->phase < PH_RESOLVED) return fail(h, RG_ERR_STATE, "rg_export needs the k-mer count first");
	const uint64_t n = h->stats.stored_pairs;
	*n_records = n;
	if (!host_buf || cap_records == 0 || n == 0) return RG_OK;
	if (cap_records < n) return fail(h, RG_ERR_CAPACITY, "export buffer holds %llu records, need %llu", (unsigned long long)cap_records, (unsigned long long)n);
	HIP_TRY(h, hipSetDevice(h->cfg.device));
	const uint64_t rec = (uint64_t)h->W * 8 + 5;
	uint8_t *d_out = nullptr;
	if (int rc = export_to_device(h, &d_out)) return rc;
	hipError_t e = hipMemcpyAsync(host_buf, d_out, n * rec, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = rg_stream_sync(h);
	pool_free(h->cfg.device, d_out);
	if (e != hipSuccess) return fail(h, RG_ERR_HIP, "export failed: %s", hipGetErrorString(e));
	return RG_OK;
}

// The hand-back as a stream: the records reach the caller chunk by chunk in page-locked staging memory of the library's own
// (two buffers), and while the caller's function works through one chunk -- GridTable::insert + Vertex::load per record in
// Ray's glue -- the copy of the next one is already running.  Wall time = max(copy, consumer) instead of their sum, and the
// caller needs no buffer of the whole store (1.3 GB for C2's 1e8 vertices).
int rg_export_stream(rg_handle *h, uint64_t chunk_records, rg_export_fn consume, void *user) {
	if (!h || !consume) return RG_ERR_ARG;
	if (h->phase < PH_RESOLVED) return fail(h, RG_ERR_STATE, "rg_export_stream needs the k-mer count first");
	const uint64_t n = h->stats.stored_pairs;
	if (n == 0) return RG_OK;
	HIP_TRY(h, hipSetDevice(h->cfg.device));
	const uint64_t rec = (uint64_t)h->W * 8 + 5;
	if (chunk_records == 0) chunk_records = 1ull << 22;  // 4 M records: 55 / 88 MB per staging buffer
	chunk_records = std::min(chunk_records, n);
	if (h->export_stage_bytes < chunk_records * rec) {  // the staging buffers stay with the handle
		for (int i = 0; i < 2; ++i) {
			if (h->export_stage[i]) (void)hipHostFree(h->export_stage[i]);
			h->export_stage[i] = nullptr;
		}
		h->export_stage_bytes = 0;
		for (int i = 0; i < 2; ++i)
			if (hipHostMalloc(&h->export_stage[i], chunk_records * rec) != hipSuccess) {
				(void)hipGetLastError();
				return fail(h, RG_ERR_NOMEM, "rg_export_stream: no page-locked memory for two chunks of %llu records", (unsigned long long)chunk_records);
			}
		h->export_stage_bytes = chunk_records * rec;
	}
	uint8_t *d_out = nullptr;
	if (int rc = export_to_device(h, &d_out)) return rc;
	hipEvent_t done[2] = {nullptr, nullptr};
	hipError_t e = hipEventCreateWithFlags(&done[0], hipEventDisableTiming);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&done[1], hipEventDisableTiming);
	const uint64_t nchunks = (n + chunk_records - 1) / chunk_records;
	auto issue = [&](uint64_t c) {
		const uint64_t lo = c * chunk_records, cnt = std::min(chunk_records, n - lo);
		hipError_t x = hipMemcpyAsync(h->export_stage[c & 1], d_out + lo * rec, cnt * rec, hipMemcpyDeviceToHost, h->stream);
		if (x == hipSuccess) x = hipEventRecord(done[c & 1], h->stream);
		return x;
	};
	int stop = 0;
	if (e == hipSuccess) e = issue(0);
	for (uint64_t c = 0; c < nchunks && e == hipSuccess && !stop; ++c) {
		e = hipEventSynchronize(done[c & 1]);
		if (e == hipSuccess && c + 1 < nchunks) e = issue(c + 1);  // the other buffer: the consumer finished with it one round ago
		if (e == hipSuccess) stop = consume(user, h->export_stage[c & 1], std::min(chunk_records, n - c * chunk_records));
	}
	(void)rg_stream_sync(h);
	for (int i = 0; i < 2; ++i)
		if (done[i]) (void)hipEventDestroy(done[i]);
	pool_free(h->cfg.device, d_out);
	if (e != hipSuccess) return fail(h, RG_ERR_HIP, "rg_export_stream: %s", hipGetErrorString(e));
	if (stop) return fail(h, RG_ERR_STATE, "rg_export_stream: the consumer stopped the export (returned %d)", stop);
	return RG_OK;
}

int rg_find(rg_handle *h, const uint64_t *kmers, uint64_t n, uint32_t *coverage, uint8_t *edges) {
	if (!h) return RG_ERR_ARG;
	if (h->phase < PH_RESOLVED) return fail(h, RG_ERR_STATE, "rg_find needs the k-mer count first");
	if (n == 0) return RG_OK;
	if (!kmers || !coverage || !edges) return fail(h, RG_ERR_ARG, "null argument");
	HIP_TRY(h, hipSetDevice(h->cfg.device));
	PoolBlock gq(h->cfg.device), gc(h->cfg.device), ge(h->cfg.device);
	HIP_TRY(h, gq.alloc(n * h->W * sizeof(uint64_t)));
	HIP_TRY(h, gc.alloc(n * sizeof(uint32_t)));
	HIP_TRY(h, ge.alloc(n));
	uint64_t *dq = gq.as<uint64_t>();
	uint32_t *dc = gc.as<uint32_t>();
	uint8_t *de = ge.as<uint8_t>();
	hipError_t e = hipMemcpyAsync(dq, kmers, n * h->W * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream);
	if (e == hipSuccess) {
		if (h->W == 1) hipLaunchKernelGGL(k_find<1>, dim3(grid_for(n)), dim3(BLOCK), 0, h->stream, h->tab.slots, h->tab.cap, h->cfg.k, dq, n, dc, de, h->phase >= PH_EDGES ? 0xFFu : 0u);
		else hipLaunchKernelGGL(k_find<2>, dim3(grid_for(n)), dim3(BLOCK), 0, h->stream, h->tab.slots, h->tab.cap, h->cfg.k, dq, n, dc, de, h->phase >= PH_EDGES ? 0xFFu : 0u);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(coverage, dc, n * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(edges, de, n, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = rg_stream_sync(h);
	if (e != hipSuccess) return fail(h, RG_ERR_HIP, "find failed: %s", hipGetErrorString(e));
	return RG_OK;
}

int rg_add_reads_text(rg_handle *h, const char *text, uint64_t text_bytes, const uint64_t *starts, const uint32_t *raw_lengths,
                      uint64_t n_reads) {
	if (!h) return RG_ERR_ARG;
	if (h->phase != PH_CREATED) return fail(h, RG_ERR_STATE, "reads must be added before counting");
	if (n_reads == 0) return RG_OK;
	if (!text || !starts || !raw_lengths) return fail(h, RG_ERR_ARG, "null read arrays");
	for (uint64_t i = 0; i < n_reads; ++i) {
		// the reference does not trim (and cannot store, uint16_t m_length) a sequence of 65535 symbols or more
		if (raw_lengths[i] >= 65535u) return fail(h, RG_ERR_ARG, "sequence %llu has %u symbols (limit 65534)", (unsigned long long)i, raw_lengths[i]);
		if (starts[i] + raw_lengths[i] > text_bytes) return fail(h, RG_ERR_ARG, "sequence %llu outside the text buffer", (unsigned long long)i);
	}
	const int dev = h->cfg.device;
	HIP_TRY(h, hipSetDevice(dev));
	uint8_t *d_text = nullptr, *d_packed = nullptr;
	uint64_t *d_starts = nullptr, *d_tstart = nullptr;
	uint32_t *d_raw = nullptr, *d_tlen = nullptr;
	unsigned long long *d_nbytes = nullptr, *d_off = nullptr, *d_sums = nullptr;
	const uint32_t nblk = (uint32_t)((n_reads + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK);
	auto cleanup = [&](bool keep) {
		pool_free(dev, d_text);
		pool_free(dev, d_starts);
		pool_free(dev, d_tstart);
		pool_free(dev, d_raw);
		pool_free(dev, d_nbytes);
		pool_free(dev, d_sums);
		if (!keep) {
			pool_free(dev, d_packed);
			pool_free(dev, d_tlen);
			pool_free(dev, d_off);
		}
	};
	hipError_t e = hipSuccess;
	auto A = [&](void **p, size_t bytes) {
		if (e == hipSuccess) e = pool_alloc(dev, p, bytes);
	};
	A((void **)&d_text, text_bytes + 8);
	A((void **)&d_starts, n_reads * sizeof(uint64_t));
	A((void **)&d_tstart, n_reads * sizeof(uint64_t));
	A((void **)&d_raw, n_reads * sizeof(uint32_t));
	A((void **)&d_tlen, n_reads * sizeof(uint32_t));
	A((void **)&d_nbytes, n_reads * sizeof(unsigned long long));
	A((void **)&d_off, (n_reads + 1) * sizeof(unsigned long long));
	A((void **)&d_sums, ((size_t)nblk + 1) * sizeof(unsigned long long));
	if (e == hipSuccess) e = hipMemcpyAsync(d_text, text, text_bytes, hipMemcpyHostToDevice, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(d_starts, starts, n_reads * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(d_raw, raw_lengths, n_reads * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream);
	unsigned long long total = 0;
	if (e == hipSuccess) {
		ProfScope ps(h, "k_trim_reads");
		hipLaunchKernelGGL(k_trim_reads, dim3(grid_for(n_reads)), dim3(BLOCK), 0, h->stream, d_text, d_starts, d_raw, n_reads, d_tstart, d_tlen, d_nbytes, text_bytes);
		hipLaunchKernelGGL(k_scan_local, dim3(nblk), dim3(256), 0, h->stream, d_nbytes, d_off, n_reads, d_sums);
		hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(64), 0, h->stream, d_sums, nblk, d_sums + nblk);
		hipLaunchKernelGGL(k_scan_add, dim3(nblk), dim3(256), 0, h->stream, d_off, n_reads, d_sums, d_sums + nblk, (unsigned long long *)nullptr);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(&total, d_off + n_reads, sizeof total, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = rg_stream_sync(h);
	const uint64_t padded = ((total + 7) / 8) * 8 + 16;
	A((void **)&d_packed, padded);
	if (e == hipSuccess) e = hipMemsetAsync(d_packed + (padded - 24), 0, 24, h->stream);
	if (e == hipSuccess) {
		ProfScope ps(h, "k_pack_reads");
		hipLaunchKernelGGL(k_pack_reads, dim3((unsigned)((n_reads + 63) / 64)), dim3(BLOCK), 0, h->stream, d_text, d_tstart, d_tlen, d_off, n_reads, d_packed, ~0ull);
		e = hipGetLastError();
	}
	std::vector<uint32_t> tl(n_reads);
	if (e == hipSuccess) e = hipMemcpyAsync(tl.data(), d_tlen, n_reads * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = rg_stream_sync(h);
	if (e != hipSuccess) {
		(void)hipGetLastError();
		cleanup(false);
		return fail(h, e == hipErrorOutOfMemory ? RG_ERR_NOMEM : RG_ERR_HIP, "rg_add_reads_text: %s", hipGetErrorString(e));
	}
	uint32_t maxlen = 0;
	for (uint64_t i = 0; i < n_reads; ++i) maxlen = std::max(maxlen, tl[i]);
	cleanup(true);
	ReadBatch b;
	b.d_packed = (const uint64_t *)d_packed;
	b.d_off = (const uint64_t *)d_off;
	b.d_len = d_tlen;
	b.n = n_reads;
	b.first_id = h->cfg.first_read_id + h->n_reads;
	b.maxlen = maxlen;
	b.owned = true;
	h->reads.push_back(b);
	h->n_reads += n_reads;
	return RG_OK;
}

// The read ingest of rg_add_reads_text for callers that cannot allocate: same kernels, their per-read scratch
// in the file-scope array, results in the caller's buffers.  No hipMalloc / hipFree / pool / event creation here.
int rg_pack_reads_noalloc(rg_handle *h, const char *d_text, uint64_t text_bytes, const uint64_t *d_starts,
                          const uint32_t *d_raw_lengths, uint64_t n_reads, uint8_t *d_packed, uint64_t packed_cap,
                          uint64_t *d_byte_offsets, uint32_t *d_lengths) {
	if (!h) return RG_ERR_ARG;
	if (n_reads > NOALLOC_MAX_READS)
		return fail(h, RG_ERR_CAPACITY, "rg_pack_reads_noalloc: %llu sequences, the scratch holds %llu per call", (unsigned long long)n_reads,
		            (unsigned long long)NOALLOC_MAX_READS);
	if (!d_byte_offsets) return fail(h, RG_ERR_ARG, "null offsets array");
	if (n_reads && (!d_text || !d_starts || !d_raw_lengths || !d_lengths || (!d_packed && packed_cap))) return fail(h, RG_ERR_ARG, "null read arrays");
	const int dev = h->cfg.device;
	HIP_TRY(h, hipSetDevice(dev));
	HIP_TRY(h, noalloc_init(dev));
	// One scratch array per GPU: a call made while another thread is inside this function, or while the work an earlier
	// call queued on ANOTHER stream has not finished, is refused instead of racing (same stream: stream order serialises).
	NoallocGuard guard(dev);
	if (!guard.ok) return fail(h, RG_ERR_STATE, "rg_pack_reads_noalloc: another call is using this GPU's scratch");
	if (g_noalloc_last_stream[dev] && g_noalloc_last_stream[dev] != h->stream && hipEventQuery(g_noalloc_done[dev]) == hipErrorNotReady)
		return fail(h, RG_ERR_STATE, "rg_pack_reads_noalloc: the previous call's work (another stream) still uses this GPU's scratch");
	(void)hipGetLastError();
	unsigned long long *base = (unsigned long long *)g_noalloc_ptr[dev];
	uint64_t *s_tstart = (uint64_t *)base;
	unsigned long long *s_nbytes = base + NOALLOC_MAX_READS;
	unsigned long long *s_sums = base + 2 * NOALLOC_MAX_READS;
	unsigned long long *off = (unsigned long long *)d_byte_offsets;
	if (n_reads == 0) {  // d_byte_offsets[0] = 0 packed bytes
		HIP_TRY(h, hipMemsetAsync(off, 0, sizeof(unsigned long long), h->stream));
		return RG_OK;
	}
	const uint32_t nblk = (uint32_t)((n_reads + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK);
	hipLaunchKernelGGL(k_trim_reads, dim3(grid_for(n_reads)), dim3(BLOCK), 0, h->stream, (const uint8_t *)d_text, d_starts, d_raw_lengths, n_reads, s_tstart,
	                   d_lengths, s_nbytes, text_bytes);
	hipLaunchKernelGGL(k_scan_local, dim3(nblk), dim3(256), 0, h->stream, s_nbytes, off, n_reads, s_sums);
	hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(64), 0, h->stream, s_sums, nblk, s_sums + nblk);
	hipLaunchKernelGGL(k_scan_add, dim3(nblk), dim3(256), 0, h->stream, off, n_reads, s_sums, s_sums + nblk, (unsigned long long *)nullptr);
	hipLaunchKernelGGL(k_pack_reads, dim3((unsigned)((n_reads + 63) / 64)), dim3(BLOCK), 0, h->stream, (const uint8_t *)d_text, s_tstart, d_lengths, off, n_reads,
	                   d_packed, packed_cap);
	HIP_TRY(h, hipGetLastError());
	HIP_TRY(h, hipEventRecord(g_noalloc_done[dev], h->stream));  // the event was created at set-up: nothing is allocated here
	g_noalloc_last_stream[dev] = h->stream;
	return RG_OK;
}

int rg_get_reads(rg_handle *h, uint8_t *packed, uint64_t packed_cap, uint64_t *byte_offsets, uint32_t *lengths, uint64_t cap_reads,
                 uint64_t *n_reads, uint64_t *packed_bytes) {
	if (!h || !n_reads || !packed_bytes) return RG_ERR_ARG;
	HIP_TRY(h, hipSetDevice(h->cfg.device));
	uint64_t nr = 0, nb = 0;
	for (const ReadBatch &b : h->reads) {
		if (b.n == 0) continue;
		std::vector<uint64_t> off(b.n);
		std::vector<uint32_t> len(b.n);
		HIP_TRY(h, hipMemcpy(off.data(), b.d_off, b.n * sizeof(uint64_t), hipMemcpyDeviceToHost));
		HIP_TRY(h, hipMemcpy(len.data(), b.d_len, b.n * sizeof(uint32_t), hipMemcpyDeviceToHost));
		uint64_t end = 0;
		for (uint64_t i = 0; i < b.n; ++i) end = std::max<uint64_t>(end, off[i] + (len[i] + 3) / 4);
		if (packed && byte_offsets && lengths && nr + b.n <= cap_reads && nb + end <= packed_cap) {
			if (end) HIP_TRY(h, hipMemcpy(packed + nb, b.d_packed, end, hipMemcpyDeviceToHost));
			for (uint64_t i = 0; i < b.n; ++i) {
				byte_offsets[nr + i] = nb + off[i];
				lengths[nr + i] = len[i];
			}
		}
		nr += b.n;
		nb += end;
	}
	*n_reads = nr;
	*packed_bytes = nb;
	return RG_OK;
}

int rg_read_markers(rg_handle *h, int32_t *d_forward, int32_t *d_reverse) {
	if (!h || ((!d_forward || !d_reverse) && h->n_reads)) return RG_ERR_ARG;
	if (h->phase < PH_RESOLVED) return fail(h, RG_ERR_STATE, "rg_read_markers needs the k-mer count first");
	if (h->cfg.size != 1) return fail(h, RG_ERR_STATE, "rg_read_markers supports size == 1 only in this version");
	HIP_TRY(h, hipSetDevice(h->cfg.device));
	uint64_t done = 0;
	for (const ReadBatch &b : h->reads) {
		if (b.n == 0) continue;
		ProfScope ps(h, "k_read_markers");
		const unsigned grid = (unsigned)std::min<uint64_t>((b.n + 3) / 4, 65536);
		if (h->W == 1) hipLaunchKernelGGL(k_read_markers<1>, dim3(grid), dim3(256), 0, h->stream, b.d_packed, b.d_off, b.d_len, b.n, h->cfg.k, h->tab.slots, h->tab.cap, d_forward + done, d_reverse + done);
		else hipLaunchKernelGGL(k_read_markers<2>, dim3(grid), dim3(256), 0, h->stream, b.d_packed, b.d_off, b.d_len, b.n, h->cfg.k, h->tab.slots, h->tab.cap, d_forward + done, d_reverse + done);
		HIP_TRY(h, hipGetLastError());
		done += b.n;
	}
	HIP_TRY(h, rg_stream_sync(h));
	return RG_OK;
}

int rg_hash_kmers(rg_handle *h, const uint64_t *kmers, uint64_t n, uint64_t *hash1_out, uint64_t *hash2_out) {
	if (!h) return RG_ERR_ARG;
	if (n == 0) return RG_OK;
	if (!kmers || !hash1_out || !hash2_out) return fail(h, RG_ERR_ARG, "null argument");
	if (n > (1ull << 24)) return fail(h, RG_ERR_CAPACITY, "rg_hash_kmers takes at most 2^24 k-mers per call");
	HIP_TRY(h, hipSetDevice(h->cfg.device));
	PoolBlock gk(h->cfg.device), g1(h->cfg.device), g2(h->cfg.device);
	HIP_TRY(h, gk.alloc(n * h->W * sizeof(uint64_t)));
	HIP_TRY(h, g1.alloc(n * sizeof(uint64_t)));
	HIP_TRY(h, g2.alloc(n * sizeof(uint64_t)));
	HIP_TRY(h, hipMemcpyAsync(gk.p, kmers, n * h->W * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
	const unsigned grid = (unsigned)((n + BLOCK - 1) / BLOCK);
	if (h->W == 1) hipLaunchKernelGGL(k_hash_kmers<1>, dim3(grid), dim3(BLOCK), 0, h->stream, gk.as<uint64_t>(), n, g1.as<uint64_t>(), g2.as<uint64_t>());
	else hipLaunchKernelGGL(k_hash_kmers<2>, dim3(grid), dim3(BLOCK), 0, h->stream, gk.as<uint64_t>(), n, g1.as<uint64_t>(), g2.as<uint64_t>());
	HIP_TRY(h, hipGetLastError());
	HIP_TRY(h, hipMemcpyAsync(hash1_out, g1.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(h, hipMemcpyAsync(hash2_out, g2.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(h, rg_stream_sync(h));
	return RG_OK;
}

int rg_check_hashes(rg_handle *h, rg_hash_fn hash1_fn, rg_hash_fn hash2_fn, void *user, uint32_t n_samples) {
	if (!h) return RG_ERR_ARG;
	if (!hash1_fn || !hash2_fn) return fail(h, RG_ERR_ARG, "rg_check_hashes needs Ray's two hash functions");
	const uint64_t n = std::min<uint64_t>(std::max<uint32_t>(n_samples, 16), 1u << 20);
	const int W = h->W, k = h->cfg.k;
	// sample k-mers: the corners (all A, all T, alternating) and a splitmix64 stream, cut to 2k bits
	const uint64_t mask0 = (2 * k >= 64) ? ~0ull : ((1ull << (2 * k)) - 1);
	const uint64_t mask1 = (k > 32) ? ((2 * k - 64 >= 64) ? ~0ull : ((1ull << (2 * k - 64)) - 1)) : 0ull;
	std::vector<uint64_t> keys(n * W), d1(n), d2(n);
	uint64_t x = 0x243F6A8885A308D3ull + (uint64_t)k;
	auto next = [&x]() {
		uint64_t z = (x += 0x9E3779B97F4A7C15ull);
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		return z ^ (z >> 31);
	};
	for (uint64_t i = 0; i < n; ++i) {
		uint64_t a = next(), b = next();
		if (i == 0) a = b = 0;
		if (i == 1) a = b = ~0ull;
		if (i == 2) a = b = 0x24e4e4e4e4e4e4e4ull;
		if (i == 3) a = b = 1;
		keys[i * W] = a & mask0;
		if (W == 2) keys[i * W + 1] = b & mask1;
	}
	int rc = rg_hash_kmers(h, keys.data(), n, d1.data(), d2.data());
	if (rc != RG_OK) return rc;
	for (uint64_t i = 0; i < n; ++i) {
		const uint64_t r1 = hash1_fn(&keys[i * W], user), r2 = hash2_fn(&keys[i * W], user);
		if (r1 != d1[i] || r2 != d2[i])
			return fail(h, RG_ERR_HASH,
			            "hash mismatch on k-mer words %016llx %016llx (k=%d): caller's hash_function_%d gives %016llx, the GPU path %016llx -- "
			            "owners (vertexRank) and Bloom bits would differ from Ray's",
			            (unsigned long long)keys[i * W], (unsigned long long)(W == 2 ? keys[i * W + 1] : 0), k, r1 != d1[i] ? 1 : 2,
			            (unsigned long long)(r1 != d1[i] ? r1 : r2), (unsigned long long)(r1 != d1[i] ? d1[i] : d2[i]));
	}
	return RG_OK;
}

// reads [b.n * part / nparts, b.n * (part + 1) / nparts) of a batch: the slice of rg_records_local_part
static inline void batch_slice(const ReadBatch &b, uint32_t part, uint32_t nparts, uint64_t *lo, uint64_t *hi) {
	*lo = (uint64_t)((unsigned __int128)b.n * part / nparts);
	*hi = (uint64_t)((unsigned __int128)b.n * (part + 1) / nparts);
}

static void markers_free(rg_handle *h) {
	if (h->d_marker_qidx) pool_free(h->cfg.device, h->d_marker_qidx);
	if (h->d_marker_kprefix) pool_free(h->cfg.device, h->d_marker_kprefix);
	h->d_marker_qidx = nullptr;
	h->d_marker_kprefix = nullptr;
	h->marker_counts.clear();
}

int rg_markers_query_counts_part(rg_handle *h, uint32_t part, uint32_t nparts, uint64_t *counts) {
	if (!h || !counts) return RG_ERR_ARG;
	if (nparts == 0 || part >= nparts) return fail(h, RG_ERR_ARG, "part %u of %u", part, nparts);
	if (h->phase < PH_RESOLVED) return fail(h, RG_ERR_STATE, "rg_markers_query_counts needs the k-mer count first");
	const int dev = h->cfg.device;
	HIP_TRY(h, hipSetDevice(dev));
	const uint32_t size = (uint32_t)h->cfg.size;
	markers_free(h);
	h->marker_part = part;
	h->marker_nparts = nparts;
	// k-mers per read of the slice -> exclusive scan: k-mer p of slice read r lives at kprefix[r] + p
	uint64_t n_slice = 0;
	for (const ReadBatch &b : h->reads) {
		uint64_t lo, hi;
		batch_slice(b, part, nparts, &lo, &hi);
		n_slice += hi - lo;
	}
	h->marker_slice_reads = n_slice;
	const uint32_t nblk = (uint32_t)((std::max<uint64_t>(1, n_slice) + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK);
	PoolBlock nk(dev), sums(dev);
	HIP_TRY(h, nk.alloc(std::max<uint64_t>(1, n_slice) * sizeof(unsigned long long)));
	HIP_TRY(h, sums.alloc(((size_t)nblk + 1) * sizeof(unsigned long long)));
	if (pool_alloc(dev, (void **)&h->d_marker_kprefix, (n_slice + 1) * sizeof(unsigned long long)) != hipSuccess) {
		(void)hipGetLastError();
		return fail(h, RG_ERR_NOMEM, "rg_markers_query_counts: out of device memory");
	}
	HIP_TRY(h, hipMemsetAsync(h->d_owner, 0, sizeof(unsigned long long) * 2 * size, h->stream));
	uint64_t done = 0;
	for (const ReadBatch &b : h->reads) {
		uint64_t lo, hi;
		batch_slice(b, part, nparts, &lo, &hi);
		if (hi == lo) continue;
		ProfScope ps(h, "k_marker_queries");
		const uint64_t n = hi - lo;
		hipLaunchKernelGGL(k_marker_kmer_counts, dim3(grid_for(n)), dim3(BLOCK), 0, h->stream, b.d_len + lo, n, h->cfg.k, nk.as<unsigned long long>() + done);
		const unsigned grid = (unsigned)std::min<uint64_t>((n + 3) / 4, 65536);
		if (h->W == 1) hipLaunchKernelGGL((k_marker_queries<1, false>), dim3(grid), dim3(256), size * 12, h->stream, b.d_packed, b.d_off + lo, b.d_len + lo, n, h->cfg.k, size, h->d_owner, (uint64_t *)nullptr, (const unsigned long long *)nullptr, (uint64_t *)nullptr);
		else hipLaunchKernelGGL((k_marker_queries<2, false>), dim3(grid), dim3(256), size * 12, h->stream, b.d_packed, b.d_off + lo, b.d_len + lo, n, h->cfg.k, size, h->d_owner, (uint64_t *)nullptr, (const unsigned long long *)nullptr, (uint64_t *)nullptr);
		HIP_TRY(h, hipGetLastError());
		done += n;
	}
	if (n_slice) {
		hipLaunchKernelGGL(k_scan_local, dim3(nblk), dim3(256), 0, h->stream, nk.as<unsigned long long>(), h->d_marker_kprefix, n_slice, sums.as<unsigned long long>());
		hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(64), 0, h->stream, sums.as<unsigned long long>(), nblk, sums.as<unsigned long long>() + nblk);
		hipLaunchKernelGGL(k_scan_add, dim3(nblk), dim3(256), 0, h->stream, h->d_marker_kprefix, n_slice, sums.as<unsigned long long>(), sums.as<unsigned long long>() + nblk, (unsigned long long *)nullptr);
		HIP_TRY(h, hipGetLastError());
	} else {
		HIP_TRY(h, hipMemsetAsync(h->d_marker_kprefix, 0, sizeof(unsigned long long), h->stream));
	}
	std::vector<unsigned long long> tmp(size);
	HIP_TRY(h, hipMemcpyAsync(tmp.data(), h->d_owner, sizeof(unsigned long long) * size, hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(h, rg_stream_sync(h));
	h->marker_counts.assign(tmp.begin(), tmp.end());
	h->marker_total = 0;
	for (uint32_t r = 0; r < size; ++r) {
		counts[r] = tmp[r];
		h->marker_total += tmp[r];
	}
	return RG_OK;
}

int rg_markers_query_counts(rg_handle *h, uint64_t *counts) { return rg_markers_query_counts_part(h, 0, 1, counts); }

int rg_markers_export_queries(rg_handle *h, void *d_queries, uint64_t cap_queries) {
	if (!h || (!d_queries && cap_queries)) return RG_ERR_ARG;
	if (h->phase < PH_RESOLVED || h->marker_counts.size() != (size_t)h->cfg.size || !h->d_marker_kprefix) return fail(h, RG_ERR_STATE, "rg_markers_export_queries needs rg_markers_query_counts first");
	if (cap_queries < h->marker_total) return fail(h, RG_ERR_CAPACITY, "query buffer holds %llu keys, %llu needed", (unsigned long long)cap_queries, (unsigned long long)h->marker_total);
	HIP_TRY(h, hipSetDevice(h->cfg.device));
	const uint32_t size = (uint32_t)h->cfg.size;
	if (h->d_marker_qidx) {
		pool_free(h->cfg.device, h->d_marker_qidx);
		h->d_marker_qidx = nullptr;
	}
	if (pool_alloc(h->cfg.device, (void **)&h->d_marker_qidx, sizeof(uint64_t) * std::max<uint64_t>(1, h->marker_total)) != hipSuccess) {
		(void)hipGetLastError();
		return fail(h, RG_ERR_NOMEM, "rg_markers_export_queries: out of device memory");
	}
	std::vector<unsigned long long> cursors(size);
	unsigned long long run = 0;
	for (uint32_t r = 0; r < size; ++r) {
		cursors[r] = run;
		run += h->marker_counts[r];
	}
	HIP_TRY(h, hipMemcpyAsync(h->d_owner + size, cursors.data(), sizeof(unsigned long long) * size, hipMemcpyHostToDevice, h->stream));
	uint64_t done = 0;
	for (const ReadBatch &b : h->reads) {
		uint64_t lo, hi;
		batch_slice(b, h->marker_part, h->marker_nparts, &lo, &hi);
		if (hi == lo) continue;
		const uint64_t n = hi - lo;
		ProfScope ps(h, "k_marker_queries");
		const unsigned grid = (unsigned)std::min<uint64_t>((n + 3) / 4, 65536);
		if (h->W == 1) hipLaunchKernelGGL((k_marker_queries<1, true>), dim3(grid), dim3(256), size * 12, h->stream, b.d_packed, b.d_off + lo, b.d_len + lo, n, h->cfg.k, size, h->d_owner + size, (uint64_t *)d_queries, h->d_marker_kprefix + done, h->d_marker_qidx);
		else hipLaunchKernelGGL((k_marker_queries<2, true>), dim3(grid), dim3(256), size * 12, h->stream, b.d_packed, b.d_off + lo, b.d_len + lo, n, h->cfg.k, size, h->d_owner + size, (uint64_t *)d_queries, h->d_marker_kprefix + done, h->d_marker_qidx);
		HIP_TRY(h, hipGetLastError());
		done += n;
	}
	HIP_TRY(h, rg_stream_sync(h));
	return RG_OK;
}

int rg_markers_answer(rg_handle *h, const void *d_queries, uint64_t n, uint32_t *d_coverage) {
	if (!h || (n && (!d_queries || !d_coverage))) return RG_ERR_ARG;
	if (h->phase < PH_RESOLVED) return fail(h, RG_ERR_STATE, "rg_markers_answer needs the k-mer count first");
	if (n == 0) return RG_OK;
	HIP_TRY(h, hipSetDevice(h->cfg.device));
	{
		ProfScope ps(h, "k_marker_answer");
		if (h->W == 1) hipLaunchKernelGGL(k_marker_answer<1>, dim3(grid_for(n)), dim3(BLOCK), 0, h->stream, h->tab.slots, h->tab.cap, (const uint64_t *)d_queries, n, d_coverage);
		else hipLaunchKernelGGL(k_marker_answer<2>, dim3(grid_for(n)), dim3(BLOCK), 0, h->stream, h->tab.slots, h->tab.cap, (const uint64_t *)d_queries, n, d_coverage);
	}
	HIP_TRY(h, hipGetLastError());
	HIP_TRY(h, rg_stream_sync(h));
	return RG_OK;
}

int rg_markers_apply(rg_handle *h, const uint32_t *d_coverage, uint64_t n, int32_t *d_forward, int32_t *d_reverse) {
	if (!h || (n && !d_coverage) || ((!d_forward || !d_reverse) && h->n_reads)) return RG_ERR_ARG;
	if (!h->d_marker_qidx || !h->d_marker_kprefix) return fail(h, RG_ERR_STATE, "rg_markers_apply needs rg_markers_export_queries first");
	if (n != h->marker_total) return fail(h, RG_ERR_ARG, "rg_markers_apply: %llu answers for %llu queries", (unsigned long long)n, (unsigned long long)h->marker_total);
	const int dev = h->cfg.device;
	HIP_TRY(h, hipSetDevice(dev));
	hipError_t e = hipSuccess;
	uint64_t done = 0, before = 0;  // reads of the slice done; reads of the batches before this one
	for (const ReadBatch &b : h->reads) {
		uint64_t lo, hi;
		batch_slice(b, h->marker_part, h->marker_nparts, &lo, &hi);
		if (hi > lo && e == hipSuccess) {
			ProfScope ps(h, "k_markers_from_coverage");
			const uint64_t nr = hi - lo;
			const unsigned grid = (unsigned)std::min<uint64_t>((nr + 3) / 4, 65536);
			hipLaunchKernelGGL(k_markers_from_coverage, dim3(grid), dim3(256), 0, h->stream, b.d_len + lo, nr, h->cfg.k, d_coverage, h->d_marker_kprefix + done,
			                   h->d_marker_qidx, d_forward + before + lo, d_reverse + before + lo);
			e = hipGetLastError();
			done += nr;
		}
		before += b.n;
	}
	if (e == hipSuccess) e = rg_stream_sync(h);
	markers_free(h);
	if (e != hipSuccess) return fail(h, RG_ERR_HIP, "rg_markers_apply: %s", hipGetErrorString(e));
	return RG_OK;
}

int rg_import_vertices(rg_handle *h, const void *records, uint64_t n_pairs) {
	if (h) h->hist_valid = false;
	if (!h || (n_pairs && !records)) return RG_ERR_ARG;
	if (h->phase != PH_CREATED || h->n_reads) return fail(h, RG_ERR_STATE, "rg_import_vertices needs a fresh handle without reads");
	HIP_TRY(h, hipSetDevice(h->cfg.device));
	int rc = table_reserve(h, h->tab, std::max<uint64_t>(1, n_pairs), C_FILL);
	if (rc) return rc;
	const size_t bytes = (size_t)n_pairs * (h->W * 8 + 5);
	uint8_t *d = nullptr;
	if (n_pairs) {
		HIP_TRY(h, pool_alloc(h->cfg.device, (void **)&d, bytes));
		hipError_t e = hipMemcpyAsync(d, records, bytes, hipMemcpyHostToDevice, h->stream);
		if (e == hipSuccess) {
			if (h->W == 1) hipLaunchKernelGGL(k_import<1>, dim3(grid_for(n_pairs)), dim3(BLOCK), 0, h->stream, d, n_pairs, h->tab.slots, h->tab.cap, h->d_counters);
			else hipLaunchKernelGGL(k_import<2>, dim3(grid_for(n_pairs)), dim3(BLOCK), 0, h->stream, d, n_pairs, h->tab.slots, h->tab.cap, h->d_counters);
			e = hipGetLastError();
		}
		rc = (e == hipSuccess) ? sync_counters(h) : fail(h, RG_ERR_HIP, "rg_import_vertices: %s", hipGetErrorString(e));
		pool_free(h->cfg.device, d);
		if (rc) return rc;
	}
	h->tab.fill_known = h->h_counters[C_FILL];
	h->tab.fill_pending = 0;
	h->stats.candidates = h->stats.stored_pairs = h->h_counters[C_FILL];
	h->stats.table_slots = h->tab.cap;
	h->phase = PH_EDGES;  // a checkpoint is written after pass 2: coverage and edges are final
	return RG_OK;
}

int rg_degree_distribution(rg_handle *h, uint64_t out[25]) {
	if (!h || !out) return RG_ERR_ARG;
	if (h->phase < PH_RESOLVED) return fail(h, RG_ERR_STATE, "rg_degree_distribution needs the k-mer count first");
	HIP_TRY(h, hipSetDevice(h->cfg.device));
	unsigned long long *d = nullptr;
	HIP_TRY(h, pool_alloc(h->cfg.device, (void **)&d, 25 * sizeof(unsigned long long)));
	HIP_TRY(h, hipMemsetAsync(d, 0, 25 * sizeof(unsigned long long), h->stream));
	hipLaunchKernelGGL(k_degrees, dim3(grid_for(h->tab.cap)), dim3(BLOCK), 0, h->stream, h->tab.slots, h->tab.cap, d, h->phase >= PH_EDGES ? 0xFFu : 0u);
	unsigned long long tmp[25];
	hipError_t e = hipMemcpyAsync(tmp, d, sizeof tmp, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = rg_stream_sync(h);
	pool_free(h->cfg.device, d);
	if (e != hipSuccess) return fail(h, RG_ERR_HIP, "degree distribution failed: %s", hipGetErrorString(e));
	for (int i = 0; i < 25; ++i) out[i] = tmp[i];
	return RG_OK;
}

int rg_release_cached_memory(void) {
	std::vector<void *> drop;
	{
		std::lock_guard<std::mutex> lk(g_pool.mu);
		for (int d = 0; d < 64; ++d) {
			for (auto &kv : g_pool.free_blocks[d]) drop.push_back(kv.second);
			g_pool.free_blocks[d].clear();
		}
	}
	for (void *p : drop) (void)hipFree(p);
	return RG_OK;
}

int rg_debug_set(rg_handle *h, uint32_t flags) {
	if (!h) return RG_ERR_ARG;
	if (h->phase != PH_CREATED) return fail(h, RG_ERR_STATE, "rg_debug_set must come before counting starts");
	h->dbg = flags;
	return RG_OK;
}

int rg_set_profiling(rg_handle *h, int enable) {
	if (!h) return RG_ERR_ARG;
	h->profiling = enable != 0;
	return RG_OK;
}

int rg_get_kernel_profile(rg_handle *h, char *buf, uint64_t cap) {
	if (!h || !buf || cap == 0) return RG_ERR_ARG;
	HIP_TRY(h, hipSetDevice(h->cfg.device));
	HIP_TRY(h, rg_stream_sync(h));
	std::string out;
	for (auto &kv : h->prof) {
		double ms = 0;
		for (auto &p : kv.second) {
			float t = 0;
			HIP_TRY(h, hipEventElapsedTime(&t, p.first, p.second));
			ms += t;
		}
		char line[160];
		snprintf(line, sizeof line, "%s %.6f %zu\n", kv.first.c_str(), ms, kv.second.size());
		out += line;
	}
	if (out.size() + 1 > cap) return fail(h, RG_ERR_CAPACITY, "profile needs %zu bytes", out.size() + 1);
	memcpy(buf, out.c_str(), out.size() + 1);
	return RG_OK;
}

int rg_get_stats(rg_handle *h, rg_stats *out) {
	if (!h || !out) return RG_ERR_ARG;
	*out = h->stats;
	out->reads = h->n_reads;
	return RG_OK;
}

}  // extern "C"
