// rg_bloom.h -- Bloom admission resolved without scattered global atomics.
//
// The dense method (raygraph.hip: k_bloom_touch / k_admit) keeps one 8-byte first-touch stamp per
// Bloom bit and pays 8 scattered device-scope atomics per distinct key (~17.5 G/s chip-wide).  At the
// fill the reference's sizing aims for (a few percent) almost every bit is touched by one key only, and
// then nothing has to be ordered at all.  So instead:
//
//   k_bloom_hist + k_bloom_scatter_items   8 items (bit index, table slot) per key, produced straight into the
//                   ranges of the first partition level (histogram first, so nothing is streamed out and read back)
//                   (dense key arrays: the log of the sweep; sparse ones -- a vertex table -- go through k_bloom_items,
//                   which streams the items out, and one more k_scatter64 level)
//   k_scatter64     the second partition level by Bloom region (2^17 bits per region), when there are more than 256 regions
//   k_bloom_region  one workgroup per region, all in LDS: two bitmaps (touched / touched again);
//                   only for bits touched more than once are the stamps fetched and the earliest
//                   found; every later toucher of such a bit gets one "lost" mark on its key
//   k_admit_lost    a key is a false positive iff all 8 of its bit checks lost (MessageProcessor.cpp:741-750)
//
// Same result as the dense method bit for bit; used when the expected fill is low, falls back to
// the dense method otherwise or when a region holds more contested bits than LDS can rank (4096, or
// 10240 in the variant picked for fuller filters).
#pragma once
#include "rg_device.h"

namespace rg {

static constexpr int REGION_BITS_LOG2 = 17;
static constexpr uint32_t REGION_BITS = 1u << REGION_BITS_LOG2;
static constexpr uint32_t REGION_WORDS = REGION_BITS / 32;  // 4096 words per bitmap
static constexpr uint32_t REGION_MULTI_CAP = 4096;          // contested bits a region can rank (32 KB of stamps): 2 workgroups per CU
static constexpr uint32_t REGION_MULTI_CAP_BIG = 10240;     // fuller filters: 80 KB of stamps, 1 workgroup per CU
static constexpr uint64_t ITEM_INVALID = ~0ull;
static constexpr uint32_t ITEM_CHUNK_KEYS = 512;  // keys (x8 items) a wavefront claims at a time (k_bloom_items)
static constexpr int SCATTER64_TILE = 4096;

enum BloomCounter { BC_ITEM_CURSOR = 24, BC_ITEM_OVF, BC_REGION_OVF };

// item = bit index (33 bits) << 31 | table slot (31 bits)
// Sparse key arrays (the vertex table of a rank that received its keys from others): the items are streamed out
// densely, chunk by chunk, and partitioned by k_scatter64 afterwards -- a tile of slots would hold too few keys for
// k_bloom_scatter_items to write runs.
template <int W>
__global__ __launch_bounds__(256) void k_bloom_items(const Slot *__restrict__ tab, uint64_t cap, uint64_t m_bits,
                                                     unsigned long long *__restrict__ items, uint64_t item_cap, int shift1,
                                                     unsigned long long *__restrict__ counters, unsigned long long *__restrict__ hist1) {
	__shared__ uint32_t s_hist[256];
	const uint32_t lane = threadIdx.x & 63u;
	s_hist[threadIdx.x] = 0;
	__syncthreads();
	uint64_t chunk_base = 0;
	uint32_t chunk_used = ITEM_CHUNK_KEYS;
	bool have_chunk = false, dead = false;
	const uint64_t nwaves = (uint64_t)gridDim.x * 4, wave_id = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	for (uint64_t base = wave_id * 64; base < cap; base += nwaves * 64) {
		const uint64_t i = base + lane;
		Slot v;
		bool used = false;
		if (i < cap) {
			v = tab[i];
			used = v.k0 != EMPTY;
		}
		const unsigned long long ballot = __ballot(used);
		const uint32_t c = (uint32_t)__popcll(ballot);
		if (c == 0) continue;
		// keys that do not fit the current chunk continue in a freshly claimed one (no holes in between)
		const uint32_t room = have_chunk ? ITEM_CHUNK_KEYS - chunk_used : 0u;
		uint64_t new_base = 0;
		if (c > room) {
			unsigned long long nb = 0;
			if (lane == 0) nb = atomicAdd(&counters[BC_ITEM_CURSOR], (unsigned long long)ITEM_CHUNK_KEYS * 8);
			nb = __shfl(nb, 0);
			if (nb + ITEM_CHUNK_KEYS * 8 > item_cap) {
				dead = true;
				if (lane == 0) atomicAdd(&counters[BC_ITEM_OVF], 1ull);
			}
			new_base = nb;
		}
		if (used && !dead) {
			const uint32_t idx = (uint32_t)__popcll(ballot & ((1ull << lane) - 1));
			uint64_t K0, K1;
			from_internal<W>(v.k0, v.k1, K0, K1);
			const uint64_t origin = hash2<W>(K0, K1);
			unsigned long long *dst = (idx < room) ? items + chunk_base + (uint64_t)(chunk_used + idx) * 8
			                                       : items + new_base + (uint64_t)(idx - room) * 8;
#pragma unroll
			for (int j = 0; j < 8; ++j) {
				const uint64_t b = (origin ^ BLOOM_MAGIC[j]) % m_bits;
				const unsigned long long item = (b << 31) | i;
				dst[j] = item;
				atomicAdd(&s_hist[(item >> shift1) & 255u], 1u);
			}
		}
		if (c > room) {
			chunk_base = new_base;
			chunk_used = c - room;
			have_chunk = true;
		} else {
			chunk_used += c;
		}
	}
	if (have_chunk && !dead)
		for (uint32_t j = chunk_used * 8 + lane; j < ITEM_CHUNK_KEYS * 8; j += 64) items[chunk_base + j] = ITEM_INVALID;
	__syncthreads();
	if (s_hist[threadIdx.x]) atomicAdd(&hist1[threadIdx.x], (unsigned long long)s_hist[threadIdx.x]);
}

// x mod d with d fixed for the launch: inv = floor(2^64 / d) (host), or 0 when d is a power of two.  The estimate
// q = floor(x * inv / 2^64) is floor(x / d) or one less, so one conditional subtraction settles it -- a 64-bit
// division costs ~200 instructions per Bloom bit otherwise.
__device__ __forceinline__ uint64_t bloom_mod(uint64_t x, uint64_t d, uint64_t inv) {
	if (inv == 0) return x & (d - 1);
	const uint64_t r = x - __umul64hi(x, inv) * d;
	return r >= d ? r - d : r;
}

// Level-0 digit histogram of the items of all keys, nothing written: with it the items can be produced straight
// into their level-0 ranges (k_bloom_scatter_items) instead of being streamed out and partitioned afterwards.
template <int W>
__global__ __launch_bounds__(256) void k_bloom_hist(const Slot *__restrict__ tab, uint64_t cap, uint64_t m_bits, uint64_t m_inv, int shift1,
                                                    unsigned long long *__restrict__ hist1) {
	__shared__ uint32_t s_hist[256];
	s_hist[threadIdx.x] = 0;
	__syncthreads();
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * 256) {
		const uint64_t k0 = tab[i].k0;
		if (k0 == EMPTY) continue;
		const uint64_t k1 = (W == 2) ? tab[i].k1 : EMPTY;
		uint64_t K0, K1;
		from_internal<W>(k0, k1, K0, K1);
		const uint64_t origin = hash2<W>(K0, K1);
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			const unsigned long long item = (bloom_mod(origin ^ BLOOM_MAGIC[j], m_bits, m_inv) << 31) | i;
			atomicAdd(&s_hist[(item >> shift1) & 255u], 1u);
		}
	}
	__syncthreads();
	if (s_hist[threadIdx.x]) atomicAdd(&hist1[threadIdx.x], (unsigned long long)s_hist[threadIdx.x]);
}

// The items of a tile of slots are made in registers, sorted by their level-0 digit in LDS and written as
// runs at the tile's cursor claims: k_scatter64's scheme with the items computed instead of loaded.
// 1024 keys = 8192 items per workgroup (64 KB of items + digits in LDS, two workgroups per CU): runs of 32 items per
// (tile, digit); with 512 keys k_bloom_scatter_items took 4.9-5.0 ms on C2, with 1024 2.8-3.7 (same box)
static constexpr int ITEM_TILE_SLOTS = 1024;
template <int W>
__global__ __launch_bounds__(SCATTER_THREADS) void k_bloom_scatter_items(const Slot *__restrict__ tab, uint64_t cap, uint64_t m_bits, uint64_t m_inv,
                                                             int shift1, unsigned long long *__restrict__ cursors,
                                                             unsigned long long *__restrict__ out, uint8_t *__restrict__ side, int shift2) {
	// side != null: side[i] = the second level's digit of out[i] (k_level_hist_side reads it instead of the items)
	constexpr int KEYS = ITEM_TILE_SLOTS / SCATTER_THREADS, PER = 8 * KEYS;
	__shared__ unsigned long long s_item[8 * ITEM_TILE_SLOTS];
	__shared__ uint8_t s_dig[8 * ITEM_TILE_SLOTS];
	__shared__ uint32_t s_cnt[256], s_off[256], s_scan[8];
	__shared__ unsigned long long s_gbase[256];
	const uint64_t base = (uint64_t)blockIdx.x * ITEM_TILE_SLOTS;
	if (threadIdx.x < 256) s_cnt[threadIdx.x] = 0;
	__syncthreads();
	unsigned long long mine[PER];
	uint32_t dr[PER];  // digit << 16 | rank inside the tile
#pragma unroll
	for (int q = 0; q < KEYS; ++q) {
		const uint64_t i = base + threadIdx.x + (uint64_t)q * SCATTER_THREADS;
		uint64_t k0 = EMPTY, k1 = EMPTY;
		if (i < cap) {
			k0 = tab[i].k0;
			if (W == 2) k1 = tab[i].k1;
		}
#pragma unroll
		for (int j = 0; j < 8; ++j) dr[q * 8 + j] = 0xFFFFFFFFu;
		if (k0 != EMPTY) {
			uint64_t K0, K1;
			from_internal<W>(k0, k1, K0, K1);
			const uint64_t origin = hash2<W>(K0, K1);
#pragma unroll
			for (int j = 0; j < 8; ++j) {
				const unsigned long long item = (bloom_mod(origin ^ BLOOM_MAGIC[j], m_bits, m_inv) << 31) | i;
				const uint32_t d = (uint32_t)(item >> shift1) & 255u;
				mine[q * 8 + j] = item;
				dr[q * 8 + j] = (d << 16) | atomicAdd(&s_cnt[d], 1u);
			}
		}
	}
	__syncthreads();
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
			if (c) g = atomicAdd(&cursors[threadIdx.x], (unsigned long long)c);
			s_gbase[threadIdx.x] = g;
		}
	}
	__syncthreads();
	const uint32_t total = s_scan[4];
#pragma unroll
	for (int j = 0; j < PER; ++j)
		if (dr[j] != 0xFFFFFFFFu) {
			const uint32_t pos = s_off[dr[j] >> 16] + (dr[j] & 0xFFFFu);
			s_item[pos] = mine[j];
			s_dig[pos] = (uint8_t)(dr[j] >> 16);
		}
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < total; i += SCATTER_THREADS) {
		const uint32_t d = s_dig[i];
		const unsigned long long item = s_item[i], dst = s_gbase[d] + (i - s_off[d]);
		out[dst] = item;
		if (side) side[dst] = (uint8_t)((item >> shift2) & 255u);
	}
}

__device__ __forceinline__ uint32_t find_seg64(const uint32_t *__restrict__ tile_first, uint32_t S, uint32_t tile) {
	uint32_t lo = 0, hi = S;
	while (hi - lo > 1) {
		uint32_t mid = (lo + hi) >> 1;
		if (tile_first[mid] <= tile) lo = mid;
		else hi = mid;
	}
	return lo;
}

__global__ __launch_bounds__(256) void k_hist64(const unsigned long long *__restrict__ in, uint32_t S,
                                                const unsigned long long *__restrict__ seg_begin,
                                                const uint32_t *__restrict__ tile_first, int shift,
                                                unsigned long long *__restrict__ hist) {
	__shared__ uint32_t s_cnt[256];
	const uint32_t seg = find_seg64(tile_first, S, blockIdx.x);
	const uint64_t begin = seg_begin[seg] + (uint64_t)(blockIdx.x - tile_first[seg]) * SCATTER64_TILE;
	const uint64_t end = begin + SCATTER64_TILE < seg_begin[seg + 1] ? begin + SCATTER64_TILE : seg_begin[seg + 1];
	s_cnt[threadIdx.x] = 0;
	__syncthreads();
	for (uint64_t i = begin + threadIdx.x; i < end; i += 256) {
		unsigned long long it = in[i];
		if (it != ITEM_INVALID) atomicAdd(&s_cnt[(it >> shift) & 255u], 1u);
	}
	__syncthreads();
	if (s_cnt[threadIdx.x]) atomicAdd(&hist[((uint64_t)seg << 8) + threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
}

// one partition level of 8-byte items: same scheme as k_scatter (tile-local counting sort in LDS)
__global__ __launch_bounds__(SCATTER_THREADS) void k_scatter64(const unsigned long long *__restrict__ in, uint32_t S,
                                                   const unsigned long long *__restrict__ seg_begin,
                                                   const uint32_t *__restrict__ tile_first, int shift,
                                                   unsigned long long *__restrict__ cursors, unsigned long long *__restrict__ out) {
	constexpr int PER = SCATTER64_TILE / SCATTER_THREADS;
	__shared__ unsigned long long s_item[SCATTER64_TILE];
	__shared__ uint8_t s_dig[SCATTER64_TILE];
	__shared__ uint32_t s_cnt[256], s_off[256], s_scan[256];
	__shared__ unsigned long long s_gbase[256];
	const uint32_t seg = find_seg64(tile_first, S, blockIdx.x);
	const uint64_t begin = seg_begin[seg] + (uint64_t)(blockIdx.x - tile_first[seg]) * SCATTER64_TILE;
	const uint64_t end = begin + SCATTER64_TILE < seg_begin[seg + 1] ? begin + SCATTER64_TILE : seg_begin[seg + 1];
	if (threadIdx.x < 256) s_cnt[threadIdx.x] = 0;
	__syncthreads();
	unsigned long long mine[PER];
	uint32_t dig[PER], rank[PER];
#pragma unroll
	for (int j = 0; j < PER; ++j) {
		const uint64_t i = begin + threadIdx.x + (uint64_t)j * SCATTER_THREADS;
		dig[j] = 0xFFFFFFFFu;
		rank[j] = 0;
		mine[j] = ITEM_INVALID;
		if (i < end) {
			mine[j] = in[i];
			if (mine[j] != ITEM_INVALID) {
				dig[j] = (uint32_t)(mine[j] >> shift) & 255u;
				rank[j] = atomicAdd(&s_cnt[dig[j]], 1u);
			}
		}
	}
	__syncthreads();
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
			if (c) g = atomicAdd(&cursors[((uint64_t)seg << 8) + threadIdx.x], (unsigned long long)c);
			s_gbase[threadIdx.x] = g;
		}
	}
	__syncthreads();
	const uint32_t total = s_scan[4];
#pragma unroll
	for (int j = 0; j < PER; ++j)
		if (dig[j] != 0xFFFFFFFFu) {
			const uint32_t pos = s_off[dig[j]] + rank[j];
			s_item[pos] = mine[j];
			s_dig[pos] = (uint8_t)dig[j];
		}
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < total; i += SCATTER_THREADS) {
		const uint32_t d = s_dig[i];
		out[s_gbase[d] + (i - s_off[d])] = s_item[i];
	}
}

// One workgroup per Bloom region at a time.  lost[slot >> 2] holds four 8-bit counters.
// The items of a region are loaded once, REGION_PER per thread with independent loads (the loop over
// them is what the kernel waits for), and stay in registers for the three passes; a region with more
// than 512*REGION_PER items takes further rounds that read again.
static constexpr int REGION_PER = 32;
template <int W, uint32_t CAP>
__global__ __launch_bounds__(512, (CAP <= REGION_MULTI_CAP ? 4 : 2)) void k_bloom_region(const unsigned long long *__restrict__ items,
                                                      const unsigned long long *__restrict__ region_begin, uint32_t nregions,
                                                      const Slot *__restrict__ tab, uint32_t *__restrict__ lost,
                                                      unsigned long long *__restrict__ counters, int c_setbits,
                                                      unsigned long long *__restrict__ stamp_buf) {
	// stamp_buf: as long as items; regions of several rounds keep the stamps they gathered for the contested items
	// there (written and read back in item order) instead of gathering them a second time
	__shared__ uint32_t s_b1[REGION_WORDS];       // bit touched
	__shared__ uint32_t s_b2[REGION_WORDS];       // bit touched more than once
	__shared__ uint16_t s_pre[REGION_WORDS];      // exclusive prefix of popc(s_b2): rank of a contested bit
	__shared__ unsigned long long s_min[CAP];
	__shared__ uint32_t s_wsum[8];
	__shared__ uint32_t s_setbits;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	uint32_t setbits_total = 0;
	for (uint32_t r = blockIdx.x; r < nregions; r += gridDim.x) {
		const uint64_t begin = region_begin[r], end = region_begin[r + 1];
		if (begin == end) continue;  // uniform
		const bool one_round = end - begin <= 512ull * REGION_PER;
		unsigned long long mine[REGION_PER];
		if (one_round) {
#pragma unroll
			for (int j = 0; j < REGION_PER; ++j) {
				const uint64_t i = begin + threadIdx.x + 512ull * j;
				mine[j] = (i < end) ? items[i] : ITEM_INVALID;
			}
		}
		for (uint32_t i = threadIdx.x; i < REGION_WORDS; i += 512) {
			s_b1[i] = 0;
			s_b2[i] = 0;
		}
		__syncthreads();
		// pass 1: touched / touched again
		auto touch = [&](unsigned long long it) {
			const uint32_t b = (uint32_t)(it >> 31) & (REGION_BITS - 1);
			const uint32_t m = 1u << (b & 31u);
			const uint32_t old = atomicOr(&s_b1[b >> 5], m);
			if (old & m) atomicOr(&s_b2[b >> 5], m);
		};
		if (one_round) {
#pragma unroll
			for (int j = 0; j < REGION_PER; ++j)
				if (mine[j] != ITEM_INVALID) touch(mine[j]);
		} else {
			for (uint64_t i = begin + threadIdx.x; i < end; i += 512) touch(items[i]);
		}
		__syncthreads();
		// rank the contested bits: exclusive prefix over the 4096 words (8 per thread)
		uint32_t local[8], sum = 0, set1 = 0;
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			local[j] = __popc(s_b2[threadIdx.x * 8 + j]);
			sum += local[j];
			set1 += __popc(s_b1[threadIdx.x * 8 + j]);
		}
		uint32_t incl = sum;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t o = __shfl_up(incl, d);
			if (lane >= (uint32_t)d) incl += o;
		}
		if (lane == 63) s_wsum[wave] = incl;
		__syncthreads();
		uint32_t run = incl - sum, nmulti = 0;
#pragma unroll
		for (uint32_t q = 0; q < 8; ++q) {
			const uint32_t t = s_wsum[q];
			if (q < wave) run += t;
			nmulti += t;
		}
		const bool fits = nmulti <= CAP;  // uniform
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			s_pre[threadIdx.x * 8 + j] = (uint16_t)run;  // only used when it fits (< 65536)
			run += local[j];
		}
		setbits_total += set1;
		if (!fits && threadIdx.x == 0) atomicAdd(&counters[BC_REGION_OVF], 1ull);
		for (uint32_t i = threadIdx.x; i < nmulti && i < CAP; i += 512) s_min[i] = INF_STAMP;
		__syncthreads();
		if (fits && nmulti) {
			// pass 2: earliest toucher of every contested bit; pass 3: every later toucher loses that bit check
			auto contested_rank = [&](unsigned long long it, uint32_t &rank) -> bool {
				const uint32_t b = (uint32_t)(it >> 31) & (REGION_BITS - 1);
				const uint32_t m = 1u << (b & 31u);
				const uint32_t w2 = s_b2[b >> 5];
				rank = (uint32_t)s_pre[b >> 5] + __popc(w2 & (m - 1));
				return (w2 & m) != 0;
			};
			if (one_round) {
				// the stamps of the contested items: gathered once, kept with a one-bit-per-item mask
				uint32_t cmask = 0;
				unsigned long long st[4];
				uint32_t rk[4], sl[4], nc = 0;
#pragma unroll
				for (int j = 0; j < REGION_PER; ++j) {
					uint32_t rank;
					if (mine[j] != ITEM_INVALID && contested_rank(mine[j], rank)) {
						cmask |= 1u << j;
#pragma unroll
						for (uint32_t q = 0; q < 4; ++q)
							if (q == nc) {
								rk[q] = rank;
								sl[q] = (uint32_t)(mine[j] & 0x7FFFFFFFull);
							}
						nc++;
					}
				}
				if (nc <= 4) {
#pragma unroll
					for (uint32_t q = 0; q < 4; ++q)
						if (q < nc) st[q] = tab[sl[q]].stamp;
#pragma unroll
					for (uint32_t q = 0; q < 4; ++q)
						if (q < nc) atomicMin(&s_min[rk[q]], st[q]);
				} else {
#pragma unroll
					for (int j = 0; j < REGION_PER; ++j)
						if (cmask & (1u << j)) {
							uint32_t rank;
							contested_rank(mine[j], rank);
							atomicMin(&s_min[rank], (unsigned long long)tab[mine[j] & 0x7FFFFFFFull].stamp);
						}
				}
				__syncthreads();
				if (nc <= 4) {
#pragma unroll
					for (uint32_t q = 0; q < 4; ++q)
						if (q < nc && st[q] > s_min[rk[q]]) atomicAdd(&lost[sl[q] >> 2], 1u << ((sl[q] & 3u) * 8u));
				} else {
#pragma unroll
					for (int j = 0; j < REGION_PER; ++j)
						if (cmask & (1u << j)) {
							uint32_t rank;
							contested_rank(mine[j], rank);
							const uint64_t slot = mine[j] & 0x7FFFFFFFull;
							if (tab[slot].stamp > s_min[rank]) atomicAdd(&lost[slot >> 2], 1u << ((slot & 3u) * 8u));
						}
				}
			} else {
				for (uint64_t i = begin + threadIdx.x; i < end; i += 512) {
					const unsigned long long it = items[i];
					uint32_t rank;
					if (contested_rank(it, rank)) {
						const unsigned long long st = tab[it & 0x7FFFFFFFull].stamp;
						atomicMin(&s_min[rank], st);
						stamp_buf[i] = st;
					}
				}
				__syncthreads();
				for (uint64_t i = begin + threadIdx.x; i < end; i += 512) {
					const unsigned long long it = items[i];
					uint32_t rank;
					if (contested_rank(it, rank)) {
						const uint64_t slot = it & 0x7FFFFFFFull;
						if (stamp_buf[i] > s_min[rank]) atomicAdd(&lost[slot >> 2], 1u << ((slot & 3u) * 8u));
					}
				}
			}
		}
		__syncthreads();
	}
	// set bits of all regions this workgroup handled
	if (threadIdx.x == 0) s_setbits = 0;
	__syncthreads();
	if (setbits_total) atomicAdd(&s_setbits, setbits_total);
	__syncthreads();
	if (threadIdx.x == 0 && s_setbits) atomicAdd(&counters[c_setbits], (unsigned long long)s_setbits);
}

}  // namespace rg
