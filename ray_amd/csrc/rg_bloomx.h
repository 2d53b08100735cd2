// rg_bloomx.h -- Bloom admission for filters of 2^B bits at any fill, without one item per (key, bit).
//
// The reference caps its filter at 2^33 bits (VerticesExtractor.cpp:308-312), so every large input runs with a
// power-of-two filter, and on error-rich reads that filter is dense (C2 with 1 % errors: 1.3e9 keys x 8 bits on
// 8.6e9 bits).  rg_bloom.h then needs 8 items per key and a stamp fetch per contested bit, the dense method 8 B of
// first-touch stamp per filter bit (64 GiB) and 8 scattered atomics per key.  With m_bits = 2^B
//     bit_i(x) = (hash2(x) ^ M_i) & (2^B - 1)                       (BloomFilter.cpp:124-150)
// so with o = hash2(x) & (2^B - 1), the region (bit >> R) of bit_i(x) is (o >> R) ^ c_i with the constants
// c_i = (M_i & (2^B - 1)) >> R: the 8 bits of a key lie in 8 regions that depend on the key's *partition*
// p = o >> R alone, and region r is touched, through hash i, by the keys of partition r ^ c_i and by no others.
// So the keys are partitioned ONCE, by p (one 16-byte item per key instead of eight 8-byte ones), and a workgroup
// that owns region r streams the eight partitions r ^ c_0 .. r ^ c_7:
//
//   k_xor_hist / k_xor_keys   one item (o, slot index, stamp) per key, produced straight into the ranges of the first
//                   partition level (tile-local counting sort in LDS, as k_bloom_scatter_items)
//   k_scatter16     a middle partition level of 16-byte items (filters beyond 2^30 bits)
//   k_xor_final     the last level: writes, per partition and padded to whole wavefronts, the 8-byte
//                   (stamp << 14 | low bits of o) array the regions read and the 4-byte slot indices
//   k_xor_region    one workgroup per region of 2^14 bits: a first-touch stamp per bit in LDS (atomicMin over the
//                   eight streams), then every key whose stamp is a bit's minimum has *won* that bit: one bit per
//                   (hash, key position), written as whole ballots
//   k_xor_fp        a key that won none of its 8 bits found them all set when it first arrived: a false positive
//                   (MessageProcessor.cpp:741-750); those few get their mark, k_admit_lost applies the rule
//
// Same result as the other two methods bit for bit.  Cost per key: 16 B written and read per partition level, 8 x 8 B
// read by the regions -- independent of the fill.
#pragma once
#include "rg_device.h"

namespace rg {

static constexpr int XOR_REGION_LOG2_MAX = 14;  // at most 2^14 first-touch stamps of 8 B = 128 KB of LDS per region
static constexpr int XOR_TILE = 4096;            // 16-byte items per scatter tile (64 KB of LDS)
static constexpr int XOR_THREADS = 512;
static constexpr unsigned long long XOR_PAD = ~0ull;  // padding entry of the packed array
static constexpr int XOR_SPER = 3;               // items per stream a thread of k_xor_region keeps in registers (8 streams)

// (Item16 and xor_origin: rg_partition.h, where k_aggregate makes such items too)

// keys per first-level digit (digit = (o >> shift) & dmask)
template <int W>
__global__ __launch_bounds__(256) void k_xor_hist(const Slot *__restrict__ tab, uint64_t cap, uint64_t m_mask, int shift, uint32_t dmask,
                                                  unsigned long long *__restrict__ hist) {
	__shared__ uint32_t s_hist[256];
	s_hist[threadIdx.x] = 0;
	__syncthreads();
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * 256) {
		const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(&tab[i].k0);
		if (a.x == EMPTY) continue;
		atomicAdd(&s_hist[(uint32_t)(xor_origin<W>(a.x, a.y, m_mask) >> shift) & dmask], 1u);
	}
	__syncthreads();
	if (s_hist[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)s_hist[threadIdx.x]);
}

// shared tail of the three scatter kernels: exclusive scan of the digit counts, one cursor claim per non-empty digit
__device__ __forceinline__ uint32_t xor_claim(uint32_t *s_cnt, uint32_t *s_off, uint32_t *s_scan, unsigned long long *s_gbase,
                                              unsigned long long *__restrict__ cursors) {
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
	__syncthreads();
	return s_scan[4];
}

// keys -> items, sorted by the first-level digit inside a tile of XOR_TILE slots and written as runs
// limits != nullptr: the digits' ranges were sized in advance (o is a hash: a share of the keys each, plus slack) instead of
// counted by k_xor_hist; limits[d] is the end of range d, an item that would land beyond it is dropped and counted in
// counters[XC_PRESIZE_OVF] (the host then falls back to counted ranges).
enum XorCounter { XC_PRESIZE_OVF = 29 };
template <int W>
__global__ __launch_bounds__(XOR_THREADS) void k_xor_keys(const Slot *__restrict__ tab, uint64_t cap, uint64_t m_mask, int shift, uint32_t dmask,
                                                          unsigned long long *__restrict__ cursors, Item16 *__restrict__ out,
                                                          uint8_t *__restrict__ side, int next_shift, uint32_t next_mask,
                                                          const unsigned long long *__restrict__ limits = nullptr,
                                                          unsigned long long *__restrict__ counters = nullptr) {
	constexpr int PER = XOR_TILE / XOR_THREADS;
	__shared__ Item16 s_item[XOR_TILE];
	__shared__ uint8_t s_dig[XOR_TILE];
	__shared__ uint32_t s_cnt[256], s_off[256], s_scan[8];
	__shared__ unsigned long long s_gbase[256];
	const uint64_t base = (uint64_t)blockIdx.x * XOR_TILE;
	if (threadIdx.x < 256) s_cnt[threadIdx.x] = 0;
	__syncthreads();
	Item16 mine[PER];
	uint32_t dr[PER];  // digit << 16 | rank inside the tile
#pragma unroll
	for (int q = 0; q < PER; ++q) {
		const uint64_t i = base + threadIdx.x + (uint64_t)q * XOR_THREADS;
		dr[q] = 0xFFFFFFFFu;
		if (i < cap) {
			const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(&tab[i].k0);
			if (a.x != EMPTY) {
				const uint64_t o = xor_origin<W>(a.x, a.y, m_mask);
				mine[q].a = (o << 31) | (i & 0x7FFFFFFFull);
				mine[q].b = tab[i].stamp | ((i >> 31) << 63);
				const uint32_t d = (uint32_t)(o >> shift) & dmask;
				dr[q] = (d << 16) | atomicAdd(&s_cnt[d], 1u);
			}
		}
	}
	__syncthreads();
	const uint32_t total = xor_claim(s_cnt, s_off, s_scan, s_gbase, cursors);
#pragma unroll
	for (int q = 0; q < PER; ++q)
		if (dr[q] != 0xFFFFFFFFu) {
			const uint32_t pos = s_off[dr[q] >> 16] + (dr[q] & 0xFFFFu);
			s_item[pos] = mine[q];
			s_dig[pos] = (uint8_t)(dr[q] >> 16);
		}
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < total; i += XOR_THREADS) {
		const uint32_t d = s_dig[i];
		const Item16 it = s_item[i];
		const unsigned long long dst = s_gbase[d] + (i - s_off[d]);
		if (limits && dst >= limits[d]) {
			atomicAdd(&counters[XC_PRESIZE_OVF], 1ull);
			continue;
		}
		*reinterpret_cast<ulonglong2 *>(&out[dst]) = make_ulonglong2(it.a, it.b);
		if (side) side[dst] = (uint8_t)((uint32_t)(it.a >> (31 + next_shift)) & next_mask);
	}
}

// A partition level of 16-byte items.  FINAL == false: items out (+ the next level's digit bytes).  FINAL == true: the
// ranges (cursors) start on multiples of 64 positions; written are the packed (stamp << 14 | o & low_mask) words and the
// slot indices.
template <bool FINAL>
__global__ __launch_bounds__(XOR_THREADS) void k_scatter16(const Item16 *__restrict__ in, uint32_t S, const unsigned long long *__restrict__ seg_begin,
                                                           const uint32_t *__restrict__ tile_first, int shift, int dbits,
                                                           unsigned long long *__restrict__ cursors, Item16 *__restrict__ out, uint8_t *__restrict__ side,
                                                           int next_shift, uint32_t next_mask, unsigned long long *__restrict__ pk,
                                                           uint32_t *__restrict__ ix, uint64_t low_mask,
                                                           const unsigned long long *__restrict__ seg_end = nullptr, uint32_t seg_div = 1) {
	// seg_div: the input is the runs k_aggregate left (AggItems) -- seg_div consecutive segments, one per workgroup of that
	// kernel, share the cursors of output segment seg / seg_div
	constexpr int PER = XOR_TILE / XOR_THREADS;
	__shared__ Item16 s_item[XOR_TILE];
	__shared__ uint8_t s_dig[XOR_TILE];
	__shared__ uint32_t s_cnt[256], s_off[256], s_scan[8];
	__shared__ unsigned long long s_gbase[256];
	uint32_t lo = 0, hi = S;  // largest s with tile_first[s] <= tile
	while (hi - lo > 1) {
		const uint32_t mid = (lo + hi) >> 1;
		if (tile_first[mid] <= blockIdx.x) lo = mid;
		else hi = mid;
	}
	const uint32_t seg = lo;
	const uint64_t begin = seg_begin[seg] + (uint64_t)(blockIdx.x - tile_first[seg]) * XOR_TILE;
	const uint64_t send = seg_end ? seg_end[seg] : seg_begin[seg + 1];
	const uint64_t end = begin + XOR_TILE < send ? begin + XOR_TILE : send;
	const uint32_t dmask = (1u << dbits) - 1;
	if (threadIdx.x < 256) s_cnt[threadIdx.x] = 0;
	__syncthreads();
	Item16 mine[PER];
	uint32_t dr[PER];
#pragma unroll
	for (int q = 0; q < PER; ++q) {
		const uint64_t i = begin + threadIdx.x + (uint64_t)q * XOR_THREADS;
		dr[q] = 0xFFFFFFFFu;
		if (i < end) {
			const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(&in[i]);
			mine[q].a = v.x;
			mine[q].b = v.y;
			const uint32_t d = (uint32_t)(v.x >> (31 + shift)) & dmask;
			dr[q] = (d << 16) | atomicAdd(&s_cnt[d], 1u);
		}
	}
	__syncthreads();
	const uint32_t total = xor_claim(s_cnt, s_off, s_scan, s_gbase, cursors + ((uint64_t)(seg / seg_div) << dbits));
#pragma unroll
	for (int q = 0; q < PER; ++q)
		if (dr[q] != 0xFFFFFFFFu) {
			const uint32_t pos = s_off[dr[q] >> 16] + (dr[q] & 0xFFFFu);
			s_item[pos] = mine[q];
			s_dig[pos] = (uint8_t)(dr[q] >> 16);
		}
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < total; i += XOR_THREADS) {
		const uint32_t d = s_dig[i];
		const Item16 it = s_item[i];
		const unsigned long long dst = s_gbase[d] + (i - s_off[d]);
		if (FINAL) {
			pk[dst] = ((it.b & ~(1ull << 63)) << 14) | ((it.a >> 31) & low_mask);
			ix[dst] = (uint32_t)(it.a & 0x7FFFFFFFull) | ((uint32_t)(it.b >> 63) << 31);
		} else {
			*reinterpret_cast<ulonglong2 *>(&out[dst]) = make_ulonglong2(it.a, it.b);
			if (side) side[dst] = (uint8_t)((uint32_t)(it.a >> (31 + next_shift)) & next_mask);
		}
	}
}

// counts -> counts rounded up to whole wavefronts (the final level's ranges are padded so that a wavefront's 64
// consecutive positions never straddle two partitions)
__global__ __launch_bounds__(256) void k_round64(unsigned long long *__restrict__ v, uint64_t n) {
	const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) v[i] = (v[i] + 63ull) & ~63ull;
}

struct XorConsts {
	uint32_t c[8];    // partition that touches region r through hash i: r ^ c[i]
	uint32_t low[8];  // M_i & (2^R - 1)
};

// One workgroup per region of 2^R filter bits (R <= 14).  part_begin: the padded ranges of the final level (P + 1
// entries, multiples of 64).  planes: bit j of plane i = the key at position j won bit_i, i.e. its arrival set it.
// RMAX, THREADS: 2^RMAX cells of LDS and the workgroup size -- (14, 1024): one workgroup per CU; (13, 512) / (12, 256): two /
// four, whose loads and LDS phases overlap (a region is a chain of dependent loads: range bounds, then items).
// The cells are cleared once: a region writes its stamps under a tag that falls from region to region (the top 14 bits of the
// cell; a stamp is 49 bits), so what earlier regions left in a cell is larger than anything this one offers to the atomic
// minimum and equal to nothing it compares with.  count_winners: the set bits are the winners (one per touched bit: stamps
// are unique, and the host checks that no two hashes of a key can name the same bit) instead of a sweep over the cells.
// Together: no clearing and no sweep per region (2 x 16 LDS operations per thread at 2^14 cells, beside some 40 for the
// keys of a crowded filter) and one barrier less.
template <int RMAX, int THREADS>
__global__ __launch_bounds__(THREADS) void k_xor_region(const unsigned long long *__restrict__ pk, const unsigned long long *__restrict__ part_begin,
                                                        uint32_t P, int R, XorConsts K, unsigned long long *__restrict__ planes, uint64_t plane_words,
                                                        unsigned long long *__restrict__ counters, int c_setbits, int count_winners) {
	__shared__ unsigned long long s_cell[1u << RMAX];
	__shared__ uint32_t s_setbits;
	const uint32_t ncell = 1u << R, cmask = ncell - 1;
	const uint32_t lane = threadIdx.x & 63u;
	uint32_t set_total = 0;
	if (threadIdx.x == 0) s_setbits = 0;
	constexpr uint32_t EPOCHS = 1u << 14;
	uint32_t epoch = EPOCHS;
	for (uint32_t r = blockIdx.x; r < P; r += gridDim.x) {
		if (epoch == EPOCHS) {
			for (uint32_t i = threadIdx.x; i < ncell; i += THREADS) s_cell[i] = INF_STAMP;
			epoch = 0;
			__syncthreads();
		}
		const unsigned long long tag = (unsigned long long)(EPOCHS - 1u - epoch) << 50;
		++epoch;
		// the eight streams of this region, laid end to end (every length is a multiple of 64)
		unsigned long long sb[8];
		uint32_t pre[9];
		pre[0] = 0;
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			const uint32_t q = r ^ K.c[i];
			sb[i] = part_begin[q];
			pre[i + 1] = pre[i] + (uint32_t)(part_begin[q + 1] - sb[i]);
		}
		uint32_t longest = 0;
#pragma unroll
		for (int i = 0; i < 8; ++i) longest = max(longest, pre[i + 1] - pre[i]);
		if (longest <= (uint32_t)THREADS * XOR_SPER) {
			// one read: the items stay in registers between the two passes, XOR_SPER per stream and thread (the stream index
			// is a compile-time constant everywhere: no selects, the constants stay in scalar registers)
			unsigned long long item[8][XOR_SPER];
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				const uint32_t len = pre[i + 1] - pre[i];
#pragma unroll
				for (int u = 0; u < XOR_SPER; ++u) {
					const uint32_t t = threadIdx.x + (uint32_t)THREADS * u;
					item[i][u] = (t < len) ? pk[sb[i] + t] : XOR_PAD;
				}
			}
#pragma unroll
			for (int i = 0; i < 8; ++i)
#pragma unroll
				for (int u = 0; u < XOR_SPER; ++u)
					if (item[i][u] != XOR_PAD) atomicMin(&s_cell[((uint32_t)item[i][u] ^ K.low[i]) & cmask], tag | (item[i][u] >> 14));
			__syncthreads();
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				const uint32_t len = pre[i + 1] - pre[i];
#pragma unroll
				for (int u = 0; u < XOR_SPER; ++u) {
					const uint32_t t = threadIdx.x + (uint32_t)THREADS * u;
					if (t - lane < len) {  // wave-uniform: every stream is a whole number of wavefronts long
						const bool win = item[i][u] != XOR_PAD && s_cell[((uint32_t)item[i][u] ^ K.low[i]) & cmask] == (tag | (item[i][u] >> 14));
						const unsigned long long bal = __ballot(win);
						if (lane == 0) {
							planes[(uint64_t)i * plane_words + ((sb[i] + t) >> 6)] = bal;
							if (count_winners) set_total += (uint32_t)__popcll(bal);
						}
					}
				}
			}
		} else {
			// a region with more keys than the registers hold: the streams are read twice
#pragma unroll 1
			for (int i = 0; i < 8; ++i) {
				const uint32_t len = pre[i + 1] - pre[i];
				for (uint32_t t = threadIdx.x; t < len; t += THREADS) {
					const unsigned long long it = pk[sb[i] + t];
					if (it != XOR_PAD) atomicMin(&s_cell[((uint32_t)it ^ K.low[i]) & cmask], tag | (it >> 14));
				}
			}
			__syncthreads();
#pragma unroll 1
			for (int i = 0; i < 8; ++i) {
				const uint32_t len = pre[i + 1] - pre[i];
				for (uint32_t t = threadIdx.x; t < len; t += THREADS) {
					const unsigned long long it = pk[sb[i] + t];
					const bool win = it != XOR_PAD && s_cell[((uint32_t)it ^ K.low[i]) & cmask] == (tag | (it >> 14));
					const unsigned long long bal = __ballot(win);
					if (lane == 0) {
						planes[(uint64_t)i * plane_words + ((sb[i] + t) >> 6)] = bal;
						if (count_winners) set_total += (uint32_t)__popcll(bal);
					}
				}
			}
		}
		if (!count_winners)  // bits of this region that any key touched: the cells that carry its tag
			for (uint32_t i = threadIdx.x; i < ncell; i += THREADS) set_total += ((s_cell[i] >> 50) == (tag >> 50) && s_cell[i] != INF_STAMP) ? 1u : 0u;
		__syncthreads();  // the next region's stamps must not land before every comparison above has read its cell
	}
	if (set_total) atomicAdd(&s_setbits, set_total);
	__syncthreads();
	if (threadIdx.x == 0 && s_setbits) atomicAdd(&counters[c_setbits], (unsigned long long)s_setbits);
}

// Workgroup barrier for data that travels through LDS only (see rg_partition.h: __syncthreads() would also wait for the
// global loads that are in flight for the next region)
__device__ __forceinline__ void xor_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// k_xor_region for sparse filters: regions whose eight streams hold at most THREADS keys each (one item per stream and thread).
// A region is a chain of two dependent loads (the bounds of its eight partitions, then their items) in front of a few LDS
// operations; with one workgroup per CU nothing hid them (C2 error-free: 14 ms for 6.4 GB whatever the region size).  Here the
// chain is software-pipelined over the workgroup's regions: while region r is resolved in LDS, the items of region r + G are
// on their way and so are the bounds of region r + 2G (G = the grid).  The bounds are loaded by lanes (lane i & 7 fetches stream
// i's) and handed round with readlane, so that they are vector loads like the items: nothing here waits on the scalar cache
// behind the LDS barriers.  count_winners: the filter's set bits are counted as the winners (one per set bit when the eight
// hashes of a key cannot share a bit, which the host checks), not by a sweep over the cells.
template <int RMAX, int THREADS>
__global__ __launch_bounds__(THREADS) void k_xor_region_sparse(const unsigned long long *__restrict__ pk, const unsigned long long *__restrict__ part_begin,
                                                               uint32_t P, int R, XorConsts K, unsigned long long *__restrict__ planes, uint64_t plane_words,
                                                               unsigned long long *__restrict__ counters, int c_setbits, int count_winners) {
	__shared__ unsigned long long s_cell[1u << RMAX];
	__shared__ uint32_t s_setbits;
	const uint32_t ncell = 1u << R, cmask = ncell - 1;
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t G = gridDim.x;
	uint32_t set_total = 0;
	if (threadIdx.x == 0) s_setbits = 0;
	uint32_t my_c = K.c[0];  // the stream whose bounds this lane fetches
#pragma unroll
	for (int i = 1; i < 8; ++i)
		if ((lane & 7u) == (uint32_t)i) my_c = K.c[i];
	// bounds as fetched (per lane), and handed round (uniform)
	auto fetch_bounds = [&](uint32_t r, unsigned long long &b0, unsigned long long &b1) {
		const uint32_t q = (r < P ? r : 0u) ^ my_c;
		b0 = part_begin[q];
		b1 = part_begin[q + 1];
	};
	auto spread = [&](unsigned long long b0, unsigned long long b1, unsigned long long (&sb)[8], uint32_t (&len)[8]) -> bool {
		bool fits = true;
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b0, i), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b0 >> 32), i);
			const uint32_t elo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b1, i), ehi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b1 >> 32), i);
			sb[i] = ((unsigned long long)hi << 32) | lo;
			len[i] = (uint32_t)((((unsigned long long)ehi << 32) | elo) - sb[i]);
			fits = fits && len[i] <= (uint32_t)THREADS;
		}
		return fits;
	};
	uint32_t r = blockIdx.x;
	unsigned long long sbA[8], itA[8];
	uint32_t lenA[8];
	unsigned long long nb0 = 0, nb1 = 0;  // raw bounds of region r + G
	bool fitsA = false;
	{
		unsigned long long b0, b1;
		fetch_bounds(r, b0, b1);
		fitsA = spread(b0, b1, sbA, lenA);
#pragma unroll
		for (int i = 0; i < 8; ++i) itA[i] = (fitsA && r < P && threadIdx.x < lenA[i]) ? pk[sbA[i] + threadIdx.x] : XOR_PAD;
		fetch_bounds(r + G, nb0, nb1);
	}
	// The cells are cleared once.  A region writes its stamps under a tag that falls from region to region (the top 14 bits of
	// the cell: a stamp is 49 bits), so whatever earlier regions left in a cell is larger than anything this one offers to the
	// atomic minimum and equal to nothing it compares with: no clearing between regions (sixteen stores per thread, then eight
	// with only the touched cells freed: 6.5 and 6.0 ms on C2) and one barrier less; after 2^14 regions the cells are cleared again.
	constexpr uint32_t EPOCHS = 1u << 14;
	uint32_t epoch = EPOCHS;
	for (; r < P; r += G) {
		// the next region's items, the bounds of the one after
		unsigned long long sbB[8], itB[8];
		uint32_t lenB[8];
		const bool haveB = r + G < P;
		const bool fitsB = spread(nb0, nb1, sbB, lenB) && haveB;
#pragma unroll
		for (int i = 0; i < 8; ++i) itB[i] = (fitsB && threadIdx.x < lenB[i]) ? pk[sbB[i] + threadIdx.x] : XOR_PAD;
		fetch_bounds(r + 2 * G, nb0, nb1);

		if (epoch == EPOCHS) {
			for (uint32_t i = threadIdx.x; i < ncell; i += THREADS) s_cell[i] = INF_STAMP;
			xor_lds_barrier();
			epoch = 0;
		}
		const unsigned long long tag = (unsigned long long)(EPOCHS - 1u - epoch) << 50;
		++epoch;
		if (fitsA) {
#pragma unroll
			for (int i = 0; i < 8; ++i)
				if (itA[i] != XOR_PAD) atomicMin(&s_cell[((uint32_t)itA[i] ^ K.low[i]) & cmask], tag | (itA[i] >> 14));
			xor_lds_barrier();
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				if (threadIdx.x - lane < lenA[i]) {  // wave-uniform: every stream is a whole number of wavefronts long
					const bool win = itA[i] != XOR_PAD && s_cell[((uint32_t)itA[i] ^ K.low[i]) & cmask] == (tag | (itA[i] >> 14));
					const unsigned long long bal = __ballot(win);
					if (lane == 0) {
						planes[(uint64_t)i * plane_words + ((sbA[i] + threadIdx.x) >> 6)] = bal;
						if (count_winners) set_total += (uint32_t)__popcll(bal);
					}
				}
			}
		} else {
			// a region with a stream longer than the workgroup: read twice
#pragma unroll 1
			for (int i = 0; i < 8; ++i)
				for (uint32_t t = threadIdx.x; t < lenA[i]; t += THREADS) {
					const unsigned long long it = pk[sbA[i] + t];
					if (it != XOR_PAD) atomicMin(&s_cell[((uint32_t)it ^ K.low[i]) & cmask], tag | (it >> 14));
				}
			xor_lds_barrier();
#pragma unroll 1
			for (int i = 0; i < 8; ++i)
				for (uint32_t t = threadIdx.x; t < lenA[i]; t += THREADS) {
					const unsigned long long it = pk[sbA[i] + t];
					const bool win = it != XOR_PAD && s_cell[((uint32_t)it ^ K.low[i]) & cmask] == (tag | (it >> 14));
					const unsigned long long bal = __ballot(win);
					if (lane == 0) {
						planes[(uint64_t)i * plane_words + ((sbA[i] + t) >> 6)] = bal;
						if (count_winners) set_total += (uint32_t)__popcll(bal);
					}
				}
		}
		if (!count_winners)  // bits of this region that any key touched: the cells that carry its tag
			for (uint32_t i = threadIdx.x; i < ncell; i += THREADS) set_total += ((s_cell[i] >> 50) == (tag >> 50) && s_cell[i] != INF_STAMP) ? 1u : 0u;
		xor_lds_barrier();  // the next region's stamps must not land before every comparison above has read its cell
		fitsA = fitsB;
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			sbA[i] = sbB[i];
			lenA[i] = lenB[i];
			itA[i] = itB[i];
		}
	}
	if (set_total) atomicAdd(&s_setbits, set_total);
	__syncthreads();
	if (threadIdx.x == 0 && s_setbits) atomicAdd(&counters[c_setbits], (unsigned long long)s_setbits);
}

// a key that won none of its eight bits is a false positive: its byte of `lost` becomes 8 (what k_admit_lost tests).
// A wavefront's 64 positions share one word of each plane: lanes 0..7 fetch the eight words in one load and the OR is handed
// round (eight loads of one word for all lanes: C5's share per GPU 40.8 ms at 0.9 TB/s).
// Tried and dropped for crowded filters, where a quarter of the keys end here: the marks written from a list sorted by
// stretch of the log (a pass that lists, a tile sort, a pass that marks): 21 + 17 ms against 40 on C5's share, 9.5 + 5.5
// against 10.7 on C2 with 1 % errors -- the scattered bytes were not what the time went on.
// One position per lane and trip kept 512 bytes per wavefront in flight: 1.9 TB/s on C2 with 1 % errors (8.3 ms), 1.1 on C5's
// share per GPU (34 ms).  A wavefront now takes 256 positions a trip, four consecutive ones per lane: the 8 x 4 plane words
// come in one load by lanes 0..31, and only a lane with a candidate among its four -- a position that won nothing, i.e. a
// false positive or a pad -- loads their packed words (32 bytes) to tell the two apart, and then the log position.
__global__ __launch_bounds__(256) void k_xor_fp(const unsigned long long *__restrict__ pk, const uint32_t *__restrict__ ix,
                                                const unsigned long long *__restrict__ planes, uint64_t plane_words, uint64_t npos,
                                                uint8_t *__restrict__ lost, unsigned long long *__restrict__ fp_count = nullptr) {
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint64_t nwaves = (uint64_t)gridDim.x * 4;
	uint32_t marked = 0;  // false positives this lane found (their number bounds what admission stores: sizes the table build)
	for (uint64_t w0 = ((uint64_t)blockIdx.x * 4 + wave) * 4; w0 < plane_words; w0 += nwaves * 4) {  // (wave-uniform) four plane words = 256 positions
		unsigned long long mine = 0;
		{
			const uint64_t word = w0 + (lane & 3u);
			if (lane < 32 && word < plane_words) mine = planes[(uint64_t)(lane >> 2) * plane_words + word];
		}
		// OR over the eight planes: lanes l, l + 4, ..., l + 28 hold the same word of different planes
		uint32_t lo = (uint32_t)mine, hi = (uint32_t)(mine >> 32);
		lo |= (uint32_t)__shfl_xor((int)lo, 4);
		hi |= (uint32_t)__shfl_xor((int)hi, 4);
		lo |= (uint32_t)__shfl_xor((int)lo, 8);
		hi |= (uint32_t)__shfl_xor((int)hi, 8);
		lo |= (uint32_t)__shfl_xor((int)lo, 16);
		hi |= (uint32_t)__shfl_xor((int)hi, 16);
		// lane l's positions: bits 4 (l & 15) .. + 3 of word l >> 4, which lanes (l >> 4) + 4 i hold (lane l >> 4 among them)
		const uint32_t src = lane >> 4;
		const uint32_t wl = (uint32_t)__shfl((int)lo, (int)src), wh = (uint32_t)__shfl((int)hi, (int)src);
		const uint32_t sh = 4u * (lane & 15u);
		const uint32_t won = (sh < 32u ? wl >> sh : wh >> (sh - 32u)) & 15u;
		const uint64_t j = (w0 << 6) + 4ull * lane;
		if (won == 15u || j >= npos) continue;
		unsigned long long key[4];
		if (j + 4 <= npos) {
			const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(pk + j), b = *reinterpret_cast<const ulonglong2 *>(pk + j + 2);
			key[0] = a.x;
			key[1] = a.y;
			key[2] = b.x;
			key[3] = b.y;
		} else {
#pragma unroll
			for (int q = 0; q < 4; ++q) key[q] = j + q < npos ? pk[j + q] : XOR_PAD;
		}
#pragma unroll
		for (int q = 0; q < 4; ++q)
			if (!((won >> q) & 1u) && key[q] != XOR_PAD) {
				lost[ix[j + q]] = 8;
				marked++;
			}
	}
	if (fp_count) {
		for (int d = 32; d > 0; d >>= 1) marked += __shfl_down(marked, d);
		if (lane == 0 && marked) atomicAdd(fp_count, (unsigned long long)marked);
	}
}

}  // namespace rg
