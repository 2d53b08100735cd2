// rg_table.h -- the vertex table built without global atomics.
//
// k_insert_admitted claims one slot per vertex with a device-scope compare-and-swap on a random sector of a
// multi-GiB table and writes the rest of the slot after the line has left L2: bound by the atomic rate (11.6 G/s)
// and 3.6x its bytes in write traffic, on top of a memset of the whole table.  The admitted entries are distinct
// keys, so nothing has to be atomic if one workgroup owns a whole stretch of the table:
//
//   k_entry_hist + k_scatter_entries   the admitted entries are partitioned by the table range their home slot
//                   lies in.  A range is a stretch of the *hash*: range = top `bits` bits of slot_hash, whose slots are
//                   [umulhi(r << (64 - bits), cap), umulhi((r + 1) << (64 - bits), cap)) -- at most 4096 of them -- so
//                   the partition (two or three levels of at most 8 bits, tile-local counting sort in LDS, runs out: the
//                   scheme of k_scatter on 32-byte entries) does not depend on the capacity, and its first level can be
//                   counted before the number of admitted vertices, hence the capacity, is known: by the admission pass
//                   itself (k_admit_hist, raygraph.hip), which then need not write the log at all -- the first level
//                   applies the admission rule again to what it reads (`lost` != nullptr)
//   k_build_ranges  one workgroup per range: the range's slots are filled in LDS (linear probing inside the
//                   range, LDS compare-and-swap), then written out once, free slots included -- no k_init_table,
//                   every byte of the table is written exactly once, coalesced
//   k_insert_admitted   only for the few entries whose probe sequence leaves their range (the overflow list)
//
// Look-ups are the table's usual linear probing from the home slot: an entry that left its range was placed by
// the ordinary insert after every range had been written, so there is no free slot between its home and itself.
#pragma once
#include "rg_device.h"
#include "rg_partition.h"
#include <type_traits>

namespace rg {

static constexpr int ENT_TILE = 2048;     // entries per scatter tile: 64 KB of LDS
static constexpr int ENT_THREADS = 512;

// An admitted entry on its way to the table.  The arrival stamp has done its work once admission is decided and one-word keys
// have no second word, so they travel as 16 bytes (half the bytes of every pass between the log and the table, tiles of
// twice the entries); two-word keys keep the slot's 32.
struct Ent16 {
	uint64_t k0, meta;
};
template <int W>
struct EntT {
	using type = Slot;
};
template <>
struct EntT<1> {
	using type = Ent16;
};
template <int W>
using Ent = typename EntT<W>::type;
template <int W>
constexpr int ent_tile() { return ENT_TILE * (int)(sizeof(Slot) / sizeof(Ent<W>)); }
__device__ __forceinline__ Slot ent_load(const Slot *p) {
	const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(&p->k0), b = *reinterpret_cast<const ulonglong2 *>(&p->stamp);
	Slot v;
	v.k0 = a.x;
	v.k1 = a.y;
	v.stamp = b.x;
	v.meta = b.y;
	return v;
}
__device__ __forceinline__ Slot ent_load(const Ent16 *p) {
	const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(p);
	Slot v;
	v.k0 = a.x;
	v.k1 = EMPTY;
	v.stamp = INF_STAMP;
	v.meta = a.y;
	return v;
}
__device__ __forceinline__ void ent_pack(const Slot &v, Slot *out) { *out = v; }
__device__ __forceinline__ void ent_pack(const Slot &v, Ent16 *out) {
	out->k0 = v.k0;
	out->meta = v.meta;
}

enum TableCounter { TC_OVF_CURSOR = 27, TC_OVF_LOST = 28 };

// range index of an entry: the top `bits` bits of its slot hash
template <int W>
__device__ __forceinline__ uint32_t entry_range(const Slot &v, int bits) {
	return (uint32_t)(slot_hash<W>(v.k0, W == 2 ? v.k1 : 0) >> (64 - bits));
}

__device__ __forceinline__ bool entry_admitted(const Slot &v) { return v.k0 != EMPTY && (v.meta & COUNT_MASK) != 0; }

// an entry of the log as admission leaves it, from the entry as the sweep wrote it and its "lost" mark (the rule of
// k_admit_lost: MessageProcessor.cpp:741-804); false = not stored
__device__ __forceinline__ bool admit_entry(Slot &v, const uint8_t *__restrict__ lost, uint64_t i) {
	if (v.k0 == EMPTY) return false;
	const uint64_t N = v.meta & COUNT_MASK & ~LOG_UNIQUE;
	uint64_t cov = (lost[i] == 8) ? N + 1 : (N >= 2 ? N : 0);
	if (cov > 0xFFFFFFFFull) cov = 0xFFFFFFFFull;
	if (cov == 0) return false;
	v.meta = admitted_meta(v.meta, cov);
	return true;
}

// first level's digit histogram: admitted entries of the log (admitted in place) per top digit
template <int W>
__global__ __launch_bounds__(256) void k_entry_hist(const Slot *__restrict__ in, uint64_t n, int bits, int shift, uint32_t dmask,
                                                    unsigned long long *__restrict__ hist) {
	__shared__ uint32_t s_cnt[256];
	s_cnt[threadIdx.x] = 0;
	__syncthreads();
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
		const Slot v = in[i];
		if (entry_admitted(v)) atomicAdd(&s_cnt[(entry_range<W>(v, bits) >> shift) & dmask], 1u);
	}
	__syncthreads();
	if (s_cnt[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
}

// The first level can be the admission's only pass over the log (round 3): with `st.counters` set it gathers what the counting
// pass gathered -- vertices stored, Bloom insertions, the coverage histogram (256 bins in LDS, the rest straight to the dense
// bins and the overflow list) -- and writes into ranges sized in advance (st.limits[d] = end of the range of digit d: the top
// byte of a slot hash takes an even share of the admitted entries, whose number is bounded before the pass by the entries seen
// twice or more plus the false positives); an entry that finds its range full is counted in counters[st.c_presize] and the
// host goes back to counting first.
static constexpr uint32_t ADMIT_SPREAD = 256;
struct AdmitStats {
	unsigned long long *counters;    // nullptr: no statistics, exact ranges (the classic first level)
	unsigned long long *hist_dense;  // [hist_dense_n]
	uint32_t *hist_ovf;              // [hist_ovf_cap], filled through counters[c_ovf]
	const unsigned long long *limits;
	unsigned long long *spread;      // [ADMIT_SPREAD][258]: the 256 low histogram bins, vertices stored and Bloom insertions of a tile go to copy
	                                 // blockIdx.x % ADMIT_SPREAD (a tile per workgroup: hundreds of thousands of workgroups adding to the same few
	                                 // addresses took 67 ms on C2 with 1 % errors where the counting pass, 2048 workgroups, took 7.6); k_fold_admit_stats sums them
	uint32_t hist_dense_n, hist_ovf_cap;
	int c_retained, c_insertions, c_ovf, c_presize;
	uint32_t compact;                // one-word keys: the log holds 16-byte entries (AggItems::compact)
};

// A compact log (AggItems::compact) in the full form, for the routes that do not read the compact one: the entries, and -- from
// the items k_aggregate made beside them, while they are held -- the arrival stamps (without them the stamps are INF_STAMP, which
// is all the routes behind a decided admission need).
__global__ __launch_bounds__(256) void k_expand_entries(const Ent16 *__restrict__ in, uint64_t n, Slot *__restrict__ out) {
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
		const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(&in[i]);
		*reinterpret_cast<ulonglong2 *>(&out[i].k0) = make_ulonglong2(a.x, EMPTY);
		*reinterpret_cast<ulonglong2 *>(&out[i].stamp) = make_ulonglong2(INF_STAMP, a.y);
	}
}
__global__ __launch_bounds__(256) void k_expand_stamps(const Item16 *__restrict__ items, const uint32_t *__restrict__ run_count, uint32_t run_cap,
                                                       Slot *__restrict__ out, uint64_t n) {
	const uint32_t run = blockIdx.x, c = run_count[run];
	const Item16 *__restrict__ p = items + (uint64_t)run * run_cap;
	for (uint32_t i = threadIdx.x; i < c; i += 256) {
		const ulonglong2 it = *reinterpret_cast<const ulonglong2 *>(&p[i]);
		const uint64_t dst = (it.x & 0x7FFFFFFFull) | ((it.y >> 63) << 31);
		if (dst < n) out[dst].stamp = it.y & ~(1ull << 63);
	}
}

// One partition level of 32-byte entries (see k_scatter for the scheme).  FIRST: the input is the log of the sweep --
// unused chunk entries and entries that are not admitted are dropped here; lost != nullptr: the log is as the sweep wrote
// it and the admission rule is applied on the way (lost[i] = the mark of entry i), lost == nullptr: admitted in place.
template <int W, bool FIRST>
__global__ __launch_bounds__(ENT_THREADS) void k_scatter_entries(const std::conditional_t<FIRST, Slot, Ent<W>> *__restrict__ in, uint32_t S,
                                                                 const unsigned long long *__restrict__ seg_begin,
                                                                 const uint32_t *__restrict__ tile_first, int bits, int shift, int dbits,
                                                                 unsigned long long *__restrict__ cursors, Ent<W> *__restrict__ out,
                                                                 uint8_t *__restrict__ side, int next_shift, uint32_t next_mask,
                                                                 const uint8_t *__restrict__ lost, AdmitStats st,
                                                                 const unsigned long long *__restrict__ seg_end = nullptr) {
	// seg_end: the input segments are ranges sized in advance and filled in part (the level behind a first level with st.limits)
	using E = Ent<W>;
	constexpr int TILE = ent_tile<W>();
	constexpr int PER = TILE / ENT_THREADS;
	__shared__ uint32_t s_h[FIRST ? 256 : 1];
	__shared__ E s_ent[TILE];
	__shared__ uint8_t s_dig[TILE], s_next[TILE];
	__shared__ uint32_t s_cnt[256], s_off[256], s_scan[8];
	__shared__ unsigned long long s_gbase[256];
	uint32_t lo = 0, hi = S;  // largest s with tile_first[s] <= tile
	while (hi - lo > 1) {
		const uint32_t mid = (lo + hi) >> 1;
		if (tile_first[mid] <= blockIdx.x) lo = mid;
		else hi = mid;
	}
	const uint32_t seg = lo;
	const uint64_t begin = seg_begin[seg] + (uint64_t)(blockIdx.x - tile_first[seg]) * TILE;
	const uint64_t send = seg_end ? seg_end[seg] : seg_begin[seg + 1];
	const uint64_t end = begin + TILE < send ? begin + TILE : send;
	const uint32_t dmask = (1u << dbits) - 1;
	if (threadIdx.x < 256) s_cnt[threadIdx.x] = 0;
	const bool stats = FIRST && st.counters != nullptr;
	if (FIRST && threadIdx.x < 256) s_h[threadIdx.x] = 0;
	uint32_t retained = 0, inserted = 0;
	__syncthreads();
	E mine[PER];
	uint32_t dig[PER], rank[PER], nxt[PER];
#pragma unroll
	for (int j = 0; j < PER; ++j) {
		const uint64_t i = begin + threadIdx.x + (uint64_t)j * ENT_THREADS;
		dig[j] = 0xFFFFFFFFu;
		rank[j] = 0;
		nxt[j] = 0;
		if (i < end) {
			Slot v;
			if constexpr (FIRST && W == 1) v = st.compact ? ent_load(reinterpret_cast<const Ent16 *>(in) + i) : ent_load(&in[i]);
			else v = ent_load(&in[i]);
			bool keep = true;
			if (FIRST && stats) {  // the admission rule with its statistics (k_admit_lost: MessageProcessor.cpp:741-804)
				keep = false;
				if (v.k0 != EMPTY) {
					const uint64_t N = v.meta & COUNT_MASK & ~LOG_UNIQUE;
					const bool fp = lost[i] == 8;
					uint64_t cov = fp ? N + 1 : (N >= 2 ? N : 0);
					if (cov > 0xFFFFFFFFull) cov = 0xFFFFFFFFull;
					if (!fp) inserted++;
					if (cov) {
						retained++;
						if (cov < 256) atomicAdd(&s_h[(uint32_t)cov], 1u);
						else if (cov < st.hist_dense_n) atomicAdd(&st.hist_dense[cov], 1ull);
						else {
							const unsigned long long pos = atomicAdd(&st.counters[st.c_ovf], 1ull);
							if (pos < st.hist_ovf_cap) st.hist_ovf[pos] = (uint32_t)cov;
						}
						v.meta = admitted_meta(v.meta, cov);
						keep = true;
					}
				}
			} else if (FIRST) {
				keep = lost ? admit_entry(v, lost, i) : entry_admitted(v);
			}
			if (keep) {
				const uint32_t r = entry_range<W>(v, bits);
				dig[j] = (r >> shift) & dmask;
				nxt[j] = (r >> next_shift) & next_mask;
				rank[j] = atomicAdd(&s_cnt[dig[j]], 1u);
				ent_pack(v, &mine[j]);
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
			if (c) g = atomicAdd(&cursors[((uint64_t)seg << dbits) + threadIdx.x], (unsigned long long)c);
			s_gbase[threadIdx.x] = g;
		}
	}
	__syncthreads();
	const uint32_t total = s_scan[4];
#pragma unroll
	for (int j = 0; j < PER; ++j)
		if (dig[j] != 0xFFFFFFFFu) {
			const uint32_t pos = s_off[dig[j]] + rank[j];
			s_ent[pos] = mine[j];
			s_dig[pos] = (uint8_t)dig[j];
			s_next[pos] = (uint8_t)nxt[j];
		}
	__syncthreads();
	// 16 bytes per thread and step (two threads per 32-byte entry): a wavefront writes 1 KB of consecutive bytes of a run
	constexpr uint32_t PARTS = sizeof(E) / 16;
	for (uint32_t i2 = threadIdx.x; i2 < PARTS * total; i2 += ENT_THREADS) {
		const uint32_t i = i2 / PARTS, part = i2 % PARTS;
		const uint32_t d = s_dig[i];
		const unsigned long long dst = s_gbase[d] + (i - s_off[d]);
		if (FIRST && stats && st.limits && dst >= st.limits[d]) {
			if (part == 0) atomicAdd(&st.counters[st.c_presize], 1ull);
			continue;
		}
		const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(reinterpret_cast<const uint64_t *>(&s_ent[i]) + 2 * part);
		*reinterpret_cast<ulonglong2 *>(reinterpret_cast<uint64_t *>(&out[dst]) + 2 * part) = v;
		if (side && part == 0) side[dst] = s_next[i];
	}
	if (FIRST && stats) {
		for (int d = 32; d > 0; d >>= 1) {
			retained += __shfl_down(retained, d);
			inserted += __shfl_down(inserted, d);
		}
		unsigned long long *copy = st.spread + (size_t)(blockIdx.x % ADMIT_SPREAD) * 258;
		if ((threadIdx.x & 63) == 0) {
			if (retained) atomicAdd(&copy[256], (unsigned long long)retained);
			if (inserted) atomicAdd(&copy[257], (unsigned long long)inserted);
		}
		if (threadIdx.x < 256 && s_h[threadIdx.x]) atomicAdd(&copy[threadIdx.x], (unsigned long long)s_h[threadIdx.x]);
	}
}

// the copies of AdmitStats::spread summed into the histogram's dense bins and the two counters
__global__ __launch_bounds__(320) void k_fold_admit_stats(AdmitStats st) {
	if (threadIdx.x >= 258) return;
	unsigned long long sum = 0;
	for (uint32_t c = 0; c < ADMIT_SPREAD; ++c) sum += st.spread[(size_t)c * 258 + threadIdx.x];
	if (!sum) return;
	if (threadIdx.x < 256) atomicAdd(&st.hist_dense[threadIdx.x], sum);
	else atomicAdd(&st.counters[threadIdx.x == 256 ? st.c_retained : st.c_insertions], sum);
}

// One workgroup per table range (at most SPAN_MAX slots): fill it in LDS, write it out whole.  range_begin[r] ..
// range_begin[r+1] are the entries whose slot hash has r in its top `bits` bits; their home slots lie in
// [lo(r), lo(r+1)] with lo(r) = umulhi(r << (64 - bits), cap) -- the slot lo(r+1) itself belongs to the next range (a slot
// that the boundary between two ranges cuts).  An entry that finds no free slot between its home and the end of the range
// goes to the overflow list (ovf, at most ovf_cap entries; beyond that TC_OVF_LOST counts and the host rebuilds the classic
// way).
static constexpr uint32_t RANGE_SPAN_MAX = 4096;
template <int W>
__global__ __launch_bounds__(1024) void k_build_ranges(const Ent<W> *__restrict__ entries, const unsigned long long *__restrict__ range_begin,
                                                       uint32_t nranges, int bits, Slot *__restrict__ tab, uint64_t cap, Slot *__restrict__ ovf,
                                                       uint64_t ovf_cap, unsigned long long *__restrict__ counters) {
	extern __shared__ Slot s_tab[];  // the largest span of this table: (cap >> bits) + 1 slots (two workgroups per CU up to 2560)
	uint32_t placed = 0;
	for (uint32_t r = blockIdx.x; r < nranges; r += gridDim.x) {
		const uint64_t lo = slot_home((uint64_t)r << (64 - bits), cap);
		const uint64_t hi = (r + 1 == nranges) ? cap : slot_home((uint64_t)(r + 1) << (64 - bits), cap);
		const uint32_t span = (uint32_t)(hi - lo);  // <= RANGE_SPAN_MAX: the host picks `bits` for that
		for (uint32_t i = threadIdx.x; i < span; i += blockDim.x) {
			*reinterpret_cast<ulonglong2 *>(&s_tab[i].k0) = make_ulonglong2(EMPTY, EMPTY);
			*reinterpret_cast<ulonglong2 *>(&s_tab[i].stamp) = make_ulonglong2(INF_STAMP, 0);
		}
		__syncthreads();
		const uint64_t begin = range_begin[r], end = range_begin[r + 1];
		for (uint64_t e = begin + threadIdx.x; e < end; e += blockDim.x) {
			const Slot v = ent_load(&entries[e]);
			const ulonglong2 a = make_ulonglong2(v.k0, v.k1), b = make_ulonglong2(v.stamp, v.meta);
			const uint64_t home = slot_home(slot_hash<W>(a.x, W == 2 ? a.y : 0), cap);
			uint32_t s = (uint32_t)(home - lo);  // the partition guarantees lo <= home <= hi
			bool done = false;
			for (; s < span; ++s) {
				// the keys are distinct: whoever sets word 0 owns the slot, nobody else looks at the rest of it
				if (s_tab[s].k0 == EMPTY && atomicCAS(reinterpret_cast<unsigned long long *>(&s_tab[s].k0), EMPTY, a.x) == EMPTY) {
					s_tab[s].k1 = (W == 2) ? a.y : 0;
					*reinterpret_cast<ulonglong2 *>(&s_tab[s].stamp) = b;
					done = true;
					break;
				}
			}
			if (done) {
				placed++;
			} else {
				const unsigned long long pos = atomicAdd(&counters[TC_OVF_CURSOR], 1ull);
				if (pos < ovf_cap) {
					*reinterpret_cast<ulonglong2 *>(&ovf[pos].k0) = a;
					*reinterpret_cast<ulonglong2 *>(&ovf[pos].stamp) = b;
				} else {
					atomicAdd(&counters[TC_OVF_LOST], 1ull);
				}
			}
		}
		__syncthreads();
		// 16 bytes per thread and step
		const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(s_tab);
		ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(tab + lo);
		for (uint32_t i = threadIdx.x; i < 2 * span; i += blockDim.x) dst[i] = src[i];
		__syncthreads();
	}
	for (int d = 32; d > 0; d >>= 1) placed += __shfl_down(placed, d);
	if ((threadIdx.x & 63) == 0 && placed) atomicAdd(&counters[0 /*C_FILL*/], (unsigned long long)placed);
}

}  // namespace rg
