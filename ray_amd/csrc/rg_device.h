// rg_device.h -- device-side building blocks of the gfx950 graph-construction path.
//
// Data layout in HBM (DESIGN.md "Data layout"):
//   * reads stay 2-bit packed exactly as Ray's Read stores them (SequencesLoader/Read.cpp:135-141):
//     base i of a read = bits [2(i%4), 2(i%4)+1] of byte i/4.  A k-mer word uses the same
//     little-endian 2-bit order (Mock/common_functions.cpp:189-207), so the forward k-mer at
//     position p is a plain 2k-bit funnel shift out of the packed read -- no per-base loop.
//   * the vertex store is one open-addressing table of 32-byte slots (one HBM sector each):
//       k0, k1   canonical ("lower") key in the table's internal form
//       stamp    arrival stamp of the key's first occurrence, (global read id << 16) | position
//       meta     bits 0..47 occurrence count (coverage after admission), bits 48..55 edge bitmap
//                gathered during pass 1 (pending), bits 56..63 published edge bitmap
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rg {

static constexpr uint64_t EMPTY = ~0ull;      // key word of a free slot
static constexpr uint64_t INF_STAMP = ~0ull;  // "never arrived"
static constexpr uint64_t COUNT_MASK = (1ull << 48) - 1;
static constexpr int PEND_SHIFT = 48;
// log entries only (never stored in a table): this key occurs in no other entry of the log it travels in
static constexpr uint64_t LOG_UNIQUE = 1ull << 47;
static constexpr int PUB_SHIFT = 56;
static constexpr uint32_t MAX_PROBE = 1u << 22;  // exit condition every lane reaches

struct __attribute__((aligned(32))) Slot {
	uint64_t k0, k1, stamp, meta;
};
static_assert(sizeof(Slot) == 32, "slot must be one 32-byte sector");

// final state of a slot after admission: coverage in the count field, the edge bits gathered during the sweep moved
// to the published field (rg_add_edges only makes them visible: the bits themselves are final once pass 1 ends)
__device__ __forceinline__ uint64_t admitted_meta(uint64_t meta, uint64_t cov) {
	const uint64_t pend = (meta >> PEND_SHIFT) & 0xFFull;
	return (meta & ~COUNT_MASK & ~(0xFFull << PUB_SHIFT)) | cov | (pend << PUB_SHIFT);
}

// ---- the reference's hashes, kept in one place (RayPlatform's crypto.cpp is absent,
// these are the published Thomas Wang mixes it is recalled to use -- parity unpinned) ------------
__host__ __device__ inline uint64_t mix1(uint64_t key) {  // uniform_hashing_function_1_64_64
	key = (~key) + (key << 21);
	key = key ^ (key >> 24);
	key = (key + (key << 3)) + (key << 8);
	key = key ^ (key >> 14);
	key = (key + (key << 2)) + (key << 4);
	key = key ^ (key >> 28);
	key = key + (key << 31);
	return key;
}
__host__ __device__ inline uint64_t mix2(uint64_t key) {  // uniform_hashing_function_2_64_64
	key = (~key) + (key << 18);
	key = key ^ (key >> 31);
	key = key * 21;
	key = key ^ (key >> 11);
	key = key + (key << 6);
	key = key ^ (key >> 22);
	return key;
}
// Kmer::hash_function_1 / _2 (KmerAcademyBuilder/Kmer.cpp:284-310): with two words, word 0 enters raw
// AND mixed
template <int W>
__host__ __device__ inline uint64_t hash1(uint64_t K0, uint64_t K1) {
	if (W == 1) return mix1(K0);
	return K0 ^ mix1(K0) ^ mix1(K1);
}
template <int W>
__host__ __device__ inline uint64_t hash2(uint64_t K0, uint64_t K1) {
	if (W == 1) return mix2(K0);
	return K0 ^ mix2(K0) ^ mix2(K1);
}

// BloomFilter.cpp:72-79
__device__ __constant__ const uint64_t BLOOM_MAGIC[8] = {
	0xe70b369c4c19f0f9ULL, 0xbeb7f38b993441a2ULL, 0x3149cd9246ca7995ULL, 0x2ef5b2c17d479ee8ULL,
	0xfb8daceab90fe233ULL, 0x20fa74e37d497859ULL, 0x9007d0caef749698ULL, 0x4e4100a5605ef967ULL};

// ---- internal key form ------------------------------------------------------------------------
// A free slot is marked by EMPTY in k0 (and k1).  A reference key can have all 64 bits of word 0
// set when k >= 32, so with two words the 126 key bits are re-split 63+63: both stored words then
// have a clear top bit and can never equal EMPTY, which lets a slot be claimed with two
// independent 64-bit compare-and-swaps and no lock.
template <int W>
__host__ __device__ inline void to_internal(uint64_t K0, uint64_t K1, uint64_t &i0, uint64_t &i1) {
	if (W == 1) {
		i0 = K0;
		i1 = 0;
	} else {
		i0 = K0 & ~(1ull << 63);
		i1 = (K1 << 1) | (K0 >> 63);
	}
}
template <int W>
__host__ __device__ inline void from_internal(uint64_t i0, uint64_t i1, uint64_t &K0, uint64_t &K1) {
	if (W == 1) {
		K0 = i0;
		K1 = 0;
	} else {
		K0 = i0 | (i1 << 63);
		K1 = i1 >> 1;
	}
}

template <int W>
__host__ __device__ inline uint64_t slot_hash(uint64_t i0, uint64_t i1) {
	uint64_t h = i0 * 0x9E3779B97F4A7C15ull;
	if (W == 2) h = (h ^ (h >> 32) ^ i1) * 0xC2B2AE3D27D4EB4Full;
	h ^= h >> 29;
	h *= 0xBF58476D1CE4E5B9ull;
	h ^= h >> 32;
	return h;
}

// home slot of a hash in a table of `cap` slots (any capacity, not only powers of two): the top bits decide, so
// keys in hash order are in slot order and a contiguous range of slots belongs to a contiguous range of hashes
// (the table is built range by range from entries partitioned by their hash, raygraph.hip: k_build_ranges)
__host__ __device__ inline uint64_t slot_home(uint64_t hash, uint64_t cap) {
#if defined(__HIP_DEVICE_COMPILE__)
	return __umul64hi(hash, cap);
#else
	return (uint64_t)(((unsigned __int128)hash * cap) >> 64);
#endif
}
__host__ __device__ inline uint64_t slot_next(uint64_t s, uint64_t cap) { return s + 1 == cap ? 0 : s + 1; }

// ---- k-mer arithmetic ----------------------------------------------------------------------------
// reverse the order of the 32 two-bit groups of a word
__device__ __forceinline__ uint64_t rev2(uint64_t x) {
	uint64_t y = __brevll(x);
	return ((y >> 1) & 0x5555555555555555ull) | ((y & 0x5555555555555555ull) << 1);
}

// Kmer::complementVertex (Kmer.cpp:398-420) without the O(k) loop
template <int W>
__device__ __forceinline__ void revcomp(uint64_t f0, uint64_t f1, int k, uint64_t &r0, uint64_t &r1) {
	if (W == 1) {
		r0 = (~rev2(f0)) >> (64 - 2 * k);
		r1 = 0;
	} else {
		uint64_t lo = ~rev2(f1), hi = ~rev2(f0);
		int s = 128 - 2 * k;  // 2..126
		if (s >= 64) {
			r0 = hi >> (s - 64);
			r1 = 0;
		} else {
			r0 = (lo >> s) | (hi << (64 - s));
			r1 = hi >> s;
		}
	}
}

// Kmer::operator< (Kmer.cpp:38-47)
template <int W>
__device__ __forceinline__ bool kmer_less(uint64_t a0, uint64_t a1, uint64_t b0, uint64_t b1) {
	if (W == 1) return a0 < b0;
	return a0 < b0 || (a0 == b0 && a1 < b1);
}

// 64 bits of the packed stream starting at an arbitrary bit address (P 8-byte aligned, padded)
__device__ __forceinline__ uint64_t ext64(const uint64_t *__restrict__ P, uint64_t bit) {
	uint64_t j = bit >> 6;
	uint32_t s = (uint32_t)bit & 63u;
	uint64_t lo = P[j], hi = P[j + 1];
	return s ? ((lo >> s) | (hi << (64 - s))) : lo;
}

__device__ __forceinline__ uint32_t base_at(const uint8_t *__restrict__ P8, uint64_t byte_off, uint32_t idx) {
	return (P8[byte_off + (idx >> 2)] >> ((idx & 3u) * 2u)) & 3u;
}

// Vertex::convertBitmap (Vertex.cpp:341-361): swap the nibbles, reverse the bits inside each nibble
__host__ __device__ inline uint32_t convert_bitmap(uint32_t m) {
	uint32_t s = ((m << 4) | (m >> 4)) & 0xFFu;
	// reverse bit order within each nibble
	s = ((s & 0xAAu) >> 1) | ((s & 0x55u) << 1);
	s = ((s & 0xCCu) >> 2) | ((s & 0x33u) << 2);
	return s;
}

// ---- table primitives ---------------------------------------------------------------------------
// Find the slot of key (i0,i1), claiming a free one if absent, then merge (stamp, count, edge bits)
// with atomics.  Every decision taken on a plain (possibly stale) load is safe: a key word only ever
// changes EMPTY -> key, stamps only fall, edge bits only rise; a stale EMPTY is settled by the CAS.
template <int W>
__device__ __forceinline__ bool upsert(Slot *__restrict__ tab, uint64_t cap, uint64_t i0, uint64_t i1,
                                       uint64_t stamp, uint64_t add, uint32_t emask, uint32_t &claimed, bool unique = false) {
	uint64_t s = slot_home(slot_hash<W>(i0, i1), cap);
	for (uint32_t probe = 0; probe < MAX_PROBE; ++probe) {
		Slot *p = tab + s;
		ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(&p->k0);
		ulonglong2 b = *reinterpret_cast<const ulonglong2 *>(&p->stamp);
		uint64_t c0 = a.x;
		bool fresh = false;  // this call made the slot this key's
		if (c0 == EMPTY) {
			uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long *>(&p->k0), EMPTY, i0);
			if (prev == EMPTY) {
				claimed++;
				c0 = i0;
				fresh = (W == 1);
			} else {
				c0 = prev;
			}
		}
		if (c0 == i0) {
			bool match = true;
			if (W == 2) {
				uint64_t c1 = a.y;
				if (c1 == EMPTY) {
					uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long *>(&p->k1), EMPTY, i1);
					c1 = (prev == EMPTY) ? i1 : prev;
					fresh = (prev == EMPTY);  // whoever sets word 1 owns the slot's key
				}
				match = (c1 == i1);
			}
			if (match) {
				if (unique && fresh) {
					// no other entry of this launch carries the key and the table did not hold it: nobody else
					// touches the payload, one plain 16-byte store replaces three atomics
					*reinterpret_cast<ulonglong2 *>(&p->stamp) = make_ulonglong2(stamp, add | ((uint64_t)emask << PEND_SHIFT));
					return true;
				}
				if (stamp < b.x) atomicMin(reinterpret_cast<unsigned long long *>(&p->stamp), stamp);
				uint32_t pend = (uint32_t)(b.y >> PEND_SHIFT) & 0xFFu;
				if (emask & ~pend) atomicOr(reinterpret_cast<unsigned long long *>(&p->meta), (uint64_t)emask << PEND_SHIFT);
				if (add) atomicAdd(reinterpret_cast<unsigned long long *>(&p->meta), add);
				return true;
			}
		}
		s = slot_next(s, cap);
	}
	return false;
}

// read-only probe, valid in a launch after the one that built the table
template <int W>
__device__ __forceinline__ const Slot *lookup(const Slot *__restrict__ tab, uint64_t cap, uint64_t i0, uint64_t i1) {
	uint64_t s = slot_home(slot_hash<W>(i0, i1), cap);
	for (uint32_t probe = 0; probe < MAX_PROBE; ++probe) {
		const Slot *p = tab + s;
		ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(&p->k0);
		if (a.x == EMPTY) return nullptr;
		if (a.x == i0 && (W == 1 || a.y == i1)) return p;
		s = slot_next(s, cap);
	}
	return nullptr;
}

}  // namespace rg
