// rg_synth.hip -- bench / test tooling, not part of the product: the synthetic reads of SURVEY.md 8(d) (i.i.d. uniform
// genome, uniform start, random strand, i.i.d. substitution errors, no N) from a COUNTER-BASED generator, so that the
// same reads can be produced on the device (here, for bench.py and the GPU tests) and on the host (oracle/ro_synth.c,
// for the oracle) without moving gigabytes between them.  Every value is a pure function of (seed, index):
//   mix            = splitmix64's finaliser
//   genome base j  = mix(mix(seed ^ "genome") + j) >> 62
//   read r         : a = mix(mix(seed ^ "reads") + r); start = mulhi(a, G - L + 1); reverse strand iff mix(a ^ 0x5555) & 1
//   base i of r    = genome base start + i, or 3 - genome base start + L - 1 - i on the reverse strand
//   error at (r,i) : u = mix(mix(seed ^ "error") + r * L + i); if (u >> 40) < rate * 2^24: base += 1 + ((u & 0xFFFFFF) * 3 >> 24) (mod 4)
// Layout: Ray's packed reads (SequencesLoader/Read.cpp:135-141), every read on a byte boundary, (L + 3) / 4 bytes each.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RGS_GENOME 0x67656E6F6D65ull
#define RGS_READS 0x7265616473ull
#define RGS_ERROR 0x6572726F72ull

__host__ __device__ static inline uint64_t rgs_mix(uint64_t x) {
	x += 0x9E3779B97F4A7C15ull;
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
	return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void k_rgs_generate(uint8_t *__restrict__ packed, unsigned long long *__restrict__ offsets, uint32_t *__restrict__ lengths,
                                                      uint64_t n, uint64_t r0, uint64_t genome, uint32_t L, uint64_t seed, uint64_t err_thr) {
	const uint32_t bpr = (L + 3) / 4;
	const uint64_t kg = rgs_mix(seed ^ RGS_GENOME), kr = rgs_mix(seed ^ RGS_READS), ke = rgs_mix(seed ^ RGS_ERROR);
	const uint64_t total = n * bpr;
	for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < total; b += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t i = b / bpr;
		const uint32_t j = (uint32_t)(b - i * bpr);
		const uint64_t r = r0 + i;
		const uint64_t a = rgs_mix(kr + r);
		const uint64_t start = __umul64hi(a, genome - L + 1);
		const bool rev = (rgs_mix(a ^ 0x5555ull) & 1ull) != 0;
		uint32_t v = 0;
		for (uint32_t q = 0; q < 4; ++q) {
			const uint32_t p = 4 * j + q;
			if (p >= L) break;
			uint32_t c = (uint32_t)(rgs_mix(kg + (rev ? start + (L - 1 - p) : start + p)) >> 62);
			if (rev) c = 3 - c;
			if (err_thr) {
				const uint64_t u = rgs_mix(ke + r * L + p);
				if ((u >> 40) < err_thr) c = (c + 1 + (uint32_t)(((u & 0xFFFFFFull) * 3) >> 24)) & 3u;
			}
			v |= c << (2 * q);
		}
		packed[b] = (uint8_t)v;
		if (j == 0) {
			offsets[i] = i * bpr;
			lengths[i] = L;
		}
	}
}

extern "C" int rgs_generate(void *d_packed, void *d_offsets, void *d_lengths, uint64_t n, uint64_t r0, uint64_t genome, uint32_t L, uint64_t seed,
                            double error, void *stream) {
	if (n == 0) return 0;
	if (!d_packed || !d_offsets || !d_lengths || L == 0 || genome < L || error < 0.0 || error >= 1.0) return -1;
	const uint64_t thr = (uint64_t)(error * 16777216.0);
	const uint64_t total = n * ((L + 3) / 4);
	const unsigned grid = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
	hipLaunchKernelGGL(k_rgs_generate, dim3(grid), dim3(256), 0, (hipStream_t)stream, (uint8_t *)d_packed, (unsigned long long *)d_offsets, (uint32_t *)d_lengths, n, r0,
	                   genome, L, seed, thr);
	if (hipGetLastError() != hipSuccess) return -2;
	return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? 0 : -2;
}

// A stand-in for the host consumer of the store hand-back (Ray's GridTable::insert + Vertex::load loop): touches every
// byte of every record.  user -> {checksum, records seen, bytes per record}.  Type: rg_export_fn (include/raygraph.h).
struct rgs_consumer {
	uint64_t checksum, records, record_bytes;
};
extern "C" int rgs_touch_records(void *user, const void *records, uint64_t n) {
	rgs_consumer *c = (rgs_consumer *)user;
	const unsigned char *p = (const unsigned char *)records;
	const uint64_t bytes = n * c->record_bytes;
	uint64_t sum = c->checksum, i = 0;
	for (; i + 8 <= bytes; i += 8) {
		uint64_t w;
		__builtin_memcpy(&w, p + i, 8);
		sum = (sum ^ w) * 0x9E3779B97F4A7C15ull;
	}
	for (; i < bytes; ++i) sum = (sum ^ p[i]) * 0x9E3779B97F4A7C15ull;
	c->checksum = sum;
	c->records += n;
	return 0;
}
