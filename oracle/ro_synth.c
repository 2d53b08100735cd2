/* ro_synth.c -- TEST INFRASTRUCTURE: the host twin of tools/rg_synth.hip (the counter-based generator of the synthetic
 * reads of SURVEY.md 8d; see that file for the definition).  Produces the reads the oracle runs on, bit for bit the ones
 * the device generator gives bench.py and the GPU tests. */
#include <stdint.h>
#include <string.h>

#define RGS_GENOME 0x67656E6F6D65ull
#define RGS_READS 0x7265616473ull
#define RGS_ERROR 0x6572726F72ull

static inline uint64_t rgs_mix(uint64_t x) {
	x += 0x9E3779B97F4A7C15ull;
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
	return x ^ (x >> 31);
}

/* reads r0 .. r0+n-1; packed holds n * ((L+3)/4) bytes */
int ro_synth_generate(uint8_t *packed, uint64_t *offsets, uint32_t *lengths, uint64_t n, uint64_t r0, uint64_t genome, uint32_t L, uint64_t seed,
                      double error) {
	if (L == 0 || genome < L || error < 0.0 || error >= 1.0) return -1;
	const uint32_t bpr = (L + 3) / 4;
	const uint64_t kg = rgs_mix(seed ^ RGS_GENOME), kr = rgs_mix(seed ^ RGS_READS), ke = rgs_mix(seed ^ RGS_ERROR);
	const uint64_t thr = (uint64_t)(error * 16777216.0);
	for (uint64_t i = 0; i < n; i++) {
		const uint64_t r = r0 + i;
		const uint64_t a = rgs_mix(kr + r);
		const uint64_t start = (uint64_t)(((unsigned __int128)a * (genome - L + 1)) >> 64);
		const int rev = (int)(rgs_mix(a ^ 0x5555ull) & 1ull);
		uint8_t *out = packed + i * bpr;
		memset(out, 0, bpr);
		for (uint32_t p = 0; p < L; p++) {
			uint32_t c = (uint32_t)(rgs_mix(kg + (rev ? start + (L - 1 - p) : start + p)) >> 62);
			if (rev) c = 3 - c;
			if (thr) {
				const uint64_t u = rgs_mix(ke + r * L + p);
				if ((u >> 40) < thr) c = (c + 1 + (uint32_t)(((u & 0xFFFFFFull) * 3) >> 24)) & 3u;
			}
			out[p / 4] |= (uint8_t)(c << (2 * (p % 4)));
		}
		offsets[i] = i * bpr;
		lengths[i] = L;
	}
	return 0;
}
