// Development microbenchmark: do 20-byte records scattered over NS open runs per wavefront merge into full lines before
// they reach HBM?  (What folding the first partition level into the record kernel would rely on.)
// Every wavefront owns NS runs; each trip a lane makes one record and appends it to run hash(trip, lane) % NS through a
// cursor in LDS.  `spin` dependent multiplications per trip stretch the kernel to the record kernel's pace.
// usage: microbench_streams [records_per_wave] [spin]; prints ms and GB/s for NS = 1, 4, 16, 32, 64, 128.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// SHARED: the four wavefronts of a workgroup share their runs (a quarter of the open runs, four times as long); SIDE: one byte
// per record goes to a parallel array as well (the next level's digit)
template <int NS, bool SHARED, bool SIDE, bool SORTED = false>
__global__ __launch_bounds__(256) void k_streams(uint32_t *__restrict__ out, uint8_t *__restrict__ side, uint32_t per_wave, uint32_t run_cap, int spin, uint32_t *sink) {
	__shared__ uint32_t cur[4][NS];
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	for (uint32_t i = lane; i < NS; i += 64) cur[wave][i] = 0;
	__syncthreads();
	const uint64_t gw = SHARED ? blockIdx.x : (uint64_t)blockIdx.x * 4 + wave;
	const uint32_t cap = SHARED ? 4 * run_cap : run_cap;
	const uint64_t base = gw * (uint64_t)NS * cap;
	uint32_t *cursors = cur[SHARED ? 0 : wave];
	uint32_t x = ((uint32_t)blockIdx.x * 4 + wave) * 2654435761u + lane;
	for (uint32_t t = 0; t < per_wave; t += 64) {
		for (int s = 0; s < spin; ++s) x = x * 1664525u + 1013904223u;
		// SORTED: the lanes of a trip hold their digits in order (neighbouring lanes write neighbouring records of a run)
		const uint32_t d = SORTED ? ((lane * NS) / 64 + t / 64) % NS : (x >> 13) % NS;
		const uint32_t pos = atomicAdd(&cursors[d], 1u);
		if (pos < cap) {
			const uint64_t at = base + (uint64_t)d * cap + pos;
			uint32_t *p = out + at * 5;
			p[0] = x; p[1] = t; p[2] = lane; p[3] = d; p[4] = pos;
			if (SIDE) side[at] = (uint8_t)(x >> 3);
		}
		x += 0x9E3779B9u;
	}
	if (x == 0x12345678u) *sink = x;
}

template <int NS, bool SHARED = false, bool SIDE = false, bool SORTED = false>
static void run(uint32_t *out, uint8_t *side, uint64_t cap_words, uint32_t per_wave, int spin, uint32_t *sink) {
	const int grid = 1024;  // 4 workgroups per CU, as the record kernel
	const uint32_t run_cap = (uint32_t)(per_wave / NS + per_wave / NS / 4 + 64);
	if ((uint64_t)grid * 4 * NS * run_cap * 5 > cap_words) { printf("NS %d: buffer too small\n", NS); return; }
	hipEvent_t a, b;
	CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
	for (int rep = 0; rep < 3; ++rep) {
		CHECK(hipEventRecord(a));
		hipLaunchKernelGGL((k_streams<NS, SHARED, SIDE, SORTED>), dim3(grid), dim3(256), 0, 0, out, side, per_wave, run_cap, spin, sink);
		CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
		float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
		if (rep == 2) printf("NS %3d%s%s%s: %7.3f ms, %7.1f GB/s of records\n", NS, SHARED ? " shared by the workgroup" : "", SIDE ? " + side byte" : "", SORTED ? ", digits in lane order" : "", ms, (double)grid * 4 * per_wave * 20 / ms / 1e6);
	}
}

int main(int argc, char **argv) {
	const uint32_t per_wave = argc > 1 ? (uint32_t)atol(argv[1]) : 139264;  // C2: 5.69e8 records over 4096 wavefronts
	const int spin = argc > 2 ? atoi(argv[2]) : 0;
	const uint64_t cap_words = (uint64_t)4096 * per_wave * 5 * 2;
	uint32_t *out, *sink;
	uint8_t *side;
	CHECK(hipMalloc(&out, cap_words * 4)); CHECK(hipMalloc(&sink, 4)); CHECK(hipMalloc(&side, cap_words / 5));
	CHECK(hipMemset(out, 0, cap_words * 4));
	printf("records per wavefront %u, spin %d\n", per_wave, spin);
	run<1>(out, side, cap_words, per_wave, spin, sink); run<4>(out, side, cap_words, per_wave, spin, sink); run<16>(out, side, cap_words, per_wave, spin, sink);
	run<32>(out, side, cap_words, per_wave, spin, sink); run<64>(out, side, cap_words, per_wave, spin, sink); run<128>(out, side, cap_words, per_wave, spin, sink);
	run<32, true, false>(out, side, cap_words, per_wave, spin, sink); run<32, true, true>(out, side, cap_words, per_wave, spin, sink);
	run<64, true, true>(out, side, cap_words, per_wave, spin, sink); run<128, true, true>(out, side, cap_words, per_wave, spin, sink);
	run<1, true, true>(out, side, cap_words, per_wave, spin, sink);
	run<32, true, true, true>(out, side, cap_words, per_wave, spin, sink); run<64, true, true, true>(out, side, cap_words, per_wave, spin, sink);
	return 0;
}
