// small host<->device copies, pageable against pinned host memory, as the library's per-level bookkeeping issues them:
// time per (copy + stream synchronize).  hipcc --offload-arch=gfx950 -O2 -o microbench_copy microbench_copy.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k_touch(unsigned long long *p) { p[threadIdx.x] += 1; }
int main() {
	hipStream_t s;
	hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	unsigned long long *d = nullptr;
	hipMalloc(&d, 1 << 20);
	void *pinned = nullptr;
	hipHostMalloc(&pinned, 1 << 20);
	std::vector<char> pageable(1 << 20);
	const size_t sizes[] = {64, 2048, 131072, 524288};
	for (size_t bytes : sizes)
		for (int dir = 0; dir < 2; ++dir)
			for (int pin = 0; pin < 2; ++pin) {
				void *hp = pin ? pinned : (void *)pageable.data();
				const int N = 200;
				double best = 1e9;
				for (int rep = 0; rep < 3; ++rep) {
					auto t0 = std::chrono::steady_clock::now();
					for (int i = 0; i < N; ++i) {
						hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, s, d);
						if (dir == 0) hipMemcpyAsync(hp, d, bytes, hipMemcpyDeviceToHost, s);
						else hipMemcpyAsync(d, hp, bytes, hipMemcpyHostToDevice, s);
						hipStreamSynchronize(s);
					}
					auto t1 = std::chrono::steady_clock::now();
					best = std::min(best, std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
				}
				printf("%7zu bytes %s %-8s %6.1f us per kernel + copy + sync\n", bytes, dir ? "H2D" : "D2H", pin ? "pinned" : "pageable", best);
			}
	return 0;
}
