// probe: what the first use of a queue costs in a fresh process, per way of getting one
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
static double now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
__global__ void k(int *p) { if (threadIdx.x == 0) *p = 1; }
int main(int argc, char **argv)
{
	int mode = argc > 1 ? atoi(argv[1]) : 0;
	double t0 = now(), t00 = t0;
	hipSetDevice(0); hipFree(0);
	printf("mode %d: init %.3f s\n", mode, now() - t0);
	int *d; hipMalloc(&d, 4);
	hipStream_t st = 0;
	t0 = now();
	if (mode == 1) hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
	else if (mode == 2) st = hipStreamPerThread;
	else if (mode == 3) hipStreamCreate(&st);
	printf("  get stream %.2f ms\n", (now() - t0) * 1e3);
	t0 = now(); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, st, d); hipStreamSynchronize(st); printf("  first kernel %.2f ms\n", (now() - t0) * 1e3);
	t0 = now(); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, st, d); hipStreamSynchronize(st); printf("  second kernel %.3f ms\n", (now() - t0) * 1e3);
	void *h; hipHostMalloc(&h, 1 << 20, 0);
	t0 = now(); hipMemcpyAsync(d, h, 4, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); printf("  first copy %.2f ms\n", (now() - t0) * 1e3);
	printf("  total %.3f s\n", now() - t00);
	return 0;
}
