// probe: cost of pinned allocations / stream + event creation / first DMA in a fresh process
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
static double now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main()
{
	double t0 = now();
	hipSetDevice(0); hipFree(0);
	printf("init %.3f s\n", now() - t0);
	size_t sizes[] = { 1u << 20, 4u << 20, 16u << 20, 64u << 20, 4u << 20, 4u << 20 };
	for (size_t s : sizes) { void *p; t0 = now(); hipHostMalloc(&p, s, hipHostMallocDefault); printf("hipHostMalloc %3zu MB %.2f ms\n", s >> 20, (now() - t0) * 1e3); }
	void *d; t0 = now(); hipMalloc(&d, 600u << 20); printf("hipMalloc 600 MB %.2f ms\n", (now() - t0) * 1e3);
	hipStream_t st; t0 = now(); hipStreamCreateWithFlags(&st, hipStreamNonBlocking); printf("stream create %.2f ms\n", (now() - t0) * 1e3);
	hipStream_t st2; t0 = now(); hipStreamCreateWithFlags(&st2, hipStreamNonBlocking); printf("stream create 2 %.2f ms\n", (now() - t0) * 1e3);
	hipEvent_t ev; t0 = now(); hipEventCreateWithFlags(&ev, hipEventDisableTiming); printf("event create %.2f ms\n", (now() - t0) * 1e3);
	void *h; hipHostMalloc(&h, 64u << 20, hipHostMallocDefault);
	for (int r = 0; r < 3; ++r) { t0 = now(); hipMemcpyAsync(d, h, 64u << 20, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); printf("DMA 64 MB rep %d: %.2f ms\n", r, (now() - t0) * 1e3); }
	for (int r = 0; r < 2; ++r) { t0 = now(); hipMemcpyAsync((char*)d + (300u << 20), h, 64u << 20, hipMemcpyHostToDevice, st2); hipStreamSynchronize(st2); printf("DMA 64 MB other stream/region rep %d: %.2f ms\n", r, (now() - t0) * 1e3); }
	void *m = malloc(64u << 20); for (size_t i = 0; i < (64u << 20); i += 4096) ((char*)m)[i] = 1;
	t0 = now(); hipHostRegister(m, 64u << 20, hipHostRegisterDefault); printf("hipHostRegister 64 MB %.2f ms\n", (now() - t0) * 1e3);
	return 0;
}
