// alloc_probe.hip -- what a big device allocation costs on this box, by API and by size (some boxes of the pool clear VRAM at ~8 GB/s inside hipMalloc: 3.7 s for the
// 30 GB text buffer of BASELINE configs[4]; others return at once).   hipcc --offload-arch=gfx950 -O2 -o alloc_probe alloc_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
static double now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
__global__ void k_touch(unsigned char *p, size_t n) { size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4096; if (i < n) p[i] = 1; }
int main(int argc, char **argv)
{
	const size_t GB = (size_t)1 << 30;
	size_t big = (argc > 1 ? (size_t)atol(argv[1]) : 30) * GB;
	hipStream_t st; hipStreamCreate(&st);
	void *p = 0; double t0, t1, t2;
	hipFree(0);
	for (int rep = 0; rep < 2; ++rep) {
		t0 = now(); hipError_t e = hipMalloc(&p, big); t1 = now();
		hipLaunchKernelGGL(k_touch, dim3((unsigned)((big / 4096 + 255) / 256)), dim3(256), 0, st, (unsigned char*)p, big); hipStreamSynchronize(st); t2 = now();
		printf("hipMalloc %zu GB (rep %d): %s %.3f s, first touch %.3f s\n", big / GB, rep, hipGetErrorString(e), t1 - t0, t2 - t1);
		t0 = now(); hipFree(p); t1 = now(); printf("  hipFree %.3f s\n", t1 - t0);
	}
	{ // the same bytes in 1 GB pieces
		const int n = (int)(big / GB); void **q = (void**)calloc(n, sizeof(void*));
		t0 = now(); for (int i = 0; i < n; ++i) hipMalloc(&q[i], GB); t1 = now();
		printf("hipMalloc %d x 1 GB: %.3f s\n", n, t1 - t0);
		t0 = now(); for (int i = 0; i < n; ++i) hipFree(q[i]); t1 = now(); printf("  hipFree %.3f s\n", t1 - t0);
		free(q);
	}
	{ // stream-ordered pool
		t0 = now(); hipError_t e = hipMallocAsync(&p, big, st); hipStreamSynchronize(st); t1 = now();
		printf("hipMallocAsync %zu GB: %s %.3f s\n", big / GB, hipGetErrorString(e), t1 - t0);
		if (e == hipSuccess) { t0 = now(); hipFreeAsync(p, st); hipStreamSynchronize(st); t1 = now(); printf("  hipFreeAsync %.3f s\n", t1 - t0);
			t0 = now(); e = hipMallocAsync(&p, big, st); hipStreamSynchronize(st); t1 = now(); printf("hipMallocAsync again: %s %.3f s\n", hipGetErrorString(e), t1 - t0); if (e == hipSuccess) { hipFreeAsync(p, st); hipStreamSynchronize(st); } }
	}
	{ // uncached / fine-grained flavours
		t0 = now(); hipError_t e = hipExtMallocWithFlags(&p, big, hipDeviceMallocUncached); t1 = now();
		printf("hipExtMallocWithFlags(uncached) %zu GB: %s %.3f s\n", big / GB, hipGetErrorString(e), t1 - t0);
		if (e == hipSuccess) hipFree(p);
	}
	t0 = now(); hipError_t e = hipMalloc(&p, big); t1 = now();
	printf("hipMalloc %zu GB (after all that): %s %.3f s\n", big / GB, hipGetErrorString(e), t1 - t0);
	return 0;
}
