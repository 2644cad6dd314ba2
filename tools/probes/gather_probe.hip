// probe: what a RANDOM gather of 32-byte records (two dwordx4 per lane, the access pattern of k_hit_sub<gather>) can reach on this chip, and what
// rocprofv3's FETCH_SIZE reports for it -- the guide's x2 correction is calibrated for wide coalesced streams only.
//   hipcc --offload-arch=gfx950 -O2 -o gather_probe gather_probe.hip
//   ./gather_probe                                   # times: stream copy, random gather (records in a 6.4 GB / 256 MB / 16 MB window), pair gather
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- ./gather_probe   # counters per kernel against the known byte counts printed here
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) { z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }

// every lane reads ITEMS records at pseudo-random positions of [0, n_rec) (window: positions are confined to blocks of `win` records that advance with the
// thread index, i.e. win = n_rec is fully random, a small win models locality), sums them and writes 4 bytes
template <int ITEMS>
__global__ __launch_bounds__(256) void k_gather32(const uint4 *__restrict__ rec, uint64_t n_rec, uint64_t win, uint64_t n_out, uint32_t *__restrict__ out, uint64_t salt)
{
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_out; i += (uint64_t)gridDim.x * 256) {
		uint32_t acc = 0;
		uint4 a[ITEMS], b[ITEMS];
#pragma unroll
		for (int k = 0; k < ITEMS; ++k) {
			const uint64_t base = win >= n_rec ? 0 : ((i * ITEMS / win) * win) % (n_rec - win);
			const uint64_t j = base + mix((i * ITEMS + k) ^ salt) % win;
			a[k] = rec[2 * j]; b[k] = rec[2 * j + 1];
		}
#pragma unroll
		for (int k = 0; k < ITEMS; ++k) acc += a[k].x + a[k].w + b[k].y + b[k].w;
		out[i] = acc;
	}
}
__global__ __launch_bounds__(256) void k_stream(const uint4 *__restrict__ in, uint4 *__restrict__ out, uint64_t n)
{
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) out[i] = in[i];
}
__global__ __launch_bounds__(256) void k_read(const uint4 *__restrict__ in, uint32_t *__restrict__ out, uint64_t n)
{
	uint32_t acc = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) { uint4 v = in[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
	if (acc == 0x12345678u) out[0] = acc;
}

int main()
{
	const uint64_t n_rec = 200000000ull, n_out = 100000000ull; // 6.4 GB of records; 1e8 lanes x ITEMS records each
	uint4 *rec, *cp; uint32_t *out;
	CK(hipMalloc(&rec, n_rec * 32)); CK(hipMalloc(&cp, n_rec * 32)); CK(hipMalloc(&out, n_out * 4));
	CK(hipMemset(rec, 1, n_rec * 32));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	float ms;
	for (int rep = 0; rep < 2; ++rep) {
		CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, rec, cp, n_rec * 2); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
		if (rep) printf("k_stream   copy 6.4 GB -> 6.4 GB                         %7.3f ms  %6.2f TB/s (read+write)\n", ms, 12.8e9 / ms / 1e9);
		CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, rec, out, n_rec * 2); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
		if (rep) printf("k_read     read 6.4 GB                                   %7.3f ms  %6.2f TB/s\n", ms, 6.4e9 / ms / 1e9);
	}
	const uint64_t wins[] = { n_rec, 8000000ull /* 256 MB */, 500000ull /* 16 MB */, 16384ull /* 512 KB */ };
	for (uint64_t win : wins) {
		for (int rep = 0; rep < 2; ++rep) {
			CK(hipEventRecord(e0)); hipLaunchKernelGGL((k_gather32<2>), dim3(8192), dim3(256), 0, 0, rec, n_rec, win, n_out, out, (uint64_t)rep); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
			if (rep) printf("k_gather32<2> 2e8 records of 32 B, window %9llu records   %7.3f ms  %6.2f G records/s  %6.2f TB/s of record bytes (+0.4 GB written)\n", (unsigned long long)win, ms, 2e8 / ms / 1e6, 6.4e9 / ms / 1e9);
		}
	}
	for (int rep = 0; rep < 2; ++rep) {
		CK(hipEventRecord(e0)); hipLaunchKernelGGL((k_gather32<4>), dim3(8192), dim3(256), 0, 0, rec, n_rec, n_rec, n_out / 2, out, (uint64_t)rep); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
		if (rep) printf("k_gather32<4> 2e8 records of 32 B, fully random, 4 per lane      %7.3f ms  %6.2f G records/s\n", ms, 2e8 / ms / 1e6);
	}
	return 0;
}
