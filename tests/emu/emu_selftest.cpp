// tests/emu/emu_selftest.cpp -- checks the CPU stand-in for the HIP runtime against the documented semantics of the
// operations the kernels use (wave64 shuffles, ballot, DPP controls, barriers, early exits, atomics).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

extern "C" void emu_stats(uint64_t *, uint64_t *, uint64_t *);
static int g_fail = 0;
#define EXPECT(c) do { if (!(c)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++g_fail; } } while (0)

__global__ void k_shfl(uint32_t *out)
{
	unsigned t = threadIdx.x, lane = t & 63;
	uint32_t v = t * 3 + 1;
	out[t * 8 + 0] = __shfl(v, 5, 64);
	out[t * 8 + 1] = __shfl_xor(v, 16, 64);
	out[t * 8 + 2] = __shfl_up(v, 3, 64);
	out[t * 8 + 3] = __shfl_down(v, 7, 64);
	uint64_t b = __ballot(lane % 3 == 0);
	out[t * 8 + 4] = (uint32_t)b;
	out[t * 8 + 5] = (uint32_t)(b >> 32);
	out[t * 8 + 6] = __shfl(v, (int)(lane & 7), 8);
	out[t * 8 + 7] = __builtin_amdgcn_readfirstlane(v);
}

__global__ void k_dpp(uint32_t *out)
{
	unsigned t = threadIdx.x;
	uint32_t v = t + 100;
	out[t * 8 + 0] = (uint32_t)__builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true);
	out[t * 8 + 1] = (uint32_t)__builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true);
	out[t * 8 + 2] = (uint32_t)__builtin_amdgcn_mov_dpp(v, 0x1B, 0xf, 0xf, true);
	out[t * 8 + 3] = (uint32_t)__builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true);
	out[t * 8 + 4] = (uint32_t)__builtin_amdgcn_mov_dpp(v, 0x140, 0xf, 0xf, true);
	out[t * 8 + 5] = (uint32_t)__builtin_amdgcn_mov_dpp(v, 0x128, 0xf, 0xf, true);
	int x = __builtin_amdgcn_mov_dpp(v, 0x104, 0xf, 0x5, true);
	out[t * 8 + 6] = (uint32_t)__builtin_amdgcn_update_dpp(x, v, 0x114, 0xf, 0xa, false);
	out[t * 8 + 7] = (uint32_t)__builtin_amdgcn_mov_dpp(v, 0x111, 0xf, 0xf, true); // row_shr:1, 0 at the row start
}

__global__ void k_perm(uint32_t *out)
{
	unsigned lane = threadIdx.x & 63;
	int v = (int)(lane * 7 + 3);
	out[lane * 4 + 0] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane + 5) & 63) << 2), v);
	out[lane * 4 + 1] = (uint32_t)__builtin_amdgcn_ds_permute((int)(((lane * 3) & 63) << 2), v); // lane*3 mod 64 is a bijection
	out[lane * 4 + 2] = (uint32_t)__builtin_amdgcn_readlane(v, 17);
	out[lane * 4 + 3] = __builtin_amdgcn_mbcnt_hi(0xF0F0F0F0u, __builtin_amdgcn_mbcnt_lo(0x0F0F0F0Fu, 0));
}

// the gfx9 wave64 inclusive scan out of DPP row shifts and row broadcasts (what LLVM's atomic optimizer emits), a wave_shr:1 and a readlane
__global__ void k_dpp_scan(int *out)
{
	unsigned lane = threadIdx.x & 63;
	int x = (int)(lane * lane % 11) - 3;
	x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
	x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
	x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
	x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
	x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
	x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
	out[lane * 3 + 0] = x;
	out[lane * 3 + 1] = __builtin_amdgcn_update_dpp(-7, x, 0x138, 0xf, 0xf, false);
	out[lane * 3 + 2] = __builtin_amdgcn_readlane(x, 63);
}

// block reduction through LDS with barriers; threads beyond n leave early; one atomic per block
__global__ void k_reduce(const uint32_t *in, size_t n, unsigned long long *total, uint32_t *maxv)
{
	__shared__ uint32_t s[256];
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	s[threadIdx.x] = i < n ? in[i] : 0;
	__syncthreads();
	for (unsigned o = 128; o > 0; o >>= 1) {
		if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
		__syncthreads();
	}
	if (threadIdx.x == 0) atomicAdd(total, (unsigned long long)s[0]);
	if (i >= n) return;
	atomicMax(maxv, in[i]);
}

// a loop whose trip count differs per lane, with a ballot inside and a wave reduction behind it
__global__ void k_diverge(uint32_t *out)
{
	unsigned lane = threadIdx.x & 63;
	uint32_t cnt = 0;
	for (unsigned k = 0; k < lane % 5; ++k) cnt += (uint32_t)__popcll(__ballot(1));
	uint32_t sum = cnt;
	for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
	out[threadIdx.x] = cnt;
	out[64 + threadIdx.x] = sum;
	if (lane & 1) return;
	out[128 + threadIdx.x] = (uint32_t)__popcll(__ballot(1));
}

int main()
{
	uint32_t *d;
	hipMalloc(&d, 1 << 20);
	{
		hipLaunchKernelGGL(k_shfl, dim3(1), dim3(128), 0, nullptr, d);
		for (unsigned t = 0; t < 128; ++t) {
			unsigned lane = t & 63, base = t & ~63u;
			auto V = [](unsigned x) { return x * 3 + 1; };
			EXPECT(d[t * 8 + 0] == V(base + 5));
			EXPECT(d[t * 8 + 1] == V(t ^ 16));
			EXPECT(d[t * 8 + 2] == (lane >= 3 ? V(t - 3) : V(t)));
			EXPECT(d[t * 8 + 3] == (lane + 7 < 64 ? V(t + 7) : V(t)));
			uint64_t b = 0;
			for (unsigned l = 0; l < 64; ++l) if (l % 3 == 0) b |= 1ull << l;
			EXPECT(d[t * 8 + 4] == (uint32_t)b && d[t * 8 + 5] == (uint32_t)(b >> 32));
			EXPECT(d[t * 8 + 6] == V((t & ~7u) + (lane & 7)));
			EXPECT(d[t * 8 + 7] == V(base));
		}
	}
	{
		hipLaunchKernelGGL(k_dpp, dim3(1), dim3(64), 0, nullptr, d);
		for (unsigned t = 0; t < 64; ++t) {
			EXPECT(d[t * 8 + 0] == (t ^ 1) + 100);
			EXPECT(d[t * 8 + 1] == (t ^ 2) + 100);
			EXPECT(d[t * 8 + 2] == (t ^ 3) + 100);
			EXPECT(d[t * 8 + 3] == (t ^ 7) + 100);
			EXPECT(d[t * 8 + 4] == (t ^ 15) + 100);
			EXPECT(d[t * 8 + 5] == (t ^ 8) + 100);
			EXPECT(d[t * 8 + 6] == (t ^ 4) + 100);
			EXPECT(d[t * 8 + 7] == ((t & 15) ? t - 1 + 100 : 0));
		}
	}
	{
		const size_t n = 100000;
		std::vector<uint32_t> h(n);
		unsigned long long want = 0;
		uint32_t wmax = 0;
		for (size_t i = 0; i < n; ++i) { h[i] = (uint32_t)(i * 2654435761u) >> 12; want += h[i]; if (h[i] > wmax) wmax = h[i]; }
		uint32_t *din;
		unsigned long long *dt;
		hipMalloc(&din, n * 4);
		hipMalloc(&dt, 16);
		hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
		hipMemset(dt, 0, 16);
		hipLaunchKernelGGL(k_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, din, n, dt, (uint32_t *)(dt + 1));
		EXPECT(dt[0] == want);
		EXPECT(*(uint32_t *)(dt + 1) == wmax);
		hipFree(din);
		hipFree(dt);
	}
	{
		hipLaunchKernelGGL(k_diverge, dim3(1), dim3(64), 0, nullptr, d);
		uint32_t total = 0;
		for (unsigned l = 0; l < 64; ++l) {
			// in round k the lanes with l%5 > k are still in the loop
			uint32_t c = 0;
			for (unsigned k = 0; k < l % 5; ++k) { unsigned act = 0; for (unsigned m = 0; m < 64; ++m) act += m % 5 > k; c += act; }
			EXPECT(d[l] == c);
			total += c;
		}
		for (unsigned l = 0; l < 64; ++l) {
			EXPECT(d[64 + l] == total);
			if (!(l & 1)) EXPECT(d[128 + l] == 32);
		}
	}
	{
		hipLaunchKernelGGL(k_perm, dim3(1), dim3(64), 0, nullptr, d);
		for (unsigned l = 0; l < 64; ++l) {
			EXPECT(d[l * 4 + 0] == ((l + 5) & 63) * 7 + 3);
			unsigned src = 0;
			for (unsigned m = 0; m < 64; ++m) if (((m * 3) & 63) == l) src = m;
			EXPECT(d[l * 4 + 1] == src * 7 + 3);
			EXPECT(d[l * 4 + 2] == 17 * 7 + 3);
			const uint64_t mask = (uint64_t)0xF0F0F0F0u << 32 | 0x0F0F0F0Fu;
			EXPECT(d[l * 4 + 3] == (uint32_t)__builtin_popcountll(mask & ((1ull << l) - 1ull)));
		}
	}
	{
		hipLaunchKernelGGL(k_dpp_scan, dim3(1), dim3(64), 0, nullptr, (int*)d);
		const int *r = (const int*)d;
		int run = 0, prev = -7;
		for (unsigned l = 0; l < 64; ++l) {
			run += (int)(l * l % 11) - 3;
			EXPECT(r[l * 3 + 0] == run);
			EXPECT(r[l * 3 + 1] == prev);
			prev = run;
		}
		for (unsigned l = 0; l < 64; ++l) EXPECT(r[l * 3 + 2] == run);
	}
	hipFree(d);
	uint64_t a, b, c;
	emu_stats(&a, &b, &c);
	printf("%s: %d failures; %llu launches, %llu cross-lane operations (%llu on part of a wave)\n", g_fail ? "FAILED" : "OK", g_fail,
		(unsigned long long)a, (unsigned long long)b, (unsigned long long)c);
	return g_fail != 0;
}
