// diag.hip -- MEASUREMENT HOOK, not part of the pipeline: plain access patterns with a KNOWN byte count, timed with HIP events on the context's stream.
//
// Two uses (tools/pmc_calibrate.py):
//   * what this GPU sustains for the access patterns the hot path is made of (a 16-byte stream, 8 bytes of every 32, a 32-byte random fetch, the run-wise
//     scatter of a radix pass, ...): the achievable figure a kernel's rate is to be held against, next to the 8 TB/s of the data sheet;
//   * calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for those patterns: MI355X_MICROARCH.md (HBM section) gives the factor for a wide coalesced
//     stream only and asks for a calibration "on a known byte count in your own access pattern" for everything else.
// Every pattern is a kernel of its own name, so that a --kernel-trace / --pmc pass attributes its counters to it.
#include "mahip_internal.hpp"

#define DG_BINS 128

// one 16-byte load per thread, one block per 4 KB (the shape of k_paf_nl_count)
__global__ __launch_bounds__(256) void k_diag_read16_flat(const uint4 *__restrict__ src, size_t n16, uint32_t *__restrict__ out)
{
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	uint32_t x = 0;
	if (i < n16) { const uint4 v = src[i]; x = v.x ^ v.y ^ v.z ^ v.w; }
	x = wv_sum_u32(x);
	if ((threadIdx.x & 63) == 0 && x == 0x9e3779b9u) out[blockIdx.x & 1023u] = x; // (keeps the load alive; practically never taken)
}

// 8 x 16-byte loads of a thread in flight together, one block per 32 KB (the shape of k_radix_hist); ATOM: + that kernel's LDS atomics (1 = once, 2 = twice)
// ROWS: the block's 128 counts leave as one row of 512 bytes (tile-major) instead of 128 words a tile-count apart (digit-major, what k_radix_hist writes)
template <int ATOM, bool ROWS = false>
__global__ __launch_bounds__(256) void k_diag_read16_x8(const ulonglong2 *__restrict__ src, size_t n16, uint32_t *__restrict__ out, int shift)
{
	__shared__ uint32_t s_cnt[ATOM > 1 ? 2 : 1][DG_BINS];
	if (ATOM) { for (unsigned d = threadIdx.x; d < (ATOM > 1 ? 2u : 1u) * DG_BINS; d += 256) (&s_cnt[0][0])[d] = 0; __syncthreads(); }
	const size_t base = (size_t)blockIdx.x * 2048;
	ulonglong2 kk[8];
	uint32_t x = 0;
	if (base + 2048 <= n16) {
#pragma unroll
		for (int it = 0; it < 8; ++it) kk[it] = src[base + it * 256 + threadIdx.x];
#pragma unroll
		for (int it = 0; it < 8; ++it) {
			if (ATOM) {
				atomicAdd(&s_cnt[0][(unsigned)(kk[it].x >> shift) & (DG_BINS - 1)], 1u);
				atomicAdd(&s_cnt[0][(unsigned)(kk[it].y >> shift) & (DG_BINS - 1)], 1u);
				if (ATOM > 1) {
					atomicAdd(&s_cnt[1][(unsigned)(kk[it].x >> (shift + 7)) & (DG_BINS - 1)], 1u);
					atomicAdd(&s_cnt[1][(unsigned)(kk[it].y >> (shift + 7)) & (DG_BINS - 1)], 1u);
				}
			} else x ^= (uint32_t)kk[it].x ^ (uint32_t)kk[it].y;
		}
	}
	if (ATOM) {
		__syncthreads();
		if (threadIdx.x < DG_BINS) out[ROWS ? (size_t)blockIdx.x * DG_BINS + threadIdx.x : (size_t)threadIdx.x * gridDim.x + blockIdx.x] = s_cnt[0][threadIdx.x] + (ATOM > 1 ? s_cnt[1][threadIdx.x] : 0u);
	} else {
		x = wv_sum_u32(x);
		if ((threadIdx.x & 63) == 0 && x == 0x9e3779b9u) out[blockIdx.x & 1023u] = x;
	}
}

// 8 bytes of every 32-byte record (the shape of k_hit_keys_tiled's read); WRITE: + the 8-byte key it writes per record
template <bool WRITE>
__global__ __launch_bounds__(256) void k_diag_read8_of32(const uint64_t *__restrict__ src, size_t n_rec, uint64_t *__restrict__ dst, uint32_t *__restrict__ out)
{
	uint32_t x = 0;
	const size_t base = (size_t)blockIdx.x * 4096;
	for (int it = 0; it < 16; ++it) {
		const size_t i = base + (size_t)it * 256 + threadIdx.x;
		if (i < n_rec) {
			const uint64_t k = src[i * 4];
			if (WRITE) dst[i] = k >> 32 << 27 | (i & 0x7ffffffu); else x ^= (uint32_t)k;
		}
	}
	if (!WRITE) { x = wv_sum_u32(x); if ((threadIdx.x & 63) == 0 && x == 0x9e3779b9u) out[blockIdx.x & 1023u] = x; }
}

// streaming writes: W = bytes per thread and store (16 or 8), 8 stores per thread
template <int W>
__global__ __launch_bounds__(256) void k_diag_write(void *__restrict__ dst, size_t n_el)
{
	const size_t base = (size_t)blockIdx.x * 2048;
#pragma unroll
	for (int it = 0; it < 8; ++it) {
		const size_t i = base + it * 256 + threadIdx.x;
		if (i < n_el) {
			if (W == 16) ((uint4*)dst)[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
			else ((uint64_t*)dst)[i] = (uint64_t)i;
		}
	}
}

// coalesced copy, W bytes per thread and access
template <int W>
__global__ __launch_bounds__(256) void k_diag_copy(const void *__restrict__ src, void *__restrict__ dst, size_t n_el)
{
	const size_t base = (size_t)blockIdx.x * 2048;
	if (W == 16) {
		uint4 v[8];
#pragma unroll
		for (int it = 0; it < 8; ++it) { const size_t i = base + it * 256 + threadIdx.x; v[it] = i < n_el ? ((const uint4*)src)[i] : make_uint4(0, 0, 0, 0); }
#pragma unroll
		for (int it = 0; it < 8; ++it) { const size_t i = base + it * 256 + threadIdx.x; if (i < n_el) ((uint4*)dst)[i] = v[it]; }
	} else {
		uint64_t v[8];
#pragma unroll
		for (int it = 0; it < 8; ++it) { const size_t i = base + it * 256 + threadIdx.x; v[it] = i < n_el ? ((const uint64_t*)src)[i] : 0; }
#pragma unroll
		for (int it = 0; it < 8; ++it) { const size_t i = base + it * 256 + threadIdx.x; if (i < n_el) ((uint64_t*)dst)[i] = v[it]; }
	}
}

// a bijection of [0, 2^bits) that scatters neighbours far apart (odd multiplier, xor-shift, odd multiplier)
__device__ __forceinline__ uint32_t dg_mix(uint32_t i, int bits)
{
	const uint32_t m = bits >= 32 ? 0xffffffffu : (1u << bits) - 1u;
	i = (i * 0x9e3779b1u) & m; i ^= i >> (bits / 2); i = (i * 0x85ebca6bu) & m;
	return i;
}
// the gather of the first coverage pass without its sort and sweep: slot i fetches the 32-byte record dg_mix(i) (2 x 16 bytes) and, COLS, writes it out as
// eight 4-byte columns (otherwise 4 bytes of it).  ILP independent fetches of a thread are in flight together.
// wbits < bits: the fetches of 2^wbits consecutive slots stay inside one window of 2^wbits records (what a gather sees after the records were partitioned
// by the top bits of their key: is a window that fits the 256 MiB Infinity Cache served from it?)
template <bool COLS, int ILP>
__global__ __launch_bounds__(256) void k_diag_gather32(const uint4 *__restrict__ rec, int bits, uint32_t *__restrict__ col, size_t n_rec, int wbits)
{
	const size_t base = (size_t)blockIdx.x * (256 * ILP);
	uint4 a[ILP], b[ILP];
#pragma unroll
	for (int u = 0; u < ILP; ++u) {
		const size_t i = base + u * 256 + threadIdx.x;
		const size_t j = i >= n_rec ? 0 : wbits < bits ? ((i >> wbits) << wbits) | dg_mix((uint32_t)i & ((1u << wbits) - 1u), wbits) : dg_mix((uint32_t)i, bits);
		a[u] = rec[2 * j]; b[u] = rec[2 * j + 1];
	}
#pragma unroll
	for (int u = 0; u < ILP; ++u) {
		const size_t i = base + u * 256 + threadIdx.x;
		if (i >= n_rec) continue;
		if (COLS) {
			col[i] = a[u].x; col[n_rec + i] = a[u].y; col[2 * n_rec + i] = a[u].z; col[3 * n_rec + i] = a[u].w;
			col[4 * n_rec + i] = b[u].x; col[5 * n_rec + i] = b[u].y; col[6 * n_rec + i] = b[u].z; col[7 * n_rec + i] = b[u].w;
		} else col[i] = a[u].x ^ b[u].w;
	}
}

// what a radix pass's scatter does to memory, without its ranking: a tile of 4096 keys (8 bytes each, read coalesced) leaves as 128 runs of 32 keys,
// run b of tile t at b * (n / 128) + 32 t
__global__ __launch_bounds__(256) void k_diag_scatter_runs(const uint64_t *__restrict__ src, uint64_t *__restrict__ dst, size_t n_tiles)
{
	const size_t t = blockIdx.x, per = n_tiles * 32;
	uint64_t v[16];
#pragma unroll
	for (int it = 0; it < 16; ++it) v[it] = src[t * 4096 + it * 256 + threadIdx.x];
#pragma unroll
	for (int it = 0; it < 16; ++it) {
		const unsigned j = it * 256 + threadIdx.x;
		dst[(size_t)(j >> 5) * per + t * 32 + (j & 31u)] = v[it];
	}
}

__global__ __launch_bounds__(256) void k_diag_fill(uint32_t *__restrict__ p, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
		uint32_t x = (uint32_t)i * 0x9e3779b1u; x ^= x >> 15; x *= 0x85ebca6bu; x ^= x >> 13;
		p[i] = x;
	}
}

static const char *const dg_names[] = {
	"read16_flat", "read16_x8", "read16_x8+lds_atomics", "read16_x8+lds_atomics_x2", "read8_of32", "read8_of32+write8", "write16", "write8", "copy16", "copy8",
	"gather32", "gather32_ilp4", "gather32+cols", "gather32_ilp4+cols", "scatter_runs32",
	"gather32+cols_win32MB", "gather32+cols_win64MB", "gather32+cols_win128MB", "gather32+cols_win256MB", "gather32+cols_win512MB",
	"read16_x8+lds_atomics+rows_out",
};
extern "C" int mahip_diag_patterns(void) { return (int)(sizeof(dg_names) / sizeof(dg_names[0])); }
extern "C" const char *mahip_diag_name(int pattern) { return pattern >= 0 && pattern < mahip_diag_patterns() ? dg_names[pattern] : nullptr; }

// Runs `pattern` over `bytes` of source data `reps` times; *best_ms = the fastest launch (HIP events on the context's stream), *moved = the bytes the
// pattern reads + writes by construction (what a perfect memory system would move: every byte once).
extern "C" int mahip_diag_run(mahip_ctx_t *c, int pattern, size_t bytes, int reps, double *best_ms, double *moved)
{
	HIPCHK(hipSetDevice(c->dev));
	if (pattern < 0 || pattern >= mahip_diag_patterns() || reps < 1) { mahip_set_error("mahip_diag_run: no such pattern"); return -1; }
	bytes &= ~(size_t)((1u << 17) - 1); // whole tiles of every pattern
	if (bytes == 0 || bytes > ((size_t)1 << 36)) { mahip_set_error("mahip_diag_run: size out of range"); return -1; }
	// source in c->key[0], destination in c->key[1], small results in c->hist
	CHK(dev_reserve(c, c->key[0], bytes + 256));
	CHK(dev_reserve(c, c->key[1], bytes + bytes / 8 + 256));
	CHK(dev_reserve(c, c->hist, bytes / 16 + (1u << 20)));
	hipLaunchKernelGGL(k_diag_fill, dim3(8192), dim3(256), 0, c->st, P<uint32_t>(c->key[0]), bytes / 4);
	const void *src = c->key[0].p;
	void *dst = c->key[1].p;
	uint32_t *out = P<uint32_t>(c->hist);
	struct Events { // (destroyed on every way out)
		hipEvent_t e0 = nullptr, e1 = nullptr;
		~Events() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
	} ev;
	HIPCHK(hipEventCreate(&ev.e0)); HIPCHK(hipEventCreate(&ev.e1));
	const hipEvent_t e0 = ev.e0, e1 = ev.e1;
	double best = 1e30, mv = 0;
	for (int r = 0; r < reps; ++r) {
		HIPCHK(hipEventRecord(e0, c->st));
		const size_t n16 = bytes / 16, n8 = bytes / 8, n_rec = bytes / 32;
		int bits = 0; while (((size_t)1 << (bits + 1)) <= n_rec) ++bits; // records the gathers touch: the largest power of two (the bijection's domain)
		const size_t n_g = (size_t)1 << bits;
		switch (pattern) {
		case 0: hipLaunchKernelGGL(k_diag_read16_flat, dim3((unsigned)(n16 / 256)), dim3(256), 0, c->st, (const uint4*)src, n16, out); mv = (double)bytes; break;
		case 1: hipLaunchKernelGGL(k_diag_read16_x8<0>, dim3((unsigned)(n16 / 2048)), dim3(256), 0, c->st, (const ulonglong2*)src, n16, out, 27); mv = (double)bytes; break;
		case 2: hipLaunchKernelGGL(k_diag_read16_x8<1>, dim3((unsigned)(n16 / 2048)), dim3(256), 0, c->st, (const ulonglong2*)src, n16, out, 27); mv = (double)bytes; break;
		case 3: hipLaunchKernelGGL(k_diag_read16_x8<2>, dim3((unsigned)(n16 / 2048)), dim3(256), 0, c->st, (const ulonglong2*)src, n16, out, 27); mv = (double)bytes; break;
		case 4: hipLaunchKernelGGL(k_diag_read8_of32<false>, dim3((unsigned)(n_rec / 4096)), dim3(256), 0, c->st, (const uint64_t*)src, n_rec, (uint64_t*)dst, out); mv = (double)bytes; break; // every line of the records
		case 5: hipLaunchKernelGGL(k_diag_read8_of32<true>, dim3((unsigned)(n_rec / 4096)), dim3(256), 0, c->st, (const uint64_t*)src, n_rec, (uint64_t*)dst, out); mv = (double)bytes + 8.0 * (double)n_rec; break;
		case 6: hipLaunchKernelGGL(k_diag_write<16>, dim3((unsigned)(n16 / 2048)), dim3(256), 0, c->st, dst, n16); mv = (double)bytes; break;
		case 7: hipLaunchKernelGGL(k_diag_write<8>, dim3((unsigned)(n8 / 2048)), dim3(256), 0, c->st, dst, n8); mv = (double)bytes; break;
		case 8: hipLaunchKernelGGL(k_diag_copy<16>, dim3((unsigned)(n16 / 2048)), dim3(256), 0, c->st, src, dst, n16); mv = 2.0 * (double)bytes; break;
		case 9: hipLaunchKernelGGL(k_diag_copy<8>, dim3((unsigned)(n8 / 2048)), dim3(256), 0, c->st, src, dst, n8); mv = 2.0 * (double)bytes; break;
		case 10: hipLaunchKernelGGL((k_diag_gather32<false, 1>), dim3((unsigned)(n_g / 256)), dim3(256), 0, c->st, (const uint4*)src, bits, (uint32_t*)dst, n_g, bits); mv = 36.0 * (double)n_g; break;
		case 11: hipLaunchKernelGGL((k_diag_gather32<false, 4>), dim3((unsigned)(n_g / 1024)), dim3(256), 0, c->st, (const uint4*)src, bits, (uint32_t*)dst, n_g, bits); mv = 36.0 * (double)n_g; break;
		case 12: hipLaunchKernelGGL((k_diag_gather32<true, 1>), dim3((unsigned)(n_g / 256)), dim3(256), 0, c->st, (const uint4*)src, bits, (uint32_t*)dst, n_g, bits); mv = 64.0 * (double)n_g; break;
		case 13: hipLaunchKernelGGL((k_diag_gather32<true, 4>), dim3((unsigned)(n_g / 1024)), dim3(256), 0, c->st, (const uint4*)src, bits, (uint32_t*)dst, n_g, bits); mv = 64.0 * (double)n_g; break;
		case 15: case 16: case 17: case 18: case 19:
			hipLaunchKernelGGL((k_diag_gather32<true, 1>), dim3((unsigned)(n_g / 256)), dim3(256), 0, c->st, (const uint4*)src, bits, (uint32_t*)dst, n_g, 20 + (pattern - 15)); mv = 64.0 * (double)n_g; break;
		case 20: hipLaunchKernelGGL((k_diag_read16_x8<1, true>), dim3((unsigned)(n16 / 2048)), dim3(256), 0, c->st, (const ulonglong2*)src, n16, out, 27); mv = (double)bytes; break;
		case 14: hipLaunchKernelGGL(k_diag_scatter_runs, dim3((unsigned)(n8 / 4096)), dim3(256), 0, c->st, (const uint64_t*)src, (uint64_t*)dst, n8 / 4096); mv = 2.0 * (double)bytes; break;
		}
		HIPCHK(hipEventRecord(e1, c->st));
		HIPCHK(hipEventSynchronize(e1));
		float ms = 0;
		HIPCHK(hipEventElapsedTime(&ms, e0, e1));
		if (ms < best) best = ms;
	}
	HIPCHK(hipGetLastError());
	if (best_ms) *best_ms = best;
	if (moved) *moved = mv;
	return 0;
}
