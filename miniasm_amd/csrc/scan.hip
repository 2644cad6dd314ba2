// scan.hip -- device-wide exclusive prefix sum (u32), three-phase reduce / scan-of-sums / downsweep.
// Used for: radix-sort digit offsets, order-preserving compaction of hits and arcs, the squeeze map
// (reference sdict.c:69-86), CSR offsets.  HBM-bound: reads the input twice, writes it once.
#include "mahip_internal.hpp"

#define SCAN_THREADS 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_THREADS * SCAN_ITEMS)

// block-wide exclusive scan of one value per thread (256 threads = 4 waves); returns the exclusive
// prefix for this thread and the block total in *total
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t x, uint32_t *s_wave, uint32_t *total)
{
	unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t incl = x;
	for (int o = 1; o < 64; o <<= 1) {
		uint32_t y = __shfl_up(incl, o, 64);
		if (lane >= (unsigned)o) incl += y;
	}
	if (lane == 63) s_wave[wave] = incl;
	__syncthreads();
	uint32_t base = 0, tot = 0;
	for (int w = 0; w < SCAN_THREADS / 64; ++w) {
		uint32_t v = s_wave[w];
		if ((unsigned)w < wave) base += v;
		tot += v;
	}
	__syncthreads();
	*total = tot;
	return base + incl - x;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_reduce(const uint32_t *__restrict__ in, uint32_t *__restrict__ bsum, size_t n)
{
	__shared__ uint32_t s_wave[SCAN_THREADS / 64];
	size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
	uint32_t s = 0;
	if (base + SCAN_ITEMS <= n) {
		const uint4 *p = (const uint4*)(in + base);
		uint4 a = p[0], b = p[1];
		s = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
	} else {
		for (int i = 0; i < SCAN_ITEMS; ++i) if (base + i < n) s += in[base + i];
	}
	s = wv_sum_u32(s);
	if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = s;
	__syncthreads();
	if (threadIdx.x == 0) bsum[blockIdx.x] = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
}

// scans one tile; base comes from bbase[blockIdx.x] (exclusive prefix of block sums) or 0
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_down(const uint32_t *in, uint32_t *out, // may alias (in-place)
                                                             const uint32_t *bbase, size_t n, uint32_t *d_total)
{
	__shared__ uint32_t s_wave[SCAN_THREADS / 64];
	size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
	uint32_t v[SCAN_ITEMS], s = 0, tot;
	if (base + SCAN_ITEMS <= n) {
		const uint4 *p = (const uint4*)(in + base);
		uint4 a = p[0], b = p[1];
		v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
	} else {
		for (int i = 0; i < SCAN_ITEMS; ++i) v[i] = base + i < n ? in[base + i] : 0;
	}
	for (int i = 0; i < SCAN_ITEMS; ++i) s += v[i];
	uint32_t ex = block_excl_scan(s, s_wave, &tot) + (bbase ? bbase[blockIdx.x] : 0);
	uint32_t run = ex;
	uint32_t o[SCAN_ITEMS];
	for (int i = 0; i < SCAN_ITEMS; ++i) { o[i] = run; run += v[i]; }
	if (base + SCAN_ITEMS <= n) {
		uint4 *q = (uint4*)(out + base);
		q[0] = make_uint4(o[0], o[1], o[2], o[3]);
		q[1] = make_uint4(o[4], o[5], o[6], o[7]);
	} else {
		for (int i = 0; i < SCAN_ITEMS; ++i) if (base + i < n) out[base + i] = o[i];
	}
	// grand total = exclusive prefix + value of the last element
	if (d_total && base < n && base + SCAN_ITEMS >= n) *d_total = run;
}

static int scan_rec(mahip_ctx *c, const uint32_t *in, uint32_t *out, size_t n, uint32_t *d_total, int level)
{
	if (n == 0) {
		if (d_total) HIPCHK(hipMemsetAsync(d_total, 0, 4, c->st));
		return 0;
	}
	size_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
	if (nb == 1) {
		hipLaunchKernelGGL(k_scan_down, dim3(1), dim3(SCAN_THREADS), 0, c->st, in, out, (const uint32_t*)nullptr, n, d_total);
		return 0;
	}
	if (level >= 3) { mahip_set_error("scan: input too large"); return -1; }
	CHK(dev_reserve(c, c->scan_tmp[level], (nb + 8) * 4));
	uint32_t *bs = P<uint32_t>(c->scan_tmp[level]);
	hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, c->st, in, bs, n);
	CHK(scan_rec(c, bs, bs, nb, nullptr, level + 1));
	hipLaunchKernelGGL(k_scan_down, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, c->st, in, out, (const uint32_t*)bs, n, d_total);
	return 0;
}

int scan_exclusive_u32(mahip_ctx *c, const uint32_t *in, uint32_t *out, size_t n, uint32_t *d_total)
{
	ProfScope ps(c, "scan_exclusive_u32", 8.0 * (double)n);
	CHK(scan_rec(c, in, out, n, d_total, 0));
	HIPCHK(hipGetLastError());
	return 0;
}
