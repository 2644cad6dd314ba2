// radix.hip -- stable LSD radix sort of (u64 key, u32 value) pairs, 8-bit digits, wave64 ballot ranking.
//
// Reproduces WHAT reference hit.c:19-22 (ma_hit_sort -> radix_sort_hit, ksort.h:134-183) and asg.c:22-25
// (asg_arc_sort) compute -- records ordered by their 64-bit key -- with a TOTAL order: ties keep input
// order (the reference's in-place American-flag sort leaves ties in a data-dependent order).
// Only the significant key bits are sorted: two bit ranges [lo0,hi0) and [lo1,hi1), e.g. the bits of the
// query start and the bits of the query id.
//
// Per pass: k_radix_hist (per-tile digit counts, digit-major) -> exclusive scan -> k_radix_scatter
// (stable multi-split).  HBM-bound: per pass reads keys twice, values once, writes both once.
#include "mahip_internal.hpp"

#define RS_THREADS 256
#define RS_ITEMS 8
#define RS_TILE (RS_THREADS * RS_ITEMS)
#define RS_WAVES (RS_THREADS / 64)

__global__ __launch_bounds__(RS_THREADS) void k_radix_hist(const uint64_t *__restrict__ key, uint32_t *__restrict__ hist,
                                                            size_t n, unsigned nb, int shift, unsigned mask)
{
	__shared__ uint32_t s_cnt[256];
	s_cnt[threadIdx.x] = 0;
	__syncthreads();
	size_t base = (size_t)blockIdx.x * RS_TILE;
	for (int it = 0; it < RS_ITEMS; ++it) {
		size_t i = base + (size_t)it * RS_THREADS + threadIdx.x;
		if (i < n) atomicAdd(&s_cnt[(unsigned)(key[i] >> shift) & mask], 1u);
	}
	__syncthreads();
	hist[(size_t)threadIdx.x * nb + blockIdx.x] = s_cnt[threadIdx.x];
}

// Element order inside a tile is (wave, item, lane): element index = tile + wave*64*ITEMS + item*64 + lane.
__global__ __launch_bounds__(RS_THREADS) void k_radix_scatter(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                               uint64_t *__restrict__ kout, uint32_t *__restrict__ vout,
                                                               const uint32_t *__restrict__ gofs, size_t n, unsigned nb, int shift, unsigned mask)
{
	__shared__ uint32_t s_cnt[RS_WAVES][256];
	const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint64_t lt = wv_lt(lane);
	for (int w = 0; w < RS_WAVES; ++w) s_cnt[w][threadIdx.x] = 0;
	__syncthreads();
	size_t wbase = (size_t)blockIdx.x * RS_TILE + (size_t)wave * 64 * RS_ITEMS;
	uint64_t k[RS_ITEMS];
	uint32_t v[RS_ITEMS], r[RS_ITEMS];
	for (int it = 0; it < RS_ITEMS; ++it) {
		size_t i = wbase + (size_t)it * 64 + lane;
		k[it] = i < n ? kin[i] : 0;
		v[it] = i < n ? vin[i] : 0;
	}
	for (int it = 0; it < RS_ITEMS; ++it) {
		size_t i = wbase + (size_t)it * 64 + lane;
		int valid = i < n;
		unsigned d = (unsigned)(k[it] >> shift) & mask;
		uint64_t peers = wv_ballot(valid);
		for (int b = 0; b < 8; ++b) {
			uint64_t bal = wv_ballot((d >> b) & 1);
			peers &= ((d >> b) & 1) ? bal : ~bal;
		}
		uint32_t prev = s_cnt[wave][d];
		wv_sync();
		if (valid && (peers & lt) == 0) s_cnt[wave][d] = prev + (uint32_t)__popcll(peers);
		wv_sync();
		r[it] = prev + (uint32_t)__popcll(peers & lt);
	}
	__syncthreads();
	{ // per digit: exclusive prefix over the waves + global offset of (digit, tile)
		unsigned d = threadIdx.x;
		uint32_t run = gofs[(size_t)d * nb + blockIdx.x];
		for (int w = 0; w < RS_WAVES; ++w) { uint32_t t = s_cnt[w][d]; s_cnt[w][d] = run; run += t; }
	}
	__syncthreads();
	for (int it = 0; it < RS_ITEMS; ++it) {
		size_t i = wbase + (size_t)it * 64 + lane;
		if (i < n) {
			unsigned d = (unsigned)(k[it] >> shift) & mask;
			uint32_t p = s_cnt[wave][d] + r[it];
			kout[p] = k[it];
			vout[p] = v[it];
		}
	}
}

int radix_sort_pairs(mahip_ctx *c, size_t n, int lo0, int hi0, int lo1, int hi1, int *gen)
{
	int g = *gen;
	if (n == 0) return 0;
	if (n >= 0xffffffffull) { mahip_set_error("radix_sort_pairs: too many records"); return -1; }
	unsigned nb = (unsigned)((n + RS_TILE - 1) / RS_TILE);
	CHK(dev_reserve(c, c->hist, ((size_t)256 * nb + 8) * 4));
	for (int range = 0; range < 2; ++range) {
		int lo = range ? lo1 : lo0, hi = range ? hi1 : hi0;
		for (int shift = lo; shift < hi; shift += 8) {
			int bits = hi - shift < 8 ? hi - shift : 8;
			unsigned mask = (1u << bits) - 1;
			uint64_t *kin = P<uint64_t>(c->key[g]), *kout = P<uint64_t>(c->key[g ^ 1]);
			uint32_t *vin = P<uint32_t>(c->val[g]), *vout = P<uint32_t>(c->val[g ^ 1]);
			uint32_t *hist = P<uint32_t>(c->hist);
			{
				ProfScope ps(c, "k_radix_hist", 8.0 * (double)n);
				hipLaunchKernelGGL(k_radix_hist, dim3(nb), dim3(RS_THREADS), 0, c->st, kin, hist, n, nb, shift, mask);
			}
			CHK(scan_exclusive_u32(c, hist, hist, (size_t)256 * nb, nullptr));
			{
				ProfScope ps(c, "k_radix_scatter", 24.0 * (double)n);
				hipLaunchKernelGGL(k_radix_scatter, dim3(nb), dim3(RS_THREADS), 0, c->st, kin, vin, kout, vout, hist, n, nb, shift, mask);
			}
			g ^= 1;
		}
	}
	HIPCHK(hipGetLastError());
	*gen = g;
	return 0;
}
