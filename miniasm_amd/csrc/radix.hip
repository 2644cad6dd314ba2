// radix.hip -- stable LSD radix sort of 64-bit keys (optionally with a 32-bit value), wave64 ballot ranking.
//
// Reproduces WHAT reference hit.c:19-22 (ma_hit_sort -> radix_sort_hit, ksort.h:134-183) and asg.c:22-25
// (asg_arc_sort) compute -- records ordered by their 64-bit key -- with a TOTAL order: ties keep input
// order (the reference's in-place American-flag sort leaves ties in a data-dependent order).
//
// Design for MI355X (HBM-bound):
//   * only the significant key bits are sorted, in digits of up to 9 bits chosen to minimise the pass count
//     (hits: 16 bits of start + 18..21 bits of read id = 4 passes of <= 10 bits instead of 8 byte-passes over 64 bits);
//   * when the record index fits below the key bits the index rides in the low bits of the key itself
//     (8-byte elements, no value array): a pass reads 8 B twice and writes 8 B per record;
//   * per pass: k_radix_hist (per-tile digit counts, one ROW per tile) -> scan down the columns (k_radix_colscan_*) -> k_radix_scatter;
//   * tiles of 4096 keys (256 threads x 16): ranks come from wave ballots (stable multi-split), the tile is
//     reordered through LDS so that consecutive lanes write consecutive addresses of a digit's run.
#include "mahip_internal.hpp"
#include <sys/mman.h>
#include <thread>

#define RS_THREADS 256
#ifndef RS_ITEMS
#define RS_ITEMS 16
#endif
#define RS_TILE (RS_THREADS * RS_ITEMS)
#define RS_WAVES (RS_THREADS / 64)
#define RS_BINS (1 << RS_MAXBITS) // RS_MAXBITS: mahip_internal.hpp

// Tiles that are neighbours write neighbouring runs of every digit; on the same XCD (workgroup i runs on XCD i mod 8, one L2 each) the pieces of a cache
// line they share meet in one L2 before they are written back: the tile a block works on is chosen so that an XCD sees consecutive tiles.
__device__ __forceinline__ unsigned rs_tile_id()
{
	const unsigned b = blockIdx.x, g8 = gridDim.x & ~7u;
	return b < g8 ? (b & 7u) * (g8 >> 3) + (b >> 3) : b;
}

__global__ __launch_bounds__(RS_THREADS) void k_radix_hist(const uint64_t *__restrict__ key, uint32_t *__restrict__ hist,
                                                            size_t n, unsigned nb, int shift, unsigned mask)
{
	__shared__ uint32_t s_cnt[RS_BINS];
	for (unsigned d = threadIdx.x; d <= mask; d += RS_THREADS) s_cnt[d] = 0;
	__syncthreads();
	const unsigned tid_ = rs_tile_id();
	size_t base = (size_t)tid_ * RS_TILE;
	if (base + RS_TILE <= n) { // full tile: two keys per 16-byte load, all loads in flight before the first LDS atomic
		const ulonglong2 *p = (const ulonglong2*)(key + base);
		ulonglong2 kk[RS_ITEMS / 2];
#pragma unroll
		for (int it = 0; it < RS_ITEMS / 2; ++it) kk[it] = p[it * RS_THREADS + threadIdx.x];
#pragma unroll
		for (int it = 0; it < RS_ITEMS / 2; ++it) {
			atomicAdd(&s_cnt[(unsigned)(kk[it].x >> shift) & mask], 1u);
			atomicAdd(&s_cnt[(unsigned)(kk[it].y >> shift) & mask], 1u);
		}
	} else {
		for (int it = 0; it < RS_ITEMS; ++it) {
			size_t i = base + (size_t)it * RS_THREADS + threadIdx.x;
			if (i < n) atomicAdd(&s_cnt[(unsigned)(key[i] >> shift) & mask], 1u);
		}
	}
	__syncthreads();
	for (unsigned d = threadIdx.x; d <= mask; d += RS_THREADS) hist[(size_t)tid_ * (mask + 1u) + d] = s_cnt[d]; // the tile's row (radix_hist_layout)
}

// Element order inside a tile is (wave, item, lane): element index = tile + wave*64*ITEMS + item*64 + lane.
// NBITS > 0: the digit's width at compile time (the hit sort's three digits of 7 bits at BASELINE configs[4]): the ballot loop unrolls, and the block keeps
// 2^NBITS bins instead of RS_BINS (35 KB of LDS against 42: four blocks per CU).  NBITS = 0: any width up to RS_MAXBITS.
// GROUPS (the LAST pass of a key sort whose sorted bits [gr.lo, ..) are a group id, e.g. the hits' read id): the pass also notes where every group starts in the
// output, gr.start[id] = smallest output slot of a key with that id -- the sweep over the sorted keys that looked for the boundaries (k_hit_goff) is not needed.
// The keys of a tile arrive ordered by the bits below this pass's digit, so inside one of the tile's runs the ids do not decrease: a key whose id differs from
// its predecessor's in the run is the first of its id anywhere; the first key of a run may continue an id of the tile before: atomicMin settles both.
// Ids without keys keep the initial ~0 (radix_group_starts_finish closes them).
struct RsGroups { uint32_t *start; int lo; uint32_t n_id; };
// Where a tile's runs start.  The per-tile digit counts are ROWS of (mask + 1) words -- a tile writes and reads its counts with one coalesced access; round 4 until
// visit R kept them digit-major, 128 words a tile-count apart: every 4-byte store a 32-byte sector of its own, and k_radix_hist at 4.6 TB/s where the same read with a
// row written runs at 5.9 (profiles/r04_experiments.txt).  The scan runs DOWN the columns in chunks of RS_CHUNK tiles: cnt[t][d] becomes the count of digit d in the
// chunk's tiles in front of t, pre[c][d] the count in the chunks in front of c, tot[d] the digit's total; a run starts at sum(tot[< d]) + pre + cnt.
struct RsOffsets { const uint32_t *cnt, *pre, *tot; };
#define RS_CHUNK 64u
template <bool HAS_VAL, int NBITS, bool GROUPS = false>
__global__ __launch_bounds__(RS_THREADS) void k_radix_scatter(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                               uint64_t *__restrict__ kout, uint32_t *__restrict__ vout,
                                                               RsOffsets go, size_t n, unsigned nb, int shift, unsigned mask, int nbits_rt, RsGroups gr)
{
	constexpr int NB = NBITS ? (1 << NBITS) : RS_BINS;
	const int nbits = nbits_rt;
	__shared__ uint32_t s_cnt[RS_WAVES][NB]; // per-wave digit counts, then the wave's first local slot per digit
	__shared__ uint32_t s_gb[NB];            // global offset of the digit's run minus its first local slot
	__shared__ uint32_t s_scan[RS_WAVES];
	__shared__ uint64_t s_key[RS_TILE];
	__shared__ uint32_t s_val[HAS_VAL ? RS_TILE : 1];
	const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint64_t lt = wv_lt(lane);
	for (unsigned d = threadIdx.x; d < RS_WAVES * NB; d += RS_THREADS) (&s_cnt[0][0])[d] = 0;
	__syncthreads();
	const unsigned tid_ = rs_tile_id();
	const size_t tile = (size_t)tid_ * RS_TILE, wbase = tile + (size_t)wave * 64 * RS_ITEMS;
	const unsigned n_tile = (unsigned)(n - tile < RS_TILE ? n - tile : RS_TILE);
	uint64_t k[RS_ITEMS];
	uint32_t v[RS_ITEMS], r[RS_ITEMS];
#pragma unroll
	for (int it = 0; it < RS_ITEMS; ++it) {
		size_t i = wbase + (size_t)it * 64 + lane;
		k[it] = i < n ? kin[i] : 0;
		if (HAS_VAL) v[it] = i < n ? vin[i] : 0;
	}
#pragma unroll
	for (int it = 0; it < RS_ITEMS; ++it) { // stable rank inside the wave: lanes with my digit before me + earlier items
		size_t i = wbase + (size_t)it * 64 + lane;
		int valid = i < n;
		unsigned d = (unsigned)(k[it] >> shift) & mask;
		uint64_t peers = wv_ballot(valid);
		uint32_t p_lo = (uint32_t)peers, p_hi = (uint32_t)(peers >> 32); // the lanes with my digit, in halves: a step is one three-input bit operation per half
		auto split = [&](int b) {
#if defined(__HIP_DEVICE_COMPILE__)
			const uint32_t same = (uint32_t)__builtin_amdgcn_sbfe((int)d, (unsigned)b, 1u); // all ones where my bit is set (v_bfe_i32)
#else
			const uint32_t same = 0u - ((d >> b) & 1u);
#endif
			const uint64_t bal = wv_ballot(same != 0u);
			p_lo &= ~((uint32_t)bal ^ same); p_hi &= ~((uint32_t)(bal >> 32) ^ same); // lanes whose bit b equals mine
		};
		if constexpr (NBITS > 0) {
#pragma unroll
			for (int b = 0; b < NBITS; ++b) split(b);
		} else for (int b = 0; b < nbits; ++b) split(b);
		peers = (uint64_t)p_hi << 32 | p_lo;
		uint32_t prev = s_cnt[wave][d];
		wv_sync();
		if (valid && (peers & lt) == 0) s_cnt[wave][d] = prev + (uint32_t)__popcll(peers);
		wv_sync();
		r[it] = prev + (uint32_t)__popcll(peers & lt);
	}
	__syncthreads();
	{ // thread t owns RS_DPT consecutive digits (the first NB threads one each when there are fewer bins than threads): waves' counts -> exclusive over
		// waves; tile counts -> exclusive over digits
		constexpr int RS_DPT = NB >= RS_THREADS ? NB / RS_THREADS : 1;
		const unsigned d0 = RS_DPT * threadIdx.x;
		uint32_t cd[RS_DPT], wo[RS_DPT][RS_WAVES], sum = 0;
#pragma unroll
		for (int k = 0; k < RS_DPT; ++k) {
			uint32_t cc = 0;
			if (d0 + k < (unsigned)NB) for (int w = 0; w < RS_WAVES; ++w) { wo[k][w] = cc; cc += s_cnt[w][d0 + k]; }
			cd[k] = cc; sum += cc;
		}
		uint32_t gt[RS_DPT], gsum = 0, goff[RS_DPT]; // the digits' totals and this tile's offsets inside them: asked for before the barriers of the scans below
#pragma unroll
		for (int k = 0; k < RS_DPT; ++k) {
			const bool in = d0 + k <= mask;
			gt[k] = in ? go.tot[d0 + k] : 0u; gsum += gt[k];
			goff[k] = in ? go.pre[(size_t)(tid_ / RS_CHUNK) * (mask + 1u) + d0 + k] + go.cnt[(size_t)tid_ * (mask + 1u) + d0 + k] : 0u;
		}
		uint32_t tot, ex = block_excl_scan_256(sum, s_scan, &tot);
		uint32_t gtot, gb = block_excl_scan_256(gsum, s_scan, &gtot); // where the digit's run starts in the output: exclusive over the totals
#pragma unroll
		for (int k = 0; k < RS_DPT; ++k) {
			if (d0 + k < (unsigned)NB) {
				for (int w = 0; w < RS_WAVES; ++w) s_cnt[w][d0 + k] = ex + wo[k][w];
				if (d0 + k <= mask) s_gb[d0 + k] = gb + goff[k] - ex;
			}
			ex += cd[k]; gb += gt[k];
		}
	}
	__syncthreads();
#pragma unroll
	for (int it = 0; it < RS_ITEMS; ++it) { // tile reordered by digit in LDS
		size_t i = wbase + (size_t)it * 64 + lane;
		if (i < n) {
			unsigned d = (unsigned)(k[it] >> shift) & mask;
			uint32_t p = s_cnt[wave][d] + r[it];
			s_key[p] = k[it];
			if (HAS_VAL) s_val[p] = v[it];
		}
	}
	__syncthreads();
#pragma unroll 4
	for (int j = 0; j < RS_ITEMS; ++j) { // consecutive lanes -> consecutive slots of a digit's run -> consecutive addresses
		unsigned p = j * RS_THREADS + threadIdx.x;
		if (p < n_tile) {
			uint64_t kk = s_key[p];
			uint32_t g = s_gb[(unsigned)(kk >> shift) & mask] + p;
			kout[g] = kk;
			if (HAS_VAL) vout[g] = s_val[p];
			if (GROUPS) {
				const uint64_t kp = p ? s_key[p - 1] : 0;
				const uint32_t id = (uint32_t)(kk >> gr.lo);
				const bool run_first = p == 0 || ((unsigned)(kp >> shift) & mask) != ((unsigned)(kk >> shift) & mask);
				if (run_first || id != (uint32_t)(kp >> gr.lo)) atomicMin(&gr.start[id < gr.n_id ? id : gr.n_id], g); // (ids out of range, a caller's error: in front of the sentinel, as k_hit_goff has them)
			}
		}
	}
}

// choose digit widths: as few passes as possible, each <= RS_MAXBITS, widths balanced
static int plan_digits(int lo, int hi, int *shift, int *bits)
{
	int nb = hi - lo, np, i, s = lo;
	if (nb <= 0) return 0;
	np = (nb + RS_MAXBITS - 1) / RS_MAXBITS;
	for (i = 0; i < np; ++i) {
		int w = (nb - (s - lo) + (np - i) - 1) / (np - i);
		shift[i] = s; bits[i] = w; s += w;
	}
	return np;
}

// sort key[*gen] (and val[*gen] if has_val) on key bits [lo0,hi0) then [lo1,hi1); result in generation *gen
// the first digit of a key sort on bits [lo,hi) and the histogram layout (a row of 2^bits words per RS_TILE keys at the start of c->hist), for a
// producer of the keys that counts the first digit on the fly (k_hit_keys_tiled) and so saves the first histogram pass
void radix_first_digit(int lo, int hi, int *shift, int *bits, unsigned *tile)
{
	int sh[16], bt[16];
	*shift = lo; *bits = 0; *tile = RS_TILE;
	if (plan_digits(lo, hi, sh, bt) > 0) *shift = sh[0], *bits = bt[0];
}

// ---- the scan down the columns of the count rows (RsOffsets) ----
// a chunk of RS_CHUNK tiles: every column's counts become exclusive prefixes inside the chunk (in place), the column's sum goes to pre[chunk][d]
__global__ __launch_bounds__(256) void k_radix_colscan_chunk(uint32_t *__restrict__ cnt, unsigned nb, unsigned nbd, uint32_t *__restrict__ pre)
{
	const size_t t0 = (size_t)blockIdx.x * RS_CHUNK;
	for (unsigned d = threadIdx.x; d < nbd; d += 256) {
		uint32_t run = 0;
		for (unsigned r = 0; r < RS_CHUNK; r += 16) { // 16 rows in flight
			uint32_t v[16];
#pragma unroll
			for (int k = 0; k < 16; ++k) v[k] = t0 + r + k < nb ? cnt[(t0 + r + k) * nbd + d] : 0u;
#pragma unroll
			for (int k = 0; k < 16; ++k) { if (t0 + r + k < nb) cnt[(t0 + r + k) * nbd + d] = run; run += v[k]; }
		}
		pre[(size_t)blockIdx.x * nbd + d] = run;
	}
}
// one block per digit: the chunks' sums of its column become exclusive prefixes (in place), the column's total goes to tot[d]
__global__ __launch_bounds__(256) void k_radix_colscan_top(uint32_t *__restrict__ pre, unsigned n_chunk, unsigned nbd, uint32_t *__restrict__ tot)
{
	__shared__ uint32_t s_w[4];
	const unsigned d = blockIdx.x, per = (n_chunk + 255) / 256, c0 = threadIdx.x * per;
	uint32_t sum = 0;
	for (unsigned c = c0; c < c0 + per && c < n_chunk; ++c) sum += pre[(size_t)c * nbd + d];
	uint32_t total, ex = block_excl_scan_256(sum, s_w, &total);
	for (unsigned c = c0; c < c0 + per && c < n_chunk; ++c) { const uint32_t v = pre[(size_t)c * nbd + d]; pre[(size_t)c * nbd + d] = ex; ex += v; }
	if (threadIdx.x == 0) tot[d] = total;
}
// the layout of c->hist for nb tiles of nbd bins: rows, then the chunks' prefixes, then the totals
static RsOffsets radix_hist_layout(mahip_ctx *c, unsigned nb, unsigned nbd, uint32_t **cnt, uint32_t **pre, uint32_t **tot)
{
	const unsigned n_chunk = (nb + RS_CHUNK - 1) / RS_CHUNK;
	*cnt = P<uint32_t>(c->hist); *pre = *cnt + (size_t)nb * nbd; *tot = *pre + (size_t)n_chunk * nbd;
	RsOffsets o = {*cnt, *pre, *tot};
	return o;
}
static size_t radix_hist_words(unsigned nb) { return ((size_t)nb + (nb + RS_CHUNK - 1) / RS_CHUNK + 1) * RS_BINS + 8; }
static int radix_colscan(mahip_ctx *c, unsigned nb, unsigned nbd)
{
	uint32_t *cnt, *pre, *tot;
	(void)radix_hist_layout(c, nb, nbd, &cnt, &pre, &tot);
	const unsigned n_chunk = (nb + RS_CHUNK - 1) / RS_CHUNK;
	ProfScope ps(c, "k_radix_colscan", 8.0 * (double)nb * nbd);
	hipLaunchKernelGGL(k_radix_colscan_chunk, dim3(n_chunk), dim3(256), 0, c->st, cnt, nb, nbd, pre);
	hipLaunchKernelGGL(k_radix_colscan_top, dim3(nbd), dim3(256), 0, c->st, pre, n_chunk, nbd, tot);
	HIPCHK(hipGetLastError());
	return 0;
}

// ---- group starts from the last pass (RsGroups) ----
// start[id] = first slot of the id's keys, ~0 for ids without keys, on entry; on exit every id without keys points at the next id's first slot (an empty group) and
// start[n_id] = n: the offsets of a CSR over the sorted keys.  Two small launches over the n_id + 1 entries (a suffix minimum).
#define GS_TILE 2048u
__global__ __launch_bounds__(256) void k_group_tile_min(const uint32_t *__restrict__ start, uint32_t n_ent, uint32_t *__restrict__ tmin)
{
	__shared__ uint32_t s_w[4];
	uint32_t m = 0xffffffffu;
	for (uint32_t i = blockIdx.x * GS_TILE + threadIdx.x; i < n_ent && i < (blockIdx.x + 1) * GS_TILE; i += 256) { const uint32_t v = start[i]; m = v < m ? v : m; }
	for (int o = 32; o; o >>= 1) { const uint32_t y = __shfl_xor(m, o, 64); m = y < m ? y : m; }
	if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = m;
	__syncthreads();
	if (threadIdx.x == 0) { for (int w = 1; w < 4; ++w) m = s_w[w] < m ? s_w[w] : m; tmin[blockIdx.x] = m; }
}
__global__ __launch_bounds__(256) void k_group_close(uint32_t *__restrict__ start, uint32_t n_ent, const uint32_t *__restrict__ tmin, uint32_t n_tiles)
{
	__shared__ uint32_t s_w[4], s_v[GS_TILE];
	uint32_t m = 0xffffffffu; // the smallest start in the tiles behind this one
	for (uint32_t t = blockIdx.x + 1 + threadIdx.x; t < n_tiles; t += 256) { const uint32_t v = tmin[t]; m = v < m ? v : m; }
	for (int o = 32; o; o >>= 1) { const uint32_t y = __shfl_xor(m, o, 64); m = y < m ? y : m; }
	if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = m;
	const uint32_t base = blockIdx.x * GS_TILE;
	for (uint32_t j = threadIdx.x; j < GS_TILE; j += 256) s_v[j] = base + j < n_ent ? start[base + j] : 0xffffffffu;
	__syncthreads();
	uint32_t behind = s_w[0];
	for (int w = 1; w < 4; ++w) behind = s_w[w] < behind ? s_w[w] : behind;
	// thread t owns 8 consecutive entries: the suffix minimum inside them, then over the threads behind it (a wave scan from the right, then the waves behind)
	const uint32_t j0 = threadIdx.x * 8u;
	uint32_t v[8], run = 0xffffffffu;
#pragma unroll
	for (int k = 7; k >= 0; --k) { run = s_v[j0 + k] < run ? s_v[j0 + k] : run; v[k] = run; }
	uint32_t suf = run; // min over this thread's entries and those of the threads behind it in the wave
	const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_down(suf, o, 64); if (lane + o < 64) suf = y < suf ? y : suf; }
	__syncthreads(); // (everybody has read the tiles' minimum from s_w)
	if (lane == 0) s_w[wave] = suf;
	__syncthreads();
	uint32_t after = behind; // everything behind this thread's wave
	for (unsigned w = wave + 1; w < 4; ++w) after = s_w[w] < after ? s_w[w] : after;
	const uint32_t nxt = __shfl_down(suf, 1, 64);
	const uint32_t right = lane < 63 ? (nxt < after ? nxt : after) : after; // min over everything behind this thread
#pragma unroll
	for (int k = 0; k < 8; ++k) if (base + j0 + k < n_ent) start[base + j0 + k] = v[k] < right ? v[k] : right;
}
int radix_group_starts_begin(mahip_ctx *c, uint32_t *start, uint32_t n_id, uint32_t n)
{
	HIPCHK(hipMemsetAsync(start, 0xff, (size_t)n_id * 4, c->st));
	HIPCHK(hipMemsetD32Async((hipDeviceptr_t)(start + n_id), (int)n, 1, c->st)); // the sentinel that closes the last groups
	return 0;
}
int radix_group_starts_finish(mahip_ctx *c, uint32_t *start, uint32_t n_id)
{
	const uint32_t n_ent = n_id + 1, n_tiles = (n_ent + GS_TILE - 1) / GS_TILE;
	CHK(dev_reserve(c, c->gs_tmp, ((size_t)n_tiles + 8) * 4));
	ProfScope ps(c, "k_group_close", 12.0 * (double)n_ent);
	hipLaunchKernelGGL(k_group_tile_min, dim3(n_tiles), dim3(256), 0, c->st, (const uint32_t*)start, n_ent, P<uint32_t>(c->gs_tmp));
	hipLaunchKernelGGL(k_group_close, dim3(n_tiles), dim3(256), 0, c->st, start, n_ent, (const uint32_t*)P<uint32_t>(c->gs_tmp), n_tiles);
	HIPCHK(hipGetLastError());
	return 0;
}

static int radix_sort_impl(mahip_ctx *c, size_t n, int lo0, int hi0, int lo1, int hi1, int *gen, bool has_val, bool first_hist_ready = false, const RadixGroups *groups = nullptr)
{
	int g = *gen, shift[16], bits[16], np;
	if (n == 0) return 0;
	if (n >= 0xffffffffull) { mahip_set_error("radix sort: too many records"); return -1; }
	np = plan_digits(lo0, hi0, shift, bits);
	np += plan_digits(lo1, hi1, shift + np, bits + np);
	unsigned nb = (unsigned)((n + RS_TILE - 1) / RS_TILE);
	CHK(dev_reserve(c, c->hist, radix_hist_words(nb) * 4));
	for (int p = 0; p < np; ++p) {
		unsigned mask = (1u << bits[p]) - 1;
		uint64_t *kin = P<uint64_t>(c->key[g]), *kout = P<uint64_t>(c->key[g ^ 1]);
		uint32_t *vin = P<uint32_t>(c->val[g]), *vout = P<uint32_t>(c->val[g ^ 1]);
		uint32_t *hist, *hpre, *htot;
		const RsOffsets ro = radix_hist_layout(c, nb, mask + 1u, &hist, &hpre, &htot);
		if (!(p == 0 && first_hist_ready)) {
			ProfScope ps(c, c->radix_arcs ? "k_arc_radix_hist" : "k_radix_hist", 8.0 * (double)n);
			hipLaunchKernelGGL(k_radix_hist, dim3(nb), dim3(RS_THREADS), 0, c->st, kin, hist, n, nb, shift[p], mask);
		}
		CHK(radix_colscan(c, nb, mask + 1u));
		{
			ProfScope ps(c, c->radix_arcs ? "k_arc_radix_scatter" : "k_radix_scatter", (has_val ? 24.0 : 16.0) * (double)n);
			const dim3 gr(nb), bl(RS_THREADS);
			const RsGroups nog = {nullptr, 0, 0};
			if (groups && p == np - 1) { // the last pass notes the group starts
				const RsGroups gg = {groups->start, groups->lo, groups->n_id};
				if (bits[p] == 7) hipLaunchKernelGGL((k_radix_scatter<false, 7, true>), gr, bl, 0, c->st, kin, vin, kout, vout, ro, n, nb, shift[p], mask, bits[p], gg);
				else hipLaunchKernelGGL((k_radix_scatter<false, 0, true>), gr, bl, 0, c->st, kin, vin, kout, vout, ro, n, nb, shift[p], mask, bits[p], gg);
			}
			else if (has_val && bits[p] == 7) hipLaunchKernelGGL((k_radix_scatter<true, 7>), gr, bl, 0, c->st, kin, vin, kout, vout, ro, n, nb, shift[p], mask, bits[p], nog);
			else if (has_val) hipLaunchKernelGGL((k_radix_scatter<true, 0>), gr, bl, 0, c->st, kin, vin, kout, vout, ro, n, nb, shift[p], mask, bits[p], nog);
			else if (bits[p] == 7) hipLaunchKernelGGL((k_radix_scatter<false, 7>), gr, bl, 0, c->st, kin, vin, kout, vout, ro, n, nb, shift[p], mask, bits[p], nog);
			else hipLaunchKernelGGL((k_radix_scatter<false, 0>), gr, bl, 0, c->st, kin, vin, kout, vout, ro, n, nb, shift[p], mask, bits[p], nog);
		}
		g ^= 1;
	}
	HIPCHK(hipGetLastError());
	*gen = g;
	return 0;
}

int radix_sort_pairs(mahip_ctx *c, size_t n, int lo0, int hi0, int lo1, int hi1, int *gen)
{
	return radix_sort_impl(c, n, lo0, hi0, lo1, hi1, gen, true);
}

int radix_sort_keys(mahip_ctx *c, size_t n, int lo, int hi, int *gen, bool first_hist_ready, const RadixGroups *groups)
{
	if (groups) CHK(radix_group_starts_begin(c, groups->start, groups->n_id, (uint32_t)n));
	CHK(radix_sort_impl(c, n, lo, hi, 0, 0, gen, false, first_hist_ready, groups));
	if (groups) CHK(radix_group_starts_finish(c, groups->start, groups->n_id));
	return 0;
}

int radix_reserve_hist(mahip_ctx *c, size_t n)
{
	unsigned nb = (unsigned)((n + RS_TILE - 1) / RS_TILE);
	return dev_reserve(c, c->hist, radix_hist_words(nb) * 4);
}

// ---- exact-tie mode (include/mahip.h: mahip_set_exact_ties) ----
// The order the reference's in-place MSD radix sort gives to equal keys is a sequential function of the whole input: the host walks it
// (host/refsort.c) on 8-byte elements -- the key squeezed to the bits it has, above the record's input position -- that the DEVICE packs on the
// way out and takes apart on the way back: one 8-byte array travels each way and the host allocates nothing but that array (until round 3 the
// raw keys went down, the host looked for their bounds, packed, sorted, unpacked, and a 4-byte permutation went up: at BASELINE configs[4] 0.4 s of
// packing, 0.2 s of unpacking and 20 GB of freshly faulted host memory around a 3 s walk).
extern "C" int ma_refsort_perm(const uint64_t *keys, size_t n, uint32_t *perm);
extern "C" int ma_refsort_packed(uint64_t *pk, size_t n, int bl, int bi, int shift_top, const uint8_t *dig_top);
extern "C" int ma_refsort_packed_wanted(uint64_t *pk, size_t n, int bl, int bi, int shift_top, const uint8_t *dig_top, const uint32_t *wcum, uint64_t n_ids);

__global__ __launch_bounds__(256) void k_key_bounds(const uint64_t *__restrict__ key, size_t n, unsigned long long *__restrict__ ctr)
{
	uint32_t mh = 0, ml = 0;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
		const uint64_t k = key[i];
		const uint32_t h = (uint32_t)(k >> 32), l = (uint32_t)k;
		mh = h > mh ? h : mh; ml = l > ml ? l : ml;
	}
	blk_max_u64(&ctr[CT_MAXQID], mh);
	blk_max_u64(&ctr[CT_MAXQS], ml);
}
// in place: key[i] -> (hi & himask) << bl | lo) << bi | i; dig (optional): the key's digit at shift_top, which the packed element no longer holds
__global__ __launch_bounds__(256) void k_pack_keys(uint64_t *__restrict__ key, size_t n, int bl, int bi, uint64_t himask, int shift_top, uint8_t *__restrict__ dig)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) {
		const uint64_t k = key[i];
		key[i] = ((((k >> 32) & himask) << bl) | (k & 0xffffffffull)) << bi | (uint64_t)i;
		if (dig) dig[i] = (uint8_t)(k >> shift_top);
	}
}
__global__ __launch_bounds__(256) void k_perm_from_packed(const uint64_t *__restrict__ pk, size_t n, uint64_t mask, uint32_t *__restrict__ perm)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) perm[i] = (uint32_t)(pk[i] & mask);
}
// restricted walk: block s writes the order of the s-th wanted read's hits (ord[off[s] ..)) over the stable order at the read's stretch pos[s] ..
__global__ __launch_bounds__(256) void k_perm_patch(const uint32_t *__restrict__ pos, const uint32_t *__restrict__ off, const uint32_t *__restrict__ ord, uint32_t *__restrict__ perm)
{
	const uint32_t p = pos[blockIdx.x], o0 = off[blockIdx.x], o1 = off[blockIdx.x + 1];
	for (uint32_t j = o0 + threadIdx.x; j < o1; j += 256) perm[p + (j - o0)] = ord[j];
}

static int bitlen64(uint64_t x) { int b = 0; while (x) ++b, x >>= 1; return b; }

// The walk's arrays (c->hwalk, c->hdig) stay with the context between the hit walk and the arc walk that follows it and are dropped by a thread of their
// own when the repair is over (walk_scratch_release): unmapping 9 GB took 0.5 s of the 5.8 s of BASELINE configs[4], 60 ms of the 1.1 s of the 50 M-overlap
// noisy input, while the device was waiting for its next kernel.
bool BigHost::reserve(size_t n)
{
	static int thp = -1;
	if (n <= bytes && p) return true;
	drop();
	if (thp < 0) { const char *s = getenv("MA_HOST_THP"); thp = s && atoi(s) != 0; }
	if (thp && n >= ((size_t)64 << 20)) { // transparent huge pages, on request: a fault per 2 MB instead of per 4 KB, fewer TLB misses at the walk's heads, a shorter munmap
		const size_t b = (n + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
		void *q = mmap(nullptr, b, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
		if (q != MAP_FAILED) { (void)madvise(q, b, MADV_HUGEPAGE); p = q; bytes = b; mapped = true; return true; }
	}
	p = malloc(n); bytes = p ? n : 0; mapped = false;
	return p != nullptr;
}
void BigHost::drop()
{
	if (p) {
		// Pages first, piece by piece, with MADV_DONTNEED -- that runs under the READ side of the address-space lock, so the page faults and allocations of the
		// other threads go on (the tail starts while this thread is still busy: a plain munmap of 9 GB kept `names+sub` of BASELINE configs[4] waiting for 0.59 s);
		// what is left for munmap / free is an empty mapping.
		if (bytes >= ((size_t)64 << 20)) {
			const size_t pg = 4096, step = (size_t)256 << 20;
			char *a = (char*)(((uintptr_t)p + pg - 1) & ~(uintptr_t)(pg - 1)), *e = (char*)(((uintptr_t)p + bytes) & ~(uintptr_t)(pg - 1));
			for (; a < e; a += step) (void)madvise(a, (size_t)(e - a) < step ? (size_t)(e - a) : step, MADV_DONTNEED);
		}
		if (mapped) munmap(p, bytes); else free(p);
	}
	p = nullptr; bytes = 0; mapped = false;
}

void walk_scratch_release(mahip_ctx *c)
{
	if (!c->hwalk.p && !c->hdig.p) return;
	BigHost a = c->hwalk, b = c->hdig;
	c->hwalk = BigHost(); c->hdig = BigHost();
	if (a.bytes + b.bytes < ((size_t)32 << 20)) { a.drop(); b.drop(); return; }
	try { std::thread([a, b]() mutable { a.drop(); b.drop(); }).detach(); }
	catch (...) { a.drop(); b.drop(); } // no thread to be had: here and now
}

// d_perm[j] <- input position of the j-th record in the reference's order.  d_keys is overwritten.
int reference_order(mahip_ctx *c, uint64_t *d_keys, size_t n, uint32_t *d_perm, const WalkWanted *w)
{
	if (n == 0) return 0;
	TieLaps tl(c);
	unsigned long long *ctr = P<unsigned long long>(c->ctr);
	HIPCHK(hipMemsetAsync(ctr + CT_MAXQID, 0, 16, c->st)); // (CT_MAXQID, CT_MAXQS: adjacent)
	hipLaunchKernelGGL(k_key_bounds, dim3(grid_for(n, 256, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, (const uint64_t*)d_keys, n, ctr);
	CHK(ctr_fetch(c));
	int bh = bitlen64(c->h_ctr[CT_MAXQID]), bl = bitlen64(c->h_ctr[CT_MAXQS]), bi = bitlen64(n - 1), shift_top = -1;
	if (bh == 0) bh = 1;
	if (bl == 0) bl = 1;
	if (bi == 0) bi = 1;
	uint64_t himask = 0xffffffffull;
	const bool force_apart = getenv("MA_REFSORT_TOP_APART") != nullptr; // tests: the wide-key form on inputs of any size
	bool packed = !force_apart && bh + bl + bi <= 64;
	if (!packed && (n > (1u << 17) || (force_apart && n > 64))) { // too wide for one word: without the top level's digit (it goes down as a byte array of its own) if the rest fits
		shift_top = (32 + bh - 1) & ~7;
		if ((shift_top - 32) + bl + bi <= 64) { packed = true; himask = shift_top > 32 ? (1ull << (shift_top - 32)) - 1 : 0; }
		else shift_top = -1;
	}
	int rc = 0;
	if (packed && !getenv("MA_REFSORT_KEYS")) {
		if (!c->hwalk.reserve(n * 8) || (shift_top >= 0 && !c->hdig.reserve(n + 16))) { mahip_set_error("reference_order: out of host memory"); return -1; }
		uint64_t *hk = (uint64_t*)c->hwalk.p;
		uint8_t *hd = shift_top >= 0 ? (uint8_t*)c->hdig.p : nullptr;
		if (shift_top >= 0) CHK(dev_reserve(c, c->tdig, n + 16));
		hipLaunchKernelGGL(k_pack_keys, dim3(grid_for(n, 256)), dim3(256), 0, c->st, d_keys, n, bl, bi, himask, shift_top, shift_top >= 0 ? P<uint8_t>(c->tdig) : (uint8_t*)nullptr);
		if (xfer_copy(c, d_keys, hk, n * 8, 0) != 0) rc = -1;
		if (rc == 0 && hd) { if (xfer_copy(c, c->tdig.p, hd, n, 0) != 0) rc = -1; memset(hd + n, 0, 16); }
		tl.lap("walk: packed keys to the host");
		const uint64_t imask = bi >= 64 ? ~0ull : (1ull << bi) - 1;
		if (w && w->n_seg) { // only the wanted reads' stretches come out in the reference's order (host/refsort.c: ma_refsort_packed_wanted), and only they go back up
			if (rc == 0 && ma_refsort_packed_wanted(hk, n, bl, bi, shift_top, hd, w->wcum, w->n_ids) != 0) rc = -1;
			tl.lap("walk: host (wanted reads only)");
			if (rc == 0) {
				const size_t S = w->n_seg;
				size_t tot = 0;
				for (size_t s = 0; s < S; ++s) tot += w->seg_len[s];
				// one block: pos[S] | off[S + 1] | ord[tot]
				std::vector<uint32_t> blk(2 * S + 1 + tot);
				uint32_t *pos = blk.data(), *off = pos + S, *ord = off + S + 1;
				size_t o = 0;
				for (size_t s = 0; s < S; ++s) {
					pos[s] = w->seg_pos[s]; off[s] = (uint32_t)o;
					const uint64_t *src = hk + w->seg_pos[s];
					for (uint32_t j = 0; j < w->seg_len[s]; ++j) ord[o++] = (uint32_t)(src[j] & imask);
				}
				off[S] = (uint32_t)o;
				if (dev_reserve(c, c->wseg, blk.size() * 4 + 64) != 0) rc = -1;
				if (rc == 0 && xfer_copy(c, c->wseg.p, blk.data(), blk.size() * 4, 1) != 0) rc = -1;
				if (rc == 0) {
					const uint32_t *d = (const uint32_t*)c->wseg.p;
					hipLaunchKernelGGL(k_perm_patch, dim3((unsigned)S), dim3(256), 0, c->st, d, d + S, d + 2 * S + 1, d_perm);
					HIPCHK(hipStreamSynchronize(c->st)); // (blk goes away)
				}
			}
			tl.lap("walk: the wanted stretches to the device");
		} else {
			if (rc == 0 && ma_refsort_packed(hk, n, bl, bi, shift_top, hd) != 0) rc = -1;
			tl.lap("walk: host");
			if (rc == 0 && xfer_copy(c, d_keys, hk, n * 8, 1) != 0) rc = -1;
			if (rc == 0) hipLaunchKernelGGL(k_perm_from_packed, dim3(grid_for(n, 256)), dim3(256), 0, c->st, (const uint64_t*)d_keys, n, imask, d_perm);
			tl.lap("walk: order to the device");
		}
	} else { // keys too wide (or MA_REFSORT_KEYS, for the tests): the raw keys go down, the host packs what it can
		uint64_t *hk = (uint64_t*)malloc(n * 8);
		uint32_t *hp = (uint32_t*)malloc(n * 4);
		if (!hk || !hp) { free(hk); free(hp); mahip_set_error("reference_order: out of host memory"); return -1; }
		if (xfer_copy(c, d_keys, hk, n * 8, 0) != 0) rc = -1;
		tl.lap("walk: keys to the host");
		if (rc == 0 && ma_refsort_perm(hk, n, hp) != 0) rc = -1;
		tl.lap("walk: host");
		if (rc == 0 && xfer_copy(c, d_perm, hp, n * 4, 1) != 0) rc = -1;
		tl.lap("walk: permutation to the device");
		free(hk); free(hp);
	}
	HIPCHK(hipGetLastError());
	if (rc) mahip_set_error("reference_order: copy or host sort failed");
	return rc;
}
