// scan.hip -- device-wide exclusive prefix sum (u32).
// Used for: radix-sort digit offsets, order-preserving compaction of hits and arcs, the squeeze map
// (reference sdict.c:69-86), CSR offsets.
//
// Arrays of up to 256 tiles (half a million elements: every scan of the graph phase, where a pass makes eleven of them on a few thousand elements
// each) take ONE launch: tiles are handed out by an atomic ticket (a tile's predecessors have therefore started and will publish without waiting
// for anybody behind them), a tile publishes its sum, looks back over its predecessors' published words (64 at a time, one per lane of a wave)
// until it meets one that already knows its inclusive prefix, and publishes its own.  A published word = launch epoch | state | value in 64 bits,
// written and read with one relaxed agent-scope atomic, so nothing has to be cleared between launches and nothing has to be fenced.  The round-2
// form cost five launches, 25 us, on such arrays.  Bigger arrays keep the three-phase form (reduce / scan of the tile sums / downsweep): there the
// launches do not matter, and the input is read twice from a cache that mostly still holds it.
#include "mahip_internal.hpp"

#define SCAN_THREADS 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_THREADS * SCAN_ITEMS)

// block-wide exclusive scan of one value per thread (256 threads = 4 waves); returns the exclusive
// prefix for this thread and the block total in *total
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t x, uint32_t *s_wave, uint32_t *total)
{
	unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t incl = (uint32_t)wv_scan_incl_i32((int)x, lane); // full waves: DPP row shifts + row broadcasts
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

// (SC_AGG / SC_INCL / sc_pack / SC_PUBLISH / SC_PEEK and the look-back itself, sc_look_back: mahip_internal.hpp -- graph.hip's one-pass arc compaction chains its tiles the same way)
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_chain(const uint32_t *in, uint32_t *out, // may alias (in-place)
                                                              size_t n, uint32_t *d_total, unsigned long long *state, uint32_t *ticket, uint32_t ticket_base, uint32_t epoch)
{
	__shared__ uint32_t s_wave[SCAN_THREADS / 64];
	__shared__ uint32_t s_tile, s_prefix;
	if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u) - ticket_base;
	__syncthreads();
	const uint32_t tile = s_tile;
	const size_t base = (size_t)tile * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
	uint32_t v[SCAN_ITEMS], s = 0, tot;
	if (base + SCAN_ITEMS <= n) {
		const uint4 *p = (const uint4*)(in + base);
		uint4 a = p[0], b = p[1];
		v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
	} else {
		for (int i = 0; i < SCAN_ITEMS; ++i) v[i] = base + i < n ? in[base + i] : 0;
	}
	for (int i = 0; i < SCAN_ITEMS; ++i) s += v[i];
	const uint32_t ex = block_excl_scan(s, s_wave, &tot);
	if (threadIdx.x == 0) {
		SC_PUBLISH(&state[tile], sc_pack(epoch, tile == 0 ? SC_INCL : SC_AGG, tot));
		if (tile == 0) s_prefix = 0;
	}
	if (tile > 0 && threadIdx.x < 64) { // look back
		const uint32_t prefix = sc_look_back(state, tile, epoch, threadIdx.x);
		if (threadIdx.x == 0) { s_prefix = prefix; SC_PUBLISH(&state[tile], sc_pack(epoch, SC_INCL, prefix + tot)); }
	}
	__syncthreads();
	uint32_t run = ex + s_prefix;
	uint32_t o[SCAN_ITEMS];
	for (int i = 0; i < SCAN_ITEMS; ++i) { o[i] = run; run += v[i]; }
	if (base + SCAN_ITEMS <= n) {
		uint4 *q = (uint4*)(out + base);
		q[0] = make_uint4(o[0], o[1], o[2], o[3]);
		q[1] = make_uint4(o[4], o[5], o[6], o[7]);
	} else {
		for (int i = 0; i < SCAN_ITEMS; ++i) if (base + i < n) out[base + i] = o[i];
	}
	if (d_total && base < n && base + SCAN_ITEMS >= n) *d_total = run; // grand total = exclusive prefix + value of the last element
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
	if (level >= 2) { mahip_set_error("scan: input too large"); return -1; }
	CHK(dev_reserve(c, c->scan_tmp[level + 1], (nb + 8) * 4));
	uint32_t *bs = P<uint32_t>(c->scan_tmp[level + 1]);
	hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, c->st, in, bs, n);
	CHK(scan_rec(c, bs, bs, nb, nullptr, level + 1));
	hipLaunchKernelGGL(k_scan_down, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, c->st, in, out, (const uint32_t*)bs, n, d_total);
	return 0;
}

// tiles up to which the one-launch chained scan is used (MA_SCAN_CHAIN_MAX); beyond it the launches of the three-phase scan do not matter and its
// traffic pattern is the safer one
static size_t scan_chain_max() { static long v = -1; if (v < 0) { const char *e = getenv("MA_SCAN_CHAIN_MAX"); v = e ? atol(e) : 256; } return (size_t)v; }

// the published words + ticket of ONE chained launch of nb tiles (k_scan_chain, graph.hip: k_arc_rm_chain): the tiles behind one ticket word; fresh memory
// is cleared once, afterwards the launch epoch tells this launch's words from older ones
int scan_chain_begin(mahip_ctx *c, size_t nb, unsigned long long **state, uint32_t **ticket, uint32_t *ticket_base, uint32_t *epoch, size_t n_tickets)
{ // nb: published words (tiles); n_tickets: blocks that draw a ticket (0: one per tile; a block that works through several consecutive tiles draws one for all of them)
	if (c->scan_tmp[0].cap < (nb + 8) * 8 + 64) { // epoch 0 is never used
		CHK(dev_reserve(c, c->scan_tmp[0], (nb + 8) * 8 * 2 + 64));
		HIPCHK(hipMemsetAsync(c->scan_tmp[0].p, 0, c->scan_tmp[0].cap, c->st));
		c->scan_ticket = 0; c->scan_epoch = 0;
	}
	*ticket = P<uint32_t>(c->scan_tmp[0]);
	*state = (unsigned long long*)((char*)c->scan_tmp[0].p + 64);
	if (++c->scan_epoch >= (1u << 30)) { HIPCHK(hipMemsetAsync(*state, 0, c->scan_tmp[0].cap - 64, c->st)); c->scan_epoch = 1; } // (after 2^30 launches)
	*ticket_base = c->scan_ticket; *epoch = c->scan_epoch;
	c->scan_ticket += (uint32_t)(n_tickets ? n_tickets : nb);
	return 0;
}

int scan_exclusive_u32(mahip_ctx *c, const uint32_t *in, uint32_t *out, size_t n, uint32_t *d_total)
{
	ProfScope ps(c, "scan_exclusive_u32", 8.0 * (double)n);
	if (n == 0) {
		if (d_total) HIPCHK(hipMemsetAsync(d_total, 0, 4, c->st));
		return 0;
	}
	const size_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
	if (nb >= 0x40000000ull) { mahip_set_error("scan: input too large"); return -1; }
	if (nb == 1) { // one tile: nothing to chain
		hipLaunchKernelGGL(k_scan_down, dim3(1), dim3(SCAN_THREADS), 0, c->st, in, out, (const uint32_t*)nullptr, n, d_total);
		HIPCHK(hipGetLastError());
		return 0;
	}
	if (nb > scan_chain_max()) { // a big array: reduce / scan of the tile sums / downsweep
		CHK(scan_rec(c, in, out, n, d_total, 0));
		HIPCHK(hipGetLastError());
		return 0;
	}
	uint32_t *ticket; unsigned long long *state; uint32_t ticket_base, epoch;
	CHK(scan_chain_begin(c, nb, &state, &ticket, &ticket_base, &epoch));
	hipLaunchKernelGGL(k_scan_chain, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, c->st, in, out, n, d_total, state, ticket, ticket_base, epoch);
	HIPCHK(hipGetLastError());
	return 0;
}
