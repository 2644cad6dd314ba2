// clean.hip -- the order-dependent graph cleaners (reference asg.c:238-433: asg_cut_tip, asg_cut_internal, asg_cut_biloop,
// asg_pop_bubble) on the device, as a fixpoint over versioned state (see clean_core.h for the method and the proof sketch).
//
//   one iteration = k_clean_rule / k_clean_bubble : every vertex evaluated against the stamps of the previous iteration,
//                                                   the stamps of this iteration rebuilt with atomicMin
//                 + k_clean_diff                  : did any stamp change?
//   after the fixpoint: k_clean_apply writes the stamps into the base flags, asg_cleanup (asg.c:72-80) compacts.
//
// Work units: one thread per vertex for the three short-unitig rules (a rule reads a handful of arc lists); one thread per
// bubble source with a private open-addressing table in HBM scratch (probes touch a handful of vertices; the table grows and
// the iteration is repeated if a probe ever fills it).  Integer work, latency-bound on dependent list reads: the reduced graph
// is small next to the hit arrays, what matters is that nothing leaves the device and that the sweep costs a few launches
// instead of one host step per vertex.
#include "mahip_internal.hpp"
#include "clean_core.h"

int graph_cleanup(mahip_ctx *c); // graph.hip: asg_cleanup on the current graph
uint32_t graph_nseq(const mahip_ctx *c);

enum { CLEAN_TIP = 0, CLEAN_INTERNAL = 1, CLEAN_BILOOP = 2 };

template <int RULE>
__global__ __launch_bounds__(256) void k_clean_rule(cl_view_t g, cl_stamps_t s, int max_ext, unsigned long long *__restrict__ ctr)
{
	uint32_t cnt = 0;
	for (uint32_t v = blockIdx.x * 256 + threadIdx.x; v < g.n_vtx; v += gridDim.x * 256) {
		if (RULE == CLEAN_TIP) cnt += cl_rule_tip(&g, s, v, max_ext);
		else if (RULE == CLEAN_INTERNAL) cnt += cl_rule_internal(&g, s, v, max_ext);
		else cnt += cl_rule_biloop(&g, s, v, max_ext);
	}
	blk_add_u64(&ctr[CT_LIVE], cnt);
}

// vertices that can be a bubble source at all: at least two arcs in the index (asg.c:366)
__global__ __launch_bounds__(256) void k_bubble_cand(const unsigned long long *__restrict__ idx, uint32_t n_vtx, uint32_t *__restrict__ keep)
{
	uint32_t v = blockIdx.x * 256 + threadIdx.x;
	if (v < n_vtx) keep[v] = (uint32_t)idx[v] >= 2;
}
__global__ __launch_bounds__(256) void k_bubble_list(const uint32_t *__restrict__ keep, const uint32_t *__restrict__ pos, uint32_t n_vtx, uint32_t *__restrict__ list)
{
	uint32_t v = blockIdx.x * 256 + threadIdx.x;
	if (v < n_vtx && keep[v]) list[pos[v]] = v;
}

// src[0..n_src): the sources this launch probes.  A probe that fills its table leaves the source in ovf[] (any order: the stamps are
// minima, the counts are sums) for a later launch with fewer threads and bigger tables; everybody else's stamps are final for this
// iteration.
__global__ __launch_bounds__(64) void k_clean_bubble(cl_view_t g, cl_stamps_t s, const uint32_t *__restrict__ src, uint32_t n_src, uint32_t max_dist,
                                                      cl_binfo_t *__restrict__ tabs, uint32_t *__restrict__ aux, uint32_t cap, uint32_t n_tab, uint32_t *__restrict__ ovf,
                                                      unsigned long long *__restrict__ ctr)
{
	const uint32_t tid = blockIdx.x * 64 + threadIdx.x;
	if (tid >= n_tab) return; // one table per working thread (the big tiers have fewer tables than a wave has lanes)
	cl_bscratch_t b;
	b.tab = tabs + (size_t)tid * cap; b.used = aux + (size_t)tid * 2 * cap; b.stack = b.used + cap; b.cap = cap; b.n_used = 0;
	uint32_t pops = 0, tips = 0, back = 0;
	for (uint32_t k = tid; k < n_src; k += n_tab) {
		const uint32_t v0 = src[k];
		uint32_t sink = 0, nt = 0;
		int r = cl_bubble_probe(&g, v0, max_dist, &b, &sink, &nt);
		if (r > 0) { back += cl_bubble_stamp(&g, s, v0, sink, &b); ++pops; tips += nt; }
		else if (r < 0) ovf[atomicAdd(&ctr[CT_OVF], 1ull)] = v0;
	}
	if (pops) atomicAdd(&ctr[CT_LIVE], (unsigned long long)pops); // pops are rare: a handful of atomics per launch
	if (tips) atomicAdd(&ctr[CT_REMAIN], (unsigned long long)tips);
	if (back) atomicAdd(&ctr[CT_OVF2], (unsigned long long)back); // pops that would bring a dead read back (clean_core.h: ASSUMPTION)
}

// The tiers above the first: ONE WAVE per source.  A probe that needs a big table is a long chain of dependent look-ups -- per expanded vertex its arcs, per
// arc the target's record and, for a target seen for the first time, the live arcs of its complement (asg.c:388: the in-degree): some 15 loads in a row per
// vertex, 2.3 ms for the longest probe of the 50 M-overlap noisy input, and a launch lasts as long as its longest probe.  Here the lanes fetch what lane 0 is
// going to need -- lane i takes arc i of the vertex: target, length, dead or not, and the in-degree of a target that is not in the table yet -- into LDS, and lane 0
// replays cl_bubble_probe's loop over the staged values (same order, same rules: the table it builds is the one a single thread builds).  The table itself sits in
// LDS while it fits (1 024 slots; the tiers above keep theirs in HBM).  Per vertex: three rounds of loads instead of fifteen.
enum { BW_LDS_CAP = 1024 };
__global__ __launch_bounds__(64) void k_clean_bubble_wave(cl_view_t g, cl_stamps_t s, const uint32_t *__restrict__ src, uint32_t n_src, uint32_t max_dist,
                                                           cl_binfo_t *__restrict__ tabs, uint32_t *__restrict__ aux, uint32_t cap, int in_lds, uint32_t *__restrict__ ovf,
                                                           unsigned long long *__restrict__ ctr)
{
	__shared__ cl_binfo_t l_tab[BW_LDS_CAP];
	__shared__ uint32_t l_used[BW_LDS_CAP], l_stack[BW_LDS_CAP];
	__shared__ uint32_t st_w[64], st_l[64], st_live[64], st_dead[64];
	__shared__ uint32_t sh_v, sh_first, sh_nv, sh_state, sh_n; // sh_state: 0 = walking, 1 = a bubble, 2 = none, 3 = the table is full
	const uint32_t lane = threadIdx.x;
	cl_bscratch_t b;
	if (in_lds) {
		b.tab = l_tab; b.used = l_used; b.stack = l_stack;
		for (uint32_t i = lane; i < cap; i += 64) l_tab[i].key = CL_NONE;
	} else { b.tab = tabs + (size_t)blockIdx.x * cap; b.used = aux + (size_t)blockIdx.x * 2 * cap; b.stack = b.used + cap; } // (kept empty by the probes, initialised once: k_table_init)
	b.cap = cap; b.n_used = 0;
	uint32_t pops = 0, tips_all = 0, back = 0;
	__syncthreads();
	for (uint32_t k = blockIdx.x; k < n_src; k += gridDim.x) {
		const uint32_t v0 = src[k];
		uint32_t n_stack = 0, n_pending = 0, tips = 0, d = 0, c = 0; // lane 0's
		if (lane == 0) {
			const int go = !cl_seq_dead(&g, v0 >> 1, v0) && cl_count(&g, v0) >= 2 && cl_live_out(&g, v0, v0) >= 2; /* asg.c:421-427 */
			if (go) b.stack[n_stack++] = v0;
			sh_state = go ? 0u : 2u;
		}
		__syncthreads();
		while (sh_state == 0) {
			if (lane == 0) {
				const uint32_t v = b.stack[--n_stack];
				d = c = 0;
				if (v != v0) { cl_binfo_t *tv = cl_bfind(&b, v); d = tv->d; c = tv->c; tv->fl |= CL_B_EXPANDED; }
				sh_v = v; sh_first = cl_first(&g, v); sh_nv = cl_count(&g, v);
			}
			__syncthreads();
			const uint32_t v = sh_v, first = sh_first, nv = sh_nv;
			for (uint32_t base = 0; base < nv; base += 64) {
				const uint32_t i = base + lane;
				if (i < nv) { // what lane 0 is going to ask about arc i
					const uint32_t e = first + i, w = g.av[e];
					const uint32_t dead = (uint32_t)cl_arc_dead(&g, e, v0);
					uint32_t live = 0;
					if (w != v0 && !dead && cl_bfind(&b, w) == 0) live = cl_live_out(&g, w ^ 1, v0);
					st_w[lane] = w; st_l[lane] = g.alen[e]; st_dead[lane] = dead; st_live[lane] = live;
				}
				__syncthreads();
				if (lane == 0) {
					const uint32_t m = nv - base < 64 ? nv - base : 64;
					for (uint32_t j = 0; j < m; ++j) { // cl_bubble_probe's loop body on the staged values
						const uint32_t w = st_w[j], l = st_l[j];
						cl_binfo_t *t;
						int fresh = 0;
						if (w == v0) { sh_state = 2; break; }
						if (st_dead[j]) continue;
						if (d + l > max_dist) { sh_state = 2; break; }
						t = cl_bget(&b, w, &fresh);
						if (t == 0) { sh_state = 3; break; }
						if (fresh) {
							t->p = v; t->d = d + l;
							t->r = st_live[j];
							++n_pending;
						} else {
							if (c + 1 > t->c || (c + 1 == t->c && d + l > t->d)) t->p = v;
							if (c + 1 > t->c) t->c = c + 1;
							if (d + l < t->d) t->d = d + l;
						}
						if (--t->r == 0) {
							if (cl_count(&g, w)) b.stack[n_stack++] = w;
							else t->fl |= CL_B_TIP, ++tips;
							--n_pending;
						}
					}
				}
				__syncthreads();
				if (sh_state != 0) break;
			}
			if (lane == 0 && sh_state == 0) {
				if (n_stack == 0) sh_state = 2;
				else if (!(n_stack > 1 || n_pending)) sh_state = 1;
			}
			__syncthreads();
		}
		const uint32_t state = sh_state;
		if (state == 1) {
			if (lane == 0) { back += cl_bubble_stamp(&g, s, v0, b.stack[0], &b); ++pops; tips_all += tips; } // (leaves the table empty)
		} else {
			if (lane == 0) { sh_n = b.n_used; b.n_used = 0; if (state == 3) ovf[atomicAdd(&ctr[CT_OVF], 1ull)] = v0; }
			__syncthreads();
			for (uint32_t i = lane; i < sh_n; i += 64) b.tab[b.used[i]].key = CL_NONE;
		}
		__syncthreads();
	}
	if (lane == 0) {
		if (pops) atomicAdd(&ctr[CT_LIVE], (unsigned long long)pops);
		if (tips_all) atomicAdd(&ctr[CT_REMAIN], (unsigned long long)tips_all);
		if (back) atomicAdd(&ctr[CT_OVF2], (unsigned long long)back);
	}
}

__global__ __launch_bounds__(256) void k_table_init(cl_binfo_t *__restrict__ tabs, size_t n)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) tabs[i].key = CL_NONE;
}

__global__ __launch_bounds__(256) void k_clean_diff(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, size_t n, unsigned long long *__restrict__ ctr)
{
	uint32_t d = 0;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d += a[i] != b[i];
	blk_add_u64(&ctr[CT_TOTDP], d);
}

// the fixpoint's stamps into the base flags: seq.del for stamped reads, the del bit for stamped arcs
__global__ __launch_bounds__(256) void k_clean_apply(const uint32_t *__restrict__ rst, uint32_t n_read, uint8_t *__restrict__ sdel,
                                                      const uint32_t *__restrict__ ast, size_t n_arc, uint32_t *__restrict__ aol)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n_read && rst[i] != CL_NONE) sdel[i] = 1;
	if (i < n_arc && ast[i] != CL_NONE) aol[i] |= CL_ADEL;
}

// the reference's own sweep on ONE lane (clean_core.h: cl_bubble_sweep_seq): for graphs outside the fixpoint's contract, and for MA_BUBBLE_SEQ=1
__global__ __launch_bounds__(64) void k_clean_bubble_seq(const uint32_t *au, const uint32_t *av, const uint32_t *alen, uint32_t *aol, const unsigned long long *idx, uint8_t *sdel, uint32_t n_vtx,
                                                          uint32_t max_dist, cl_seqinfo_t *info, uint32_t *stk, uint32_t *seen, uint32_t *walked, unsigned long long *ctr)
{
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	unsigned long long pops = 0, tips = 0;
	const int rc = cl_bubble_sweep_seq(au, av, alen, aol, idx, sdel, n_vtx, max_dist, info, stk, seen, walked, &pops, &tips);
	ctr[CT_LIVE] = pops; ctr[CT_REMAIN] = tips; ctr[CT_OVF] = rc != 0;
}

// bubble tables: tier 0 = one small table per thread of the full-width launch; tiers above (16x the slots, 1/16 of the threads: the same
// bytes) only ever see the sources that overflowed the tier below
enum { BUB_CAP0 = 16, BUB_THREADS0 = 524288, BUB_TIERS = 5 }; // 16, 1 024, 16 384 .. 4 M slots; 32 B per slot: 256 MiB per tier that is ever needed (tier 1: LDS)
struct BubTier { DevBuf tabs, aux; bool ready = false; };
struct CleanBufs { DevBuf st[2], src, ovf[2], seq; BubTier tier[BUB_TIERS]; uint32_t max_tier = 0; uint32_t n_seq_sweeps = 0; }; // st[k]: read stamps [R rounded up] then arc stamps [A]

static CleanBufs *clean_bufs(mahip_ctx *c)
{
	if (!c->clean) c->clean = new CleanBufs();
	return (CleanBufs*)c->clean;
}

void clean_free(mahip_ctx *c)
{
	CleanBufs *b = (CleanBufs*)c->clean;
	if (!b) return;
	DevBuf *all[] = { &b->st[0], &b->st[1], &b->src, &b->ovf[0], &b->ovf[1], &b->seq };
	for (DevBuf *x : all) dev_free(c, *x);
	for (BubTier &t : b->tier) { dev_free(c, t.tabs); dev_free(c, t.aux); }
	delete b;
	c->clean = nullptr;
}

static inline uint32_t bub_cap(int tier)
{
	// Tier 0 is small (most probes see a handful of vertices) and wide; tier 1 is what fits into LDS (the wave form); above it 16 x per tier.
	// MA_BUBBLE_CAP0 (a power of two >= 4; MA_BUBBLE_CAP1 likewise): measurements, and the tests shrink the tiers so that small graphs reach the ones above.
	static uint32_t base = 0, one = 0;
	if (!base) {
		const char *e = getenv("MA_BUBBLE_CAP0"), *f = getenv("MA_BUBBLE_CAP1");
		base = e && atoi(e) >= 4 && !(atoi(e) & (atoi(e) - 1)) ? (uint32_t)atoi(e) : (uint32_t)BUB_CAP0;
		one = f && atoi(f) >= 8 && !(atoi(f) & (atoi(f) - 1)) ? (uint32_t)atoi(f) : e ? base << 4 : (uint32_t)BW_LDS_CAP;
		if (one <= base) one = base << 1;
	}
	return tier == 0 ? base : one << (4 * (tier - 1));
}
static inline bool bub_thread_tiers() { static int v = -1; if (v < 0) v = getenv("MA_BUBBLE_THREAD_TIERS") != nullptr; return v != 0; } // measurements / tests: a thread per source in every tier
static inline uint32_t bub_lds_cap() { static long v = -1; if (v < 0) { const char *e = getenv("MA_BUBBLE_LDS_CAP"); v = e ? atol(e) : (long)BW_LDS_CAP; if (v > (long)BW_LDS_CAP) v = BW_LDS_CAP; } return (uint32_t)v; } // tests: smaller, so that small inputs reach the tables in HBM
static inline unsigned bub_threads(int tier, uint32_t n_src)
{
	static unsigned base = 0; // MA_BUBBLE_THREADS0: tier 0's launch width (measurements)
	if (!base) { const char *e = getenv("MA_BUBBLE_THREADS0"); base = e && atoi(e) >= 64 ? (unsigned)atoi(e) : (unsigned)BUB_THREADS0; }
	const unsigned most = base >> (4 * tier) ? base >> (4 * tier) : 1u; // 65536, 4096, 256, 16, 1
	return n_src < most ? n_src : most;
}

// one launch of the bubble probes over src[0..n_src) with the tables of `tier`; sources that need more room are appended to ovf
static int bubble_launch(mahip_ctx *c, CleanBufs *b, int tier, const cl_view_t &g, const cl_stamps_t &s, const uint32_t *src, uint32_t n_src, uint32_t max_dist, uint32_t *ovf)
{
	BubTier &t = b->tier[tier];
	const uint32_t cap = bub_cap(tier);
	const bool wave = tier > 0 && !bub_thread_tiers(); // a wave per source (k_clean_bubble_wave); its table in LDS while it fits
	const bool lds = wave && cap <= bub_lds_cap();
	const unsigned threads = lds ? (n_src < 8192u ? n_src : 8192u) : bub_threads(tier, n_src); // = tables = (wave form) blocks
	const size_t slots = lds ? 0 : (size_t)threads * cap;
	if (t.tabs.cap < slots * sizeof(cl_binfo_t)) {
		CHK(dev_reserve(c, t.tabs, slots * sizeof(cl_binfo_t)));
		CHK(dev_reserve(c, t.aux, slots * 2 * 4));
		t.ready = false;
	}
	if (!t.ready && slots) { // probes leave their table empty (also the ones that give up): initialise once per allocation
		const size_t all = t.tabs.cap / sizeof(cl_binfo_t);
		hipLaunchKernelGGL(k_table_init, dim3(grid_for(all, 256)), dim3(256), 0, c->st, (cl_binfo_t*)t.tabs.p, all);
		t.ready = true;
	}
	ProfScope ps(c, "k_clean_bubble", 0);
	if (wave) hipLaunchKernelGGL(k_clean_bubble_wave, dim3(threads), dim3(64), 0, c->st, g, s, src, n_src, max_dist, (cl_binfo_t*)t.tabs.p, P<uint32_t>(t.aux), cap, lds ? 1 : 0, ovf, P<unsigned long long>(c->ctr));
	else hipLaunchKernelGGL(k_clean_bubble, dim3((threads + 63) / 64), dim3(64), 0, c->st, g, s, src, n_src, max_dist, (cl_binfo_t*)t.tabs.p, P<uint32_t>(t.aux), cap, threads, ovf, P<unsigned long long>(c->ctr));
	if ((uint32_t)tier > b->max_tier) b->max_tier = tier;
	return 0;
}

// asg_pop_bubble exactly as the reference runs it, on one lane, on the base flags (no stamps): *cnt / *cnt2 as clean_sweep
static int bubble_sweep_seq(mahip_ctx *c, CleanBufs *b, uint32_t max_dist, uint32_t *cnt, uint32_t *cnt2)
{
	const uint32_t V = 2 * graph_nseq(c);
	const size_t A = c->n_arc, words = (size_t)V * 4 + (size_t)V * 2 + A + 16;
	CHK(dev_reserve(c, b->seq, words * 4));
	HIPCHK(hipMemsetAsync(b->seq.p, 0, (size_t)V * 16, c->st)); // info[]: nothing visited
	cl_seqinfo_t *info = (cl_seqinfo_t*)b->seq.p;
	uint32_t *stk = (uint32_t*)b->seq.p + (size_t)V * 4, *seen = stk + V, *walked = seen + V;
	const int ag = c->ag;
	CHK(ctr_zero(c));
	{
		ProfScope ps(c, "k_clean_bubble_seq", 0);
		hipLaunchKernelGGL(k_clean_bubble_seq, dim3(1), dim3(64), 0, c->st, (const uint32_t*)P<uint32_t>(c->au[ag]), (const uint32_t*)P<uint32_t>(c->av[ag]), (const uint32_t*)P<uint32_t>(c->alen[ag]),
		                   P<uint32_t>(c->aol[ag]), (const unsigned long long*)P<unsigned long long>(c->idx), P<uint8_t>(c->sdel), V, max_dist, info, stk, seen, walked, P<unsigned long long>(c->ctr));
	}
	CHK(ctr_fetch(c));
	HIPCHK(hipGetLastError());
	++b->n_seq_sweeps;
	if (c->h_ctr[CT_OVF]) { mahip_set_error("asg_pop_bubble: more walks into a vertex than it has arcs in -- the graph is not symmetric, and the reference's assertion (asg.c:391) ends its run here too"); return -1; }
	*cnt = (uint32_t)c->h_ctr[CT_LIVE]; *cnt2 = (uint32_t)c->h_ctr[CT_REMAIN];
	if (*cnt) { c->arcs_clean = false; CHK(ctr_zero(c)); CHK(graph_cleanup(c)); } // asg.c:430 (reads were deleted: the cleanup looks at seq.del again)
	return 0;
}
extern "C" uint32_t mahip_bubble_seq_sweeps(mahip_ctx_t *c) { return c->clean ? ((CleanBufs*)c->clean)->n_seq_sweeps : 0; } // tests: did a call take the sequential road?

// mode 0..2: the short-unitig rules with param = max_ext; mode 3: bubbles with param = max_dist.
// *cnt = actions of the sweep (tips cut / internal sequences / bi-loops / bubbles), *cnt2 = tips trimmed by bubble pops.
static int clean_sweep(mahip_ctx *c, int mode, int param, uint32_t *cnt, uint32_t *cnt2, int *n_iter)
{
	HIPCHK(hipSetDevice(c->dev));
	if (!c->graph_ready) { mahip_set_error("graph cleaner: no graph"); return -1; }
	CleanBufs *b = clean_bufs(c);
	const uint32_t R = graph_nseq(c), V = 2 * R;
	const size_t A = c->n_arc;
	*cnt = *cnt2 = 0; if (n_iter) *n_iter = 0;
	if (V == 0) return 0; // (no arcs at all is still a graph: every read without arcs is a tip, asg.c:238-254)
	const size_t Rp = ((size_t)R + 63) & ~(size_t)63, W = Rp + A; // words per stamp set
	for (int k = 0; k < 2; ++k) CHK(dev_reserve(c, b->st[k], (W + 4) * 4));
	unsigned long long *ctr = P<unsigned long long>(c->ctr);
	uint32_t n_src = 0;
	if (mode == 3) { // source list: vertices with >= 2 arcs, in vertex order
		if (A == 0) return 0;
		CHK(dev_reserve(c, c->keep, ((size_t)V + 16) * 4)); CHK(dev_reserve(c, c->pos, ((size_t)V + 16) * 4));
		CHK(dev_reserve(c, b->src, ((size_t)V + 4) * 4));
		for (int k = 0; k < 2; ++k) CHK(dev_reserve(c, b->ovf[k], ((size_t)V + 4) * 4));
		uint32_t *d_tot = (uint32_t*)(ctr + CT_TOTAL);
		hipLaunchKernelGGL(k_bubble_cand, dim3(grid_for(V, 256)), dim3(256), 0, c->st, (const unsigned long long*)P<unsigned long long>(c->idx), V, P<uint32_t>(c->keep));
		CHK(scan_exclusive_u32(c, P<uint32_t>(c->keep), P<uint32_t>(c->pos), V, d_tot));
		hipLaunchKernelGGL(k_bubble_list, dim3(grid_for(V, 256)), dim3(256), 0, c->st, (const uint32_t*)P<uint32_t>(c->keep), (const uint32_t*)P<uint32_t>(c->pos), V, P<uint32_t>(b->src));
		CHK(ctr_fetch(c));
		n_src = (uint32_t)(c->h_ctr[CT_TOTAL] & 0xffffffffu);
		if (n_src == 0) return 0;
		static const bool force_seq = getenv("MA_BUBBLE_SEQ") != nullptr;
		if (force_seq) { if (n_iter) *n_iter = 1; return bubble_sweep_seq(c, b, (uint32_t)param, cnt, cnt2); }
	}
	cl_view_t g;
	const int ag = c->ag;
	g.av = P<uint32_t>(c->av[ag]); g.alen = P<uint32_t>(c->alen[ag]); g.aol = P<uint32_t>(c->aol[ag]);
	g.idx = P<unsigned long long>(c->idx); g.sdel = P<uint8_t>(c->sdel); g.n_vtx = V;
	int cur = 0;
	for (int it = 0;; ++it) {
		if (it > 100000) { mahip_set_error("graph cleaner: no fixpoint"); return -1; }
		cl_stamps_t s; s.rst = P<uint32_t>(b->st[cur ^ 1]); s.ast = s.rst + Rp;
		g.rst = P<uint32_t>(b->st[cur]); g.ast = g.rst + Rp;
		g.no_stamps = it == 0; // the first sweep sees the base graph: the old stamp set is neither initialised nor read
		HIPCHK(hipMemsetAsync(s.rst, 0xff, W * 4, c->st));
		CHK(ctr_zero(c));
		if (mode == 3) {
			CHK(bubble_launch(c, b, 0, g, s, P<uint32_t>(b->src), n_src, (uint32_t)param, P<uint32_t>(b->ovf[0])));
		} else {
			ProfScope ps(c, "k_clean_rule", 0);
			const unsigned grid = grid_for(V, 256, MA_STREAM_BLOCKS);
			if (mode == CLEAN_TIP) hipLaunchKernelGGL(k_clean_rule<CLEAN_TIP>, dim3(grid), dim3(256), 0, c->st, g, s, param, ctr);
			else if (mode == CLEAN_INTERNAL) hipLaunchKernelGGL(k_clean_rule<CLEAN_INTERNAL>, dim3(grid), dim3(256), 0, c->st, g, s, param, ctr);
			else hipLaunchKernelGGL(k_clean_rule<CLEAN_BILOOP>, dim3(grid), dim3(256), 0, c->st, g, s, param, ctr);
		}
		if (it > 0) hipLaunchKernelGGL(k_clean_diff, dim3(grid_for(W, 256, 1024)), dim3(256), 0, c->st, g.rst, (const uint32_t*)s.rst, W, ctr);
		CHK(ctr_fetch(c));
		HIPCHK(hipGetLastError());
		if (mode == 3 && c->h_ctr[CT_OVF]) { // some probes filled their tables: those sources again, tier by tier, then the comparison once more
			const unsigned long long pops = c->h_ctr[CT_LIVE], tips = c->h_ctr[CT_REMAIN], back = c->h_ctr[CT_OVF2];
			unsigned long long more_pops = 0, more_tips = 0, more_back = 0;
			uint32_t n_ovf = (uint32_t)c->h_ctr[CT_OVF];
			for (int tier = 1, w = 0; n_ovf; ++tier, w ^= 1) {
				if (tier >= BUB_TIERS) { mahip_set_error("asg_pop_bubble: a probe visits more than %u vertices", bub_cap(BUB_TIERS - 1) / 4 * 3); return -1; }
				CHK(ctr_zero(c));
				CHK(bubble_launch(c, b, tier, g, s, P<uint32_t>(b->ovf[w]), n_ovf, (uint32_t)param, P<uint32_t>(b->ovf[w ^ 1])));
				CHK(ctr_fetch(c));
				more_pops += c->h_ctr[CT_LIVE]; more_tips += c->h_ctr[CT_REMAIN]; more_back += c->h_ctr[CT_OVF2];
				n_ovf = (uint32_t)c->h_ctr[CT_OVF];
			}
			CHK(ctr_zero(c));
			if (it > 0) hipLaunchKernelGGL(k_clean_diff, dim3(grid_for(W, 256, 1024)), dim3(256), 0, c->st, g.rst, (const uint32_t*)s.rst, W, ctr);
			CHK(ctr_fetch(c));
			c->h_ctr[CT_LIVE] = pops + more_pops; c->h_ctr[CT_REMAIN] = tips + more_tips; c->h_ctr[CT_OVF2] = back + more_back;
		}
		cur ^= 1;
		if (n_iter) *n_iter = it + 1;
		// fixpoint: the stamps did not change.  After the first sweep: no action means no stamp (an action may also be a no-op: counted, asg.c:296-302)
		if (it == 0 ? c->h_ctr[CT_LIVE] == 0 : c->h_ctr[CT_TOTDP] == 0) {
			if (mode == 3 && c->h_ctr[CT_OVF2]) // the final view holds a pop that brings a dead read back (clean_core.h: ASSUMPTION): outside the fixpoint's contract --
				return bubble_sweep_seq(c, b, (uint32_t)param, cnt, cnt2); // the base flags are untouched so far: the reference's sequential sweep, on one lane
			*cnt = (uint32_t)c->h_ctr[CT_LIVE]; *cnt2 = (uint32_t)c->h_ctr[CT_REMAIN]; break;
		}
	}
	if (*cnt) {
		const size_t m = A > R ? A : R;
		const uint32_t *fin = P<uint32_t>(b->st[cur]);
		hipLaunchKernelGGL(k_clean_apply, dim3(grid_for(m, 256)), dim3(256), 0, c->st, fin, R, P<uint8_t>(c->sdel), fin + Rp, A, P<uint32_t>(c->aol[ag]));
		c->arcs_clean = false; // reads were deleted: the cleanup has to look at seq.del again
		CHK(ctr_zero(c));
		CHK(graph_cleanup(c)); // asg.c:251, 269, 303, 430: asg_cleanup when something was cut
	}
	return 0;
}

static void clean_timing(const char *what, int n_iter, uint32_t cnt, int tier = -1)
{
	static int on = -1;
	if (on < 0) on = getenv("MA_PIPE_TIMING") && atoi(getenv("MA_PIPE_TIMING")) >= 2;
	if (!on) return;
	if (tier < 0) fprintf(stderr, "[T::clean] %-14s %2d iterations, %u actions\n", what, n_iter, cnt);
	else fprintf(stderr, "[T::clean] %-14s %2d iterations, %u actions; biggest probe table used so far: %u slots (tier %d)\n", what, n_iter, cnt, bub_cap(tier), tier);
}

extern "C" int mahip_asg_cut_tip(mahip_ctx_t *c, int max_ext, uint32_t *n_cut)
{
	uint32_t a = 0, b = 0; int it = 0;
	CHK(clean_sweep(c, CLEAN_TIP, max_ext, &a, &b, &it));
	clean_timing("cut_tip", it, a);
	if (n_cut) *n_cut = a;
	return 0;
}
extern "C" int mahip_asg_cut_internal(mahip_ctx_t *c, int max_ext, uint32_t *n_cut)
{
	uint32_t a = 0, b = 0; int it = 0;
	CHK(clean_sweep(c, CLEAN_INTERNAL, max_ext, &a, &b, &it));
	clean_timing("cut_internal", it, a);
	if (n_cut) *n_cut = a;
	return 0;
}
extern "C" int mahip_asg_cut_biloop(mahip_ctx_t *c, int max_ext, uint32_t *n_cut)
{
	uint32_t a = 0, b = 0; int it = 0;
	CHK(clean_sweep(c, CLEAN_BILOOP, max_ext, &a, &b, &it));
	clean_timing("cut_biloop", it, a);
	if (n_cut) *n_cut = a;
	return 0;
}
extern "C" int mahip_asg_pop_bubble(mahip_ctx_t *c, int max_dist, uint32_t *n_pop, uint32_t *n_tips)
{
	uint32_t a = 0, b = 0; int it = 0;
	CHK(clean_sweep(c, 3, max_dist, &a, &b, &it));
	clean_timing("pop_bubble", it, a, c->clean ? (int)((CleanBufs*)c->clean)->max_tier : 0);
	if (n_pop) *n_pop = a;
	if (n_tips) *n_tips = b;
	return 0;
}
