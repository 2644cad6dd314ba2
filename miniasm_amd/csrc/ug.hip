// ug.hip -- unitigs of the cleaned string graph on the device (reference asm.c:121-210 ma_ug_gen).
//
// The reference walks the graph vertex by vertex with a mark array: from the smallest unmarked vertex that has an arc it
// extends forward and backward while both ends of an arc are unambiguous, emits the path as a unitig, marks it.  The same
// result as parallel primitives over the vertex set:
//
//   link        : w -> x is unitig-internal  <=>  out-degree(w) == 1 and out-degree(x^1) == 1       (asm.c:140-143, 155-157)
//   chains      : maximal chains of links = the unitigs, each present in both orientations; list ranking by pointer jumping
//                 gives every vertex its chain head and its offset; a chain without a head is a cycle and is cut at its
//                 smallest vertex (where the reference's sweep enters it)
//   orientation : the reference discovers a unitig at its smallest vertex that has an arc, and emits the orientation that vertex
//                 lies in  ->  per chain the minimum over such vertices; the chain whose minimum beats its twin's is emitted
//   numbering   : unitigs are numbered in order of that discovery vertex -> one exclusive scan over the vertex set
//   members     : a segmented gather (head offset + rank) writes `vertex << 32 | length to the next read`, the last read of a
//                 linear unitig contributes its whole length (asm.c:151-153); the ranking pass carries the running length along,
//                 so a unitig's length is read off its last vertex (no atomics, however long the unitig)
//   unitig arcs : arcs of the string graph that join a unitig end to a unitig start (asm.c:185-207), compacted in arc order
//
// Only the text writer (and the reference-order sort of the handful of unitig arcs) stays on the host.
#include "mahip_internal.hpp"
#include "ug_core.h"

uint32_t graph_nseq(const mahip_ctx *c);

struct UgBufs {
	DevBuf nxt, prv, wt, mn[2], ptr[2], dist[2], ws[2], cm, tail, uid, flag, pos, circ, ishead;
	DevBuf u_head, u_n, u_len, u_start, u_end, u_off, ua, mark, akeep, apos, arcs;
	uint32_t n_utg = 0, n_mem = 0, n_uarc = 0;
};

static UgBufs *ug_bufs(mahip_ctx *c)
{
	if (!c->ug) c->ug = new UgBufs();
	return (UgBufs*)c->ug;
}

void ug_free(mahip_ctx *c)
{
	UgBufs *b = (UgBufs*)c->ug;
	if (!b) return;
	DevBuf *all[] = { &b->nxt, &b->prv, &b->wt, &b->ws[0], &b->ws[1], &b->mn[0], &b->mn[1], &b->ptr[0], &b->ptr[1], &b->dist[0], &b->dist[1], &b->cm, &b->tail, &b->uid, &b->flag, &b->pos, &b->circ, &b->ishead,
		&b->u_head, &b->u_n, &b->u_len, &b->u_start, &b->u_end, &b->u_off, &b->ua, &b->mark, &b->akeep, &b->apos, &b->arcs };
	for (DevBuf *x : all) dev_free(c, *x);
	delete b;
	c->ug = nullptr;
}

__global__ __launch_bounds__(256) void k_ug_link(ug_t a) { uint32_t w = blockIdx.x * 256 + threadIdx.x; if (w < a.n_vtx) ugk_link(&a, w); }
__global__ __launch_bounds__(256) void k_ug_jump_init(ug_t a, ug_rank_t r) { uint32_t w = blockIdx.x * 256 + threadIdx.x; if (w < a.n_vtx) ugk_jump_init(&a, w, r); }
__global__ __launch_bounds__(256) void k_ug_jump(uint32_t n_vtx, ug_rank_t i, ug_rank_t o) { uint32_t w = blockIdx.x * 256 + threadIdx.x; if (w < n_vtx) ugk_jump(w, i, o); }
// does any member sit on a chain without a head (a cycle)?  Cycles are rare: the cut and the second ranking only run when one exists
__global__ __launch_bounds__(256) void k_ug_cycle_check(ug_t a, const uint32_t *ptr, unsigned long long *ctr)
{
	uint32_t w = blockIdx.x * 256 + threadIdx.x;
	int cyc = 0, bad = 0;
	if (w < a.n_vtx) { const uint32_t p = a.prv[w]; cyc = p < UG_OUT && a.prv[ptr[w]] != UG_NONE; bad = ugk_link_bad(&a, w); }
	if (__ballot(cyc) && (threadIdx.x & 63) == 0) atomicAdd(&ctr[CT_OVF], 1ull);
	if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicAdd(&ctr[CT_OVF2], 1ull); // links that are not mirrored: the graph is not symmetric
}
// The reference's own sweep for graphs that are not symmetric (ug_core.h): ONE wave; its lanes look for start vertices 64 at a time,
// lane 0 walks.  Counts keep running past the capacity of ua, so the host can grow it and launch again.
__global__ __launch_bounds__(64) void k_ug_seq(ug_t a, uint8_t *seen, unsigned long long cap, unsigned long long *ctr)
{
	const uint32_t lane = threadIdx.x;
	ug_seq_t s;
	s.seen = seen; s.cap = cap; s.n_mem = 0; s.n_utg = 0; s.err = 0;
	for (uint32_t base = 0; base < a.n_vtx; base += 64) {
		const uint32_t v = base + lane;
		const int cand = v < a.n_vtx && !a.sdel[v >> 1] && ug_deg(&a, v) > 0;
		unsigned long long m = __ballot(cand);
		if (lane == 0)
			for (; m && !s.err; m &= m - 1) {
				const uint32_t s0 = base + (uint32_t)__builtin_ctzll(m);
				if (!seen[s0]) ug_seq_unitig(&a, &s, s0);
			}
		if (__shfl((int)s.err, 0)) break;
	}
	if (lane == 0) { ctr[CT_TOTAL] = s.n_mem; ctr[CT_LIVE] = s.n_utg; ctr[CT_CUT] = s.err; }
}
// asm.c:180-184 in unitig order (with overlapping unitigs a later one overwrites an earlier one's mark)
__global__ void k_ug_mark_seq(ug_t a, uint32_t n_utg) { if (threadIdx.x == 0 && blockIdx.x == 0) for (uint32_t k = 0; k < n_utg; ++k) ugk_mark(&a, k); }

__global__ __launch_bounds__(256) void k_ug_heads(ug_t a, uint8_t *is_head) { uint32_t w = blockIdx.x * 256 + threadIdx.x; if (w < a.n_vtx) is_head[w] = a.prv[w] == UG_NONE; }
__global__ __launch_bounds__(256) void k_ug_cut(ug_t a, const uint32_t *ptr, const uint32_t *mn, const uint8_t *is_head) { uint32_t w = blockIdx.x * 256 + threadIdx.x; if (w < a.n_vtx) ugk_cut(&a, w, ptr, mn, is_head); }
__global__ __launch_bounds__(256) void k_ug_chain(ug_t a, ug_rank_t r) { uint32_t w = blockIdx.x * 256 + threadIdx.x; if (w < a.n_vtx) ugk_chain(&a, w, r); }
__global__ __launch_bounds__(256) void k_ug_pick(ug_t a, const uint32_t *ptr) { uint32_t w = blockIdx.x * 256 + threadIdx.x; if (w < a.n_vtx) ugk_pick(&a, w, ptr); }
__global__ __launch_bounds__(256) void k_ug_units(ug_t a, ug_rank_t r) { uint32_t w = blockIdx.x * 256 + threadIdx.x; if (w < a.n_vtx) ugk_units(&a, w, r); }
__global__ __launch_bounds__(256) void k_ug_fill(ug_t a, const uint32_t *ptr, const uint32_t *dist) { uint32_t w = blockIdx.x * 256 + threadIdx.x; if (w < a.n_vtx) ugk_fill(&a, w, ptr, dist); }
__global__ __launch_bounds__(256) void k_ug_mark(ug_t a, uint32_t n_utg) { uint32_t k = blockIdx.x * 256 + threadIdx.x; if (k < n_utg) ugk_mark(&a, k); }
__global__ __launch_bounds__(256) void k_ug_arc_keep(ug_t a, size_t n, uint32_t *keep) { size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; if (e < n) keep[e] = ugk_arc_keep(&a, e); }
__global__ __launch_bounds__(256) void k_ug_arc_emit(ug_t a, size_t n, const uint32_t *keep, const uint32_t *pos, asg_arc_t *out)
{
	size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (e < n && keep[e]) { uint32_t o[4]; ugk_arc_emit(&a, e, o); *(uint4*)(out + pos[e]) = make_uint4(o[0], o[1], o[2], o[3]); }
}

static int bitlen32(uint32_t x) { int b = 0; while (x) ++b, x >>= 1; return b; }

static ug_rank_t rank_bufs(UgBufs *b, int g)
{
	ug_rank_t r;
	r.ptr = P<uint32_t>(b->ptr[g]); r.mn = P<uint32_t>(b->mn[g]); r.dist = P<uint32_t>(b->dist[g]); r.ws = P<uint32_t>(b->ws[g]);
	return r;
}

// list ranking of every member along prv by pointer jumping: afterwards ptr = chain head (or, on a cycle, some cycle vertex) and over the
// stretch head..w: mn = smallest vertex with an arc, dist = links, ws = length contributions
static int ug_rank(mahip_ctx *c, UgBufs *b, const ug_t &a, int *gen_out)
{
	const uint32_t V = a.n_vtx;
	int g = 0;
	hipLaunchKernelGGL(k_ug_jump_init, dim3(grid_for(V, 256)), dim3(256), 0, c->st, a, rank_bufs(b, 0));
	for (int k = bitlen32(V) + 1; k > 0; --k, g ^= 1)
		hipLaunchKernelGGL(k_ug_jump, dim3(grid_for(V, 256)), dim3(256), 0, c->st, V, rank_bufs(b, g), rank_bufs(b, g ^ 1));
	*gen_out = g;
	return 0;
}

extern "C" int mahip_ug_gen(mahip_ctx_t *c, uint32_t *n_utg, uint32_t *n_members, uint32_t *n_uarc)
{
	HIPCHK(hipSetDevice(c->dev));
	if (!c->graph_ready) { mahip_set_error("mahip_ug_gen: no graph"); return -1; }
	UgBufs *b = ug_bufs(c);
	const uint32_t R = graph_nseq(c), V = 2 * R;
	const size_t A = c->n_arc;
	b->n_utg = b->n_mem = b->n_uarc = 0;
	if (n_utg) *n_utg = 0;
	if (n_members) *n_members = 0;
	if (n_uarc) *n_uarc = 0;
	if (V == 0) return 0;
	const size_t vb = ((size_t)V + 16) * 4;
	DevBuf *vbufs[] = { &b->nxt, &b->prv, &b->wt, &b->ws[0], &b->ws[1], &b->mn[0], &b->mn[1], &b->ptr[0], &b->ptr[1], &b->dist[0], &b->dist[1], &b->cm, &b->tail, &b->uid, &b->flag, &b->pos,
		&b->u_head, &b->u_n, &b->u_len, &b->u_start, &b->u_end, &b->u_off, &b->mark };
	for (DevBuf *x : vbufs) CHK(dev_reserve(c, *x, vb));
	CHK(dev_reserve(c, b->ua, ((size_t)V + 16) * 8));
	CHK(dev_reserve(c, b->circ, (size_t)V + 16)); CHK(dev_reserve(c, b->ishead, (size_t)V + 16));
	const int ag = c->ag;
	ug_t a;
	a.au = P<uint32_t>(c->au[ag]); a.av = P<uint32_t>(c->av[ag]); a.alen = P<uint32_t>(c->alen[ag]); a.aol = P<uint32_t>(c->aol[ag]);
	a.idx = P<unsigned long long>(c->idx); a.sdel = P<uint8_t>(c->sdel); a.slen = P<uint32_t>(c->slen); a.n_vtx = V;
	a.nxt = P<uint32_t>(b->nxt); a.prv = P<uint32_t>(b->prv); a.wt = P<uint32_t>(b->wt); a.cm = P<uint32_t>(b->cm); a.tail = P<uint32_t>(b->tail); a.uid = P<uint32_t>(b->uid);
	a.flag = P<uint32_t>(b->flag); a.pos = P<uint32_t>(b->pos); a.circ = P<uint8_t>(b->circ); a.mark = P<int32_t>(b->mark);
	a.u_head = P<uint32_t>(b->u_head); a.u_n = P<uint32_t>(b->u_n); a.u_len = P<uint32_t>(b->u_len); a.u_start = P<uint32_t>(b->u_start); a.u_end = P<uint32_t>(b->u_end);
	a.u_off = P<uint32_t>(b->u_off); a.ua = P<unsigned long long>(b->ua);
	unsigned long long *ctr = P<unsigned long long>(c->ctr);
	const unsigned gv = grid_for(V, 256);
	ProfScope ps(c, "ug_gen", 0);
	hipLaunchKernelGGL(k_ug_link, dim3(gv), dim3(256), 0, c->st, a);
	int g = 0;
	HIPCHK(hipMemsetAsync(b->circ.p, 0, (size_t)V, c->st));
	CHK(ctr_zero(c));
	CHK(ug_rank(c, b, a, &g));                      // head, offset, discovery vertex and length of every chain in one ranking pass
	hipLaunchKernelGGL(k_ug_cycle_check, dim3(gv), dim3(256), 0, c->st, a, (const uint32_t*)P<uint32_t>(b->ptr[g]), ctr);
	CHK(ctr_fetch(c));
	uint32_t *d_tot = (uint32_t*)(ctr + CT_TOTAL);
	const bool irregular = c->h_ctr[CT_OVF2] != 0;
	if (irregular) { // not a symmetric graph: chains and twins do not exist; the reference's sweep, on one lane
		unsigned long long cap = b->ua.cap / 8;
		for (int attempt = 0;; ++attempt) {
			HIPCHK(hipMemsetAsync(b->circ.p, 0, (size_t)V, c->st));
			CHK(ctr_zero(c));
			hipLaunchKernelGGL(k_ug_seq, dim3(1), dim3(64), 0, c->st, a, P<uint8_t>(b->circ), cap, ctr);
			CHK(ctr_fetch(c));
			if (c->h_ctr[CT_CUT]) { mahip_set_error("mahip_ug_gen: a unitig walk never ends on this (asymmetric) graph; the reference does not return on it either"); return -1; }
			if (c->h_ctr[CT_TOTAL] <= cap) break;
			if (attempt || c->h_ctr[CT_TOTAL] > 0xffffffffull) { mahip_set_error("mahip_ug_gen: unitig members do not fit"); return -1; }
			cap = c->h_ctr[CT_TOTAL] + 16;
			CHK(dev_reserve(c, b->ua, cap * 8));
			a.ua = P<unsigned long long>(b->ua);
		}
		const uint32_t U = (uint32_t)c->h_ctr[CT_LIVE];
		b->n_utg = U; b->n_mem = (uint32_t)c->h_ctr[CT_TOTAL];
		if (n_utg) *n_utg = U;
		if (U == 0) return 0;
		HIPCHK(hipMemsetAsync(b->mark.p, 0xff, (size_t)V * 4, c->st));
		hipLaunchKernelGGL(k_ug_mark_seq, dim3(1), dim3(64), 0, c->st, a, U);
	} else {
	if (c->h_ctr[CT_OVF]) { // some chain has no head: cut every cycle in front of its smallest vertex, rank again
		hipLaunchKernelGGL(k_ug_heads, dim3(gv), dim3(256), 0, c->st, a, P<uint8_t>(b->ishead));
		hipLaunchKernelGGL(k_ug_cut, dim3(gv), dim3(256), 0, c->st, a, (const uint32_t*)P<uint32_t>(b->ptr[g]), (const uint32_t*)P<uint32_t>(b->mn[g]), (const uint8_t*)P<uint8_t>(b->ishead));
		CHK(ug_rank(c, b, a, &g));
	}
	const ug_rank_t rk = rank_bufs(b, g);
	const uint32_t *ptr = rk.ptr, *dist = rk.dist;
	HIPCHK(hipMemsetAsync(b->cm.p, 0xff, (size_t)V * 4, c->st));
	HIPCHK(hipMemsetAsync(b->tail.p, 0xff, (size_t)V * 4, c->st));
	hipLaunchKernelGGL(k_ug_chain, dim3(gv), dim3(256), 0, c->st, a, rk);
	HIPCHK(hipMemsetAsync(b->flag.p, 0, (size_t)V * 4, c->st));
	hipLaunchKernelGGL(k_ug_pick, dim3(gv), dim3(256), 0, c->st, a, ptr);
	CHK(scan_exclusive_u32(c, P<uint32_t>(b->flag), P<uint32_t>(b->pos), V, d_tot));
	CHK(ctr_fetch(c));
	const uint32_t U = (uint32_t)(c->h_ctr[CT_TOTAL] & 0xffffffffu);
	b->n_utg = U;
	if (n_utg) *n_utg = U;
	if (U == 0) return 0;
	hipLaunchKernelGGL(k_ug_units, dim3(gv), dim3(256), 0, c->st, a, rk);
	CHK(scan_exclusive_u32(c, P<uint32_t>(b->u_n), P<uint32_t>(b->u_off), U, d_tot));
	hipLaunchKernelGGL(k_ug_fill, dim3(gv), dim3(256), 0, c->st, a, ptr, dist);
	// arcs between unitig ends
	HIPCHK(hipMemsetAsync(b->mark.p, 0xff, (size_t)V * 4, c->st));
	hipLaunchKernelGGL(k_ug_mark, dim3(grid_for(U, 256)), dim3(256), 0, c->st, a, U);
	CHK(ctr_fetch(c));
	b->n_mem = (uint32_t)(c->h_ctr[CT_TOTAL] & 0xffffffffu);
	} // symmetric graph
	uint32_t n_ua = 0;
	if (A) {
		CHK(dev_reserve(c, b->akeep, (A + 16) * 4)); CHK(dev_reserve(c, b->apos, (A + 16) * 4));
		hipLaunchKernelGGL(k_ug_arc_keep, dim3(grid_for(A, 256)), dim3(256), 0, c->st, a, A, P<uint32_t>(b->akeep));
		CHK(scan_exclusive_u32(c, P<uint32_t>(b->akeep), P<uint32_t>(b->apos), A, d_tot));
		CHK(ctr_fetch(c));
		n_ua = (uint32_t)(c->h_ctr[CT_TOTAL] & 0xffffffffu);
		CHK(dev_reserve(c, b->arcs, ((size_t)n_ua + 1) * 16));
		if (n_ua) hipLaunchKernelGGL(k_ug_arc_emit, dim3(grid_for(A, 256)), dim3(256), 0, c->st, a, A, (const uint32_t*)P<uint32_t>(b->akeep), (const uint32_t*)P<uint32_t>(b->apos), (asg_arc_t*)b->arcs.p);
	}
	b->n_uarc = n_ua;
	HIPCHK(hipGetLastError());
	if (n_members) *n_members = b->n_mem;
	if (n_uarc) *n_uarc = n_ua;
	return 0;
}
// host copies: per unitig {n, len, start, end} (start == end == 0xffffffff: circular), the offsets, the members, the (unsorted) unitig arcs in push order
extern "C" int mahip_ug_download(mahip_ctx_t *c, uint32_t *u_n, uint32_t *u_len, uint32_t *u_start, uint32_t *u_end, uint32_t *u_off, uint64_t *members, asg_arc_t *uarcs)
{
	HIPCHK(hipSetDevice(c->dev));
	UgBufs *b = (UgBufs*)c->ug;
	if (!b) { mahip_set_error("mahip_ug_download: no unitigs"); return -1; }
	const size_t U = b->n_utg;
	if (U) {
		HIPCHK(hipMemcpyAsync(u_n, b->u_n.p, U * 4, hipMemcpyDeviceToHost, c->st));
		HIPCHK(hipMemcpyAsync(u_len, b->u_len.p, U * 4, hipMemcpyDeviceToHost, c->st));
		HIPCHK(hipMemcpyAsync(u_start, b->u_start.p, U * 4, hipMemcpyDeviceToHost, c->st));
		HIPCHK(hipMemcpyAsync(u_end, b->u_end.p, U * 4, hipMemcpyDeviceToHost, c->st));
		HIPCHK(hipMemcpyAsync(u_off, b->u_off.p, U * 4, hipMemcpyDeviceToHost, c->st));
	}
	if (b->n_mem) HIPCHK(hipMemcpyAsync(members, b->ua.p, (size_t)b->n_mem * 8, hipMemcpyDeviceToHost, c->st));
	if (b->n_uarc) HIPCHK(hipMemcpyAsync(uarcs, b->arcs.p, (size_t)b->n_uarc * 16, hipMemcpyDeviceToHost, c->st));
	HIPCHK(hipStreamSynchronize(c->st));
	return 0;
}
