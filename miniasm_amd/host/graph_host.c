/* asg.c -- host side of the assembly string graph (reference asg.h:31-42, asg.c:11-433).
 *
 * Division of labour: building the graph, sorting/indexing it, Myers' transitive reduction and the
 * symmetry passes run on the GPU over the full overlap graph (miniasm_amd/csrc/graph.hip).  What is left
 * afterwards is the small reduced graph; the cleaners below (tip cutting, bubble popping, short-overlap,
 * internal/bi-loop cuts) are inherently sequential sweeps that mutate the graph as they go (a later
 * vertex sees the deletions of an earlier one), so they run here, on the host, over that small graph.
 * Every function reproduces the reference's result exactly, including visiting order and log lines.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include "ma_host.h"

#define GPU(call) do { if ((call) != 0) ma_gpu_fail(__func__); } while (0)

/* ---------------------------------------------------------------------------------------------- basics */

asg_t *asg_init(void) { return (asg_t*)calloc(1, sizeof(asg_t)); }

void asg_destroy(asg_t *g)
{
	if (g == 0) return;
	free(g->arc); free(g->seq); free(g->idx);
	free(g);
}

static inline uint32_t roundup32(uint32_t x)
{
	--x; x |= x >> 1; x |= x >> 2; x |= x >> 4; x |= x >> 8; x |= x >> 16;
	return x + 1;
}

void asg_seq_set(asg_t *g, int sid, int len, int del) /* asg.c:44-54 */
{
	if ((uint32_t)sid >= g->m_seq) {
		g->m_seq = roundup32((uint32_t)sid + 1);
		g->seq = (asg_seq_t*)realloc(g->seq, (size_t)g->m_seq * sizeof(asg_seq_t));
	}
	if ((uint32_t)sid >= g->n_seq) g->n_seq = sid + 1;
	g->seq[sid].len = len;
	g->seq[sid].del = !!del;
}

static inline asg_arc_t *arc_push(asg_t *g) /* asg.h:45-52 */
{
	if (g->n_arc == g->m_arc) {
		g->m_arc = g->m_arc ? g->m_arc << 1 : 16;
		g->arc = (asg_arc_t*)realloc(g->arc, (size_t)g->m_arc * sizeof(asg_arc_t));
	}
	return &g->arc[g->n_arc++];
}
asg_arc_t *ma_asg_arc_pushp(asg_t *g) { return arc_push(g); }

/* flag every arc v->w (asg.h:55-61) */
static inline void arc_flag(asg_t *g, uint32_t v, uint32_t w, int del)
{
	uint32_t i, nv = asg_arc_n(g, v);
	asg_arc_t *av = asg_arc_a(g, v);
	for (i = 0; i < nv; ++i)
		if (av[i].v == w) av[i].del = !!del;
}

/* delete read s with every arc touching it, in both directions (asg.h:64-77) */
static inline void seq_drop(asg_t *g, uint32_t s)
{
	uint32_t k;
	g->seq[s].del = 1;
	for (k = 0; k < 2; ++k) {
		uint32_t i, v = s << 1 | k, nv = asg_arc_n(g, v);
		asg_arc_t *av = asg_arc_a(g, v);
		for (i = 0; i < nv; ++i) {
			av[i].del = 1;
			arc_flag(g, av[i].v ^ 1, v ^ 1, 1);
		}
	}
}

/* ---------------------------------------------------------------------------------------------- reference sort order
 * The reference sorts arcs with an in-place MSD radix sort (8-bit digits from the top byte, cycle-leader
 * permutation, insertion sort for runs of <= 64, ksort.h:134-183).  It is not stable: the order of arcs with
 * equal keys is a deterministic function of the input order, and that order is observable in the output.
 * This is an independent implementation of the same procedure (index based), used for the small host-side
 * graphs (unitig graph, per-symbol asg_arc_sort). */
#define RS_SMALL 64

static void rs_insertion(asg_arc_t *a, size_t n)
{
	size_t i, j;
	for (i = 1; i < n; ++i) {
		if (a[i].ul < a[i-1].ul) {
			asg_arc_t t = a[i];
			for (j = i; j > 0 && t.ul < a[j-1].ul; --j) a[j] = a[j-1];
			a[j] = t;
		}
	}
}

static void rs_level(asg_arc_t *a, size_t n, int shift)
{
	size_t head[256], tail[256], start[257], i;
	int k;
	memset(tail, 0, sizeof(tail));
	for (i = 0; i < n; ++i) ++tail[a[i].ul >> shift & 0xff];
	start[0] = 0;
	for (k = 0; k < 256; ++k) start[k + 1] = start[k] + tail[k], head[k] = start[k], tail[k] = start[k + 1];
	for (k = 0; k < 256;) { /* walk bucket k's unfinished part; follow displacement cycles until an element of k turns up */
		if (head[k] == tail[k]) { ++k; continue; }
		int dst = (int)(a[head[k]].ul >> shift & 0xff);
		if (dst == k) { ++head[k]; continue; }
		asg_arc_t carry = a[head[k]];
		do {
			asg_arc_t evicted = a[head[dst]];
			a[head[dst]++] = carry;
			carry = evicted;
			dst = (int)(carry.ul >> shift & 0xff);
		} while (dst != k);
		a[head[k]++] = carry;
	}
	if (shift) {
		int next = shift > 8 ? shift - 8 : 0;
		for (k = 0; k < 256; ++k) {
			size_t m = start[k + 1] - start[k];
			if (m > RS_SMALL) rs_level(a + start[k], m, next);
			else if (m > 1) rs_insertion(a + start[k], m);
		}
	}
}

void ma_refsort_arcs(asg_arc_t *beg, asg_arc_t *end)
{
	size_t n = (size_t)(end - beg);
	if (n <= RS_SMALL) rs_insertion(beg, n);
	else rs_level(beg, n, 56);
}

void asg_arc_sort(asg_t *g) { ma_refsort_arcs(g->arc, g->arc + g->n_arc); } /* asg.c:22-25 */

uint64_t *asg_arc_index_core(size_t max_seq, size_t n, const asg_arc_t *a) /* asg.c:27-36 */
{
	uint64_t *idx = (uint64_t*)calloc(max_seq * 2 ? max_seq * 2 : 1, 8);
	size_t i, first = 0;
	for (i = 1; i <= n; ++i)
		if (i == n || a[i].ul >> 32 != a[i-1].ul >> 32) {
			idx[a[i-1].ul >> 32] = (uint64_t)first << 32 | (i - first);
			first = i;
		}
	return idx;
}

void asg_arc_index(asg_t *g)
{
	free(g->idx);
	g->idx = asg_arc_index_core(g->n_seq, g->n_arc, g->arc);
}

void asg_arc_rm(asg_t *g) /* asg.c:57-70 */
{
	uint32_t e, n = 0;
	for (e = 0; e < g->n_arc; ++e) {
		const asg_arc_t *p = &g->arc[e];
		if (!p->del && !g->seq[p->ul >> 33].del && !g->seq[p->v >> 1].del) g->arc[n++] = *p;
	}
	if (n < g->n_arc) { free(g->idx); g->idx = 0; }
	g->n_arc = n;
}

void asg_cleanup(asg_t *g) /* asg.c:72-80 */
{
	asg_arc_rm(g);
	if (!g->is_srt) { asg_arc_sort(g); g->is_srt = 1; }
	if (g->idx == 0) asg_arc_index(g);
}

/* ---------------------------------------------------------------------------------------------- arc filters (host versions, small graphs) */

int asg_arc_del_multi(asg_t *g) /* asg.c:104-121: per vertex keep the first arc to each target */
{
	uint32_t v, n_vtx = g->n_seq * 2, n_multi = 0;
	uint32_t *seen = (uint32_t*)calloc(n_vtx ? n_vtx : 1, 4);
	for (v = 0; v < n_vtx; ++v) {
		asg_arc_t *av = asg_arc_a(g, v);
		int32_t i, nv = asg_arc_n(g, v);
		if (nv < 2) continue;
		for (i = 0; i < nv; ++i) { /* first occurrence of a target survives, later ones go */
			if (seen[av[i].v] == v + 1) av[i].del = 1, ++n_multi;
			else seen[av[i].v] = v + 1;
		}
	}
	free(seen);
	if (n_multi) asg_cleanup(g);
	fprintf(MA_LOG, "[M::%s] removed %d multi-arcs\n", __func__, n_multi);
	return n_multi;
}

int asg_arc_del_asymm(asg_t *g) /* asg.c:124-138 */
{
	uint32_t e, n_asymm = 0;
	for (e = 0; e < g->n_arc; ++e) {
		uint32_t v = g->arc[e].v ^ 1, u = (uint32_t)(g->arc[e].ul >> 32) ^ 1;
		uint32_t i, nv = asg_arc_n(g, v);
		const asg_arc_t *av = asg_arc_a(g, v);
		for (i = 0; i < nv; ++i)
			if (av[i].v == u) break;
		if (i == nv) g->arc[e].del = 1, ++n_asymm;
	}
	if (n_asymm) asg_cleanup(g);
	fprintf(MA_LOG, "[M::%s] removed %d asymmetric arcs\n", __func__, n_asymm);
	return n_asymm;
}

void asg_symm(asg_t *g) /* asg.c:140-145 */
{
	asg_arc_del_multi(g);
	asg_arc_del_asymm(g);
	g->is_symm = 1;
}

int asg_arc_del_short(asg_t *g, float drop_ratio) /* asg.c:83-101 */
{
	uint32_t v, n_vtx = g->n_seq * 2, n_short = 0;
	for (v = 0; v < n_vtx; ++v) {
		asg_arc_t *av = asg_arc_a(g, v);
		uint32_t i, thres, nv = asg_arc_n(g, v);
		if (nv < 2) continue;
		thres = (uint32_t)(av[0].ol * drop_ratio + .499);
		for (i = nv - 1; i >= 1 && av[i].ol < thres; --i);
		for (i = i + 1; i < nv; ++i) av[i].del = 1, ++n_short;
	}
	if (n_short) {
		asg_cleanup(g);
		asg_symm(g);
	}
	fprintf(MA_LOG, "[M::%s] removed %d short overlaps\n", __func__, n_short);
	return n_short;
}

/* ---------------------------------------------------------------------------------------------- transitive reduction: GPU */

int asg_arc_del_trans(asg_t *g, int fuzz) /* asg.c:148-193 */
{
	mahip_ctx_t *c = ma_gpu();
	uint32_t n_reduced = 0;
	GPU(mahip_asg_upload(c, g));
	GPU(mahip_asg_del_trans(c, fuzz, &n_reduced));
	fprintf(MA_LOG, "[M::%s] transitively reduced %d arcs\n", __func__, n_reduced);
	if (n_reduced) {
		uint32_t n_multi = 0, n_asymm = 0;
		GPU(mahip_asg_symm(c, &n_multi, &n_asymm));
		fprintf(MA_LOG, "[M::%s] removed %d multi-arcs\n", "asg_arc_del_multi", n_multi);
		fprintf(MA_LOG, "[M::%s] removed %d asymmetric arcs\n", "asg_arc_del_asymm", n_asymm);
		{
			asg_t t;
			memset(&t, 0, sizeof(t));
			GPU(mahip_asg_download(c, &t));
			free(g->arc); free(g->idx); free(g->seq);
			g->arc = t.arc; g->idx = t.idx; g->seq = t.seq;
			g->n_arc = t.n_arc; g->m_arc = t.m_arc; g->m_seq = t.m_seq;
			g->is_symm = 1;
		}
	}
	return n_reduced;
}

/* ---------------------------------------------------------------------------------------------- short-unitig pruning (asg.c:199-306) */

enum { UE_MERGEABLE = 0, UE_TIP = 1, UE_MULTI_OUT = 2, UE_MULTI_NEI = 3 };

/* what lies beyond the far end of vertex v (i.e. out of v^1): nothing, a fork, a unique neighbour that
 * itself forks back, or a unique mergeable neighbour (then *lw = arc length<<32 | neighbour) */
static inline int utg_end_kind(const asg_t *g, uint32_t v, uint64_t *lw)
{
	uint32_t w, n_live = 0, n_back = 0, i, nv = asg_arc_n(g, v ^ 1), nw;
	const asg_arc_t *av = asg_arc_a(g, v ^ 1), *aw;
	int last = -1;
	for (i = 0; i < nv; ++i)
		if (!av[i].del) last = (int)i, ++n_live;
	if (n_live == 0) return UE_TIP;
	if (n_live > 1) return UE_MULTI_OUT;
	if (lw) *lw = av[last].ul << 32 | av[last].v;
	w = av[last].v ^ 1;
	nw = asg_arc_n(g, w); aw = asg_arc_a(g, w);
	for (i = 0; i < nw; ++i)
		if (!aw[i].del) ++n_back;
	return n_back != 1 ? UE_MULTI_NEI : UE_MERGEABLE;
}

static inline void v64_push(asg64_v *a, uint64_t x)
{
	if (a->n == a->m) {
		a->m = a->m ? a->m << 1 : 2;
		a->a = (uint64_t*)realloc(a->a, a->m * 8);
	}
	a->a[a->n++] = x;
}

int asg_extend(const asg_t *g, uint32_t v, int max_ext, asg64_v *a) /* asg.c:217-236 */
{
	int kind;
	uint64_t lw;
	a->n = 0;
	v64_push(a, v);
	do {
		kind = utg_end_kind(g, v ^ 1, &lw);
		if (kind != UE_MERGEABLE) break;
		v64_push(a, lw);
		v = (uint32_t)lw;
	} while (--max_ext > 0);
	return kind;
}

int asg_cut_tip(asg_t *g, int max_ext) /* asg.c:238-254 */
{
	asg64_v a = {0, 0, 0};
	uint32_t v, n_vtx = g->n_seq * 2, cnt = 0;
	size_t i;
	for (v = 0; v < n_vtx; ++v) {
		if (g->seq[v >> 1].del) continue;
		if (utg_end_kind(g, v, 0) != UE_TIP) continue;
		if (asg_extend(g, v, max_ext, &a) == UE_MERGEABLE) continue; /* the unitig is longer than max_ext reads */
		for (i = 0; i < a.n; ++i) seq_drop(g, (uint32_t)a.a[i] >> 1);
		++cnt;
	}
	free(a.a);
	if (cnt > 0) asg_cleanup(g);
	fprintf(MA_LOG, "[M::%s] cut %d tips\n", __func__, cnt);
	return cnt;
}

int asg_cut_internal(asg_t *g, int max_ext) /* asg.c:256-272 */
{
	asg64_v a = {0, 0, 0};
	uint32_t v, n_vtx = g->n_seq * 2, cnt = 0;
	size_t i;
	for (v = 0; v < n_vtx; ++v) {
		if (g->seq[v >> 1].del) continue;
		if (utg_end_kind(g, v, 0) != UE_MULTI_NEI) continue;
		if (asg_extend(g, v, max_ext, &a) != UE_MULTI_NEI) continue;
		for (i = 0; i < a.n; ++i) seq_drop(g, (uint32_t)a.a[i] >> 1);
		++cnt;
	}
	free(a.a);
	if (cnt > 0) asg_cleanup(g);
	fprintf(MA_LOG, "[M::%s] cut %d internal sequences\n", __func__, cnt);
	return cnt;
}

int asg_cut_biloop(asg_t *g, int max_ext) /* asg.c:274-306 */
{
	asg64_v a = {0, 0, 0};
	uint32_t v, n_vtx = g->n_seq * 2, cnt = 0;
	for (v = 0; v < n_vtx; ++v) {
		uint32_t i, nv, nw, w = UINT32_MAX, x, ov = 0, ox = 0;
		const asg_arc_t *av, *aw;
		if (g->seq[v >> 1].del) continue;
		if (utg_end_kind(g, v, 0) != UE_MULTI_NEI) continue;
		if (asg_extend(g, v, max_ext, &a) != UE_MULTI_OUT) continue;
		x = (uint32_t)a.a[a.n - 1] ^ 1;
		nv = asg_arc_n(g, v ^ 1); av = asg_arc_a(g, v ^ 1);
		for (i = 0; i < nv; ++i)
			if (!av[i].del) w = av[i].v ^ 1;
		assert(w != UINT32_MAX);
		nw = asg_arc_n(g, w); aw = asg_arc_a(g, w);
		for (i = 0; i < nw; ++i) { /* pattern: v->...->x', w->v and w->x */
			if (aw[i].del) continue;
			if (aw[i].v == x) ox = aw[i].ol;
			if (aw[i].v == v) ov = aw[i].ol;
		}
		if (ov == 0 && ox == 0) continue;
		if (ov > ox) {
			arc_flag(g, w, x, 1);
			arc_flag(g, x ^ 1, w ^ 1, 1);
			++cnt;
		}
	}
	free(a.a);
	if (cnt > 0) asg_cleanup(g);
	fprintf(MA_LOG, "[M::%s] cut %d small bi-loops\n", __func__, cnt);
	return cnt;
}

/* ---------------------------------------------------------------------------------------------- bubble popping (asg.c:312-433) */

typedef struct {
	uint32_t parent;     /* best predecessor */
	uint32_t dist;       /* shortest distance from the source */
	uint32_t cnt;        /* most reads on a path from the source */
	uint32_t pending:31, seen:1; /* in-arcs not yet visited; visited flag */
} bub_info_t;

typedef struct { size_t n, m; uint32_t *a; } u32_v;

static inline void u32_push(u32_v *v, uint32_t x)
{
	if (v->n == v->m) {
		v->m = v->m ? v->m << 1 : 2;
		v->a = (uint32_t*)realloc(v->a, v->m * 4);
	}
	v->a[v->n++] = x;
}

typedef struct {
	bub_info_t *info;
	u32_v ready;   /* vertices whose in-arcs have all been visited */
	u32_v tips;    /* visited dead ends */
	u32_v touched; /* visited vertices */
	u32_v arcs;    /* visited arcs */
} bub_buf_t;

static inline uint32_t live_out(const asg_t *g, uint32_t v)
{
	uint32_t i, n = 0, nv = asg_arc_n(g, v);
	const asg_arc_t *av = asg_arc_a(g, v);
	for (i = 0; i < nv; ++i)
		if (!av[i].del) ++n;
	return n;
}

/* the bubble from v0 closed at b->ready.a[0]: drop everything visited, then resurrect the best path (asg.c:338-357) */
static void bub_backtrack(asg_t *g, uint32_t v0, bub_buf_t *b)
{
	size_t i;
	uint32_t v;
	assert(b->ready.n == 1);
	for (i = 0; i < b->touched.n; ++i) g->seq[b->touched.a[i] >> 1].del = 1;
	for (i = 0; i < b->arcs.n; ++i) {
		asg_arc_t *p = &g->arc[b->arcs.a[i]];
		p->del = 1;
		arc_flag(g, p->v ^ 1, (uint32_t)(p->ul >> 32) ^ 1, 1);
	}
	v = b->ready.a[0];
	do {
		uint32_t u = b->info[v].parent;
		g->seq[v >> 1].del = 0;
		arc_flag(g, u, v, 0);
		arc_flag(g, v ^ 1, u ^ 1, 0);
		v = u;
	} while (v != v0);
}

/* try to pop one bubble rooted at v0 (asg.c:360-409); returns 1 | n_tips<<32 when popped */
static uint64_t bub_pop1(asg_t *g, uint32_t v0, int max_dist, bub_buf_t *b)
{
	uint32_t i, n_pending = 0;
	uint64_t ret = 0;
	size_t k;
	if (g->seq[v0 >> 1].del) return 0;
	if ((uint32_t)g->idx[v0] < 2) return 0;
	b->ready.n = b->tips.n = b->touched.n = b->arcs.n = 0;
	b->info[v0].cnt = b->info[v0].dist = 0;
	u32_push(&b->ready, v0);
	do {
		uint32_t v = b->ready.a[--b->ready.n], d = b->info[v].dist, c = b->info[v].cnt;
		uint32_t nv = asg_arc_n(g, v);
		const asg_arc_t *av = asg_arc_a(g, v);
		assert(nv > 0);
		for (i = 0; i < nv; ++i) {
			uint32_t w = av[i].v, l = (uint32_t)av[i].ul;
			bub_info_t *t = &b->info[w];
			if (w == v0) goto reset; /* a cycle through the source */
			if (av[i].del) continue;
			u32_push(&b->arcs, (uint32_t)(g->idx[v] >> 32) + i);
			if (d + l > (uint32_t)max_dist) break; /* too far */
			if (!t->seen) {
				u32_push(&b->touched, w);
				t->parent = v, t->seen = 1, t->dist = d + l;
				t->pending = live_out(g, w ^ 1);
				++n_pending;
			} else {
				if (c + 1 > t->cnt || (c + 1 == t->cnt && d + l > t->dist)) t->parent = v;
				if (c + 1 > t->cnt) t->cnt = c + 1;
				if (d + l < t->dist) t->dist = d + l;
			}
			assert(t->pending > 0);
			if (--t->pending == 0) {
				if (asg_arc_n(g, w)) u32_push(&b->ready, w); /* counts deleted arcs too, like the reference (asg.c:393) */
				else u32_push(&b->tips, w);
				--n_pending;
			}
		}
		if (i < nv || b->ready.n == 0) goto reset;
	} while (b->ready.n > 1 || n_pending);
	bub_backtrack(g, v0, b);
	ret = 1 | (uint64_t)b->tips.n << 32;
reset:
	for (k = 0; k < b->touched.n; ++k) {
		bub_info_t *t = &b->info[b->touched.a[k]];
		t->seen = 0, t->cnt = 0, t->dist = 0;
	}
	return ret;
}

int asg_pop_bubble(asg_t *g, int max_dist) /* asg.c:412-433 */
{
	uint32_t v, n_vtx = g->n_seq * 2;
	uint64_t n_pop = 0;
	bub_buf_t b;
	if (!g->is_symm) asg_symm(g);
	memset(&b, 0, sizeof(b));
	b.info = (bub_info_t*)calloc(n_vtx ? n_vtx : 1, sizeof(bub_info_t));
	for (v = 0; v < n_vtx; ++v) {
		uint32_t i, n_live = 0, nv = asg_arc_n(g, v);
		const asg_arc_t *av = asg_arc_a(g, v);
		if (nv < 2 || g->seq[v >> 1].del) continue;
		for (i = 0; i < nv; ++i)
			if (!av[i].del) ++n_live;
		if (n_live > 1) n_pop += bub_pop1(g, v, max_dist, &b);
	}
	free(b.info); free(b.ready.a); free(b.tips.a); free(b.touched.a); free(b.arcs.a);
	if (n_pop) asg_cleanup(g);
	fprintf(MA_LOG, "[M::%s] popped %d bubbles and trimmed %d tips\n", __func__, (uint32_t)n_pop, (uint32_t)(n_pop >> 32));
	return (int)n_pop;
}
