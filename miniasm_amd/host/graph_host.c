/* graph_host.c -- the string-graph entry points of the reference's link interface (asg.h:31-42 and the non-header externals
 * of asg.c) over the device graph code.
 *
 * Every pass that computes something runs on the GPU: building / sorting / indexing, transitive reduction, the symmetry and
 * short-overlap filters (csrc/graph.hip) and the order-dependent cleaners -- tips, internal sequences, bi-loops, bubbles --
 * as a fixpoint over versioned state (csrc/clean.hip, csrc/clean_core.h).  The per-symbol entry points below keep the
 * reference's contract (a host asg_t that is valid after every call, libc-heap arrays, return values, log lines): they
 * upload the caller's graph, run the device pass and bring the result back.  The resident pipeline (pipeline.c) calls the
 * same device passes without the round trips.  What stays on the host is bookkeeping on small arrays: growing the arrays of
 * a graph under construction, the index of a sorted arc list, and the emulation of the reference's sort order for the few
 * arcs of the unitig graph.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ma_host.h"
#include "clean_core.h"

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

asg_arc_t *ma_asg_arc_pushp(asg_t *g) /* asg.h:45-52 */
{
	if (g->n_arc == g->m_arc) {
		g->m_arc = g->m_arc ? g->m_arc << 1 : 16;
		g->arc = (asg_arc_t*)realloc(g->arc, (size_t)g->m_arc * sizeof(asg_arc_t));
	}
	return &g->arc[g->n_arc++];
}

/* ---------------------------------------------------------------------------------------------- reference sort order
 * The reference sorts arcs with an in-place MSD radix sort (ksort.h:134-183) whose order of equal keys is a function of the
 * input order.  For host-side graphs (the unitig graph, the per-symbol asg_arc_sort) the same permutation refsort.c computes
 * for the device's exact-tie repair is applied to the records. */
void ma_refsort_arcs(asg_arc_t *beg, asg_arc_t *end)
{
	const size_t n = (size_t)(end - beg);
	uint64_t *keys;
	uint32_t *perm;
	asg_arc_t *tmp;
	size_t i;
	if (n < 2) return;
	keys = (uint64_t*)calloc(n, 8); perm = (uint32_t*)malloc(n * 4); tmp = (asg_arc_t*)malloc(n * sizeof(asg_arc_t));
	for (i = 0; i < n; ++i) keys[i] = beg[i].ul;
	ma_refsort_perm(keys, n, perm);
	for (i = 0; i < n; ++i) tmp[i] = beg[perm[i]];
	memcpy(beg, tmp, n * sizeof(asg_arc_t));
	free(keys); free(perm); free(tmp);
}

void asg_arc_sort(asg_t *g) { ma_refsort_arcs(g->arc, g->arc + g->n_arc); } /* asg.c:22-25 */

uint64_t *asg_arc_index_core(size_t max_seq, size_t n, const asg_arc_t *a) /* asg.c:27-36: one run of equal source vertices per entry */
{
	uint64_t *idx = (uint64_t*)calloc(max_seq ? max_seq * 2 : 1, 8);
	size_t beg = 0, i;
	for (i = 1; i <= n; ++i)
		if (i == n || a[i].ul >> 32 != a[beg].ul >> 32) {
			idx[a[beg].ul >> 32] = (uint64_t)beg << 32 | (i - beg);
			beg = i;
		}
	return idx;
}

void asg_arc_index(asg_t *g)
{
	free(g->idx);
	g->idx = asg_arc_index_core(g->n_seq, g->n_arc, g->arc);
}

void asg_arc_rm(asg_t *g) /* asg.c:57-70: order-preserving removal of flagged arcs and of arcs with a deleted endpoint */
{
	uint32_t e, n = 0;
	for (e = 0; e < g->n_arc; ++e) {
		const asg_arc_t *p = &g->arc[e];
		if (p->del || g->seq[p->ul >> 33].del || g->seq[p->v >> 1].del) continue;
		if (n != e) g->arc[n] = *p;
		++n;
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

/* ---------------------------------------------------------------------------------------------- device round trip */

static mahip_ctx_t *graph_up(asg_t *g)
{
	mahip_ctx_t *c = ma_gpu();
	if (!g->is_srt || g->idx == 0) asg_cleanup(g); /* the device works on a sorted, indexed graph */
	GPU(mahip_asg_upload(c, g));
	return c;
}

static void graph_down(mahip_ctx_t *c, asg_t *g)
{
	asg_t t;
	memset(&t, 0, sizeof(t));
	GPU(mahip_asg_download(c, &t));
	free(g->arc); free(g->idx); free(g->seq);
	g->arc = t.arc; g->idx = t.idx; g->seq = t.seq;
	g->n_arc = t.n_arc; g->m_arc = t.m_arc; g->m_seq = t.m_seq; g->n_seq = t.n_seq;
}

/* ---------------------------------------------------------------------------------------------- arc filters */

int asg_arc_del_multi(asg_t *g) /* asg.c:104-121 */
{
	mahip_ctx_t *c = graph_up(g);
	uint32_t n = 0;
	GPU(mahip_asg_del_multi(c, &n));
	if (n) graph_down(c, g);
	fprintf(MA_LOG, "[M::%s] removed %d multi-arcs\n", __func__, n);
	return n;
}

int asg_arc_del_asymm(asg_t *g) /* asg.c:124-138 */
{
	mahip_ctx_t *c = graph_up(g);
	uint32_t n = 0;
	GPU(mahip_asg_del_asymm(c, &n));
	if (n) graph_down(c, g);
	fprintf(MA_LOG, "[M::%s] removed %d asymmetric arcs\n", __func__, n);
	return n;
}

static void symm_on_device(mahip_ctx_t *c)
{
	uint32_t n_multi = 0, n_asymm = 0;
	GPU(mahip_asg_symm(c, &n_multi, &n_asymm));
	fprintf(MA_LOG, "[M::%s] removed %d multi-arcs\n", "asg_arc_del_multi", n_multi);
	fprintf(MA_LOG, "[M::%s] removed %d asymmetric arcs\n", "asg_arc_del_asymm", n_asymm);
}

void asg_symm(asg_t *g) /* asg.c:140-145 */
{
	mahip_ctx_t *c = graph_up(g);
	symm_on_device(c);
	graph_down(c, g);
	g->is_symm = 1;
}

int asg_arc_del_short(asg_t *g, float drop_ratio) /* asg.c:83-101 */
{
	mahip_ctx_t *c = graph_up(g);
	uint32_t n_short = 0;
	GPU(mahip_asg_del_short(c, drop_ratio, &n_short));
	if (n_short) {
		symm_on_device(c);
		graph_down(c, g);
		g->is_symm = 1;
	}
	fprintf(MA_LOG, "[M::%s] removed %d short overlaps\n", __func__, n_short);
	return n_short;
}

int asg_arc_del_trans(asg_t *g, int fuzz) /* asg.c:148-193 */
{
	mahip_ctx_t *c = graph_up(g);
	uint32_t n_reduced = 0;
	GPU(mahip_asg_del_trans(c, fuzz, &n_reduced));
	fprintf(MA_LOG, "[M::%s] transitively reduced %d arcs\n", __func__, n_reduced);
	if (n_reduced) {
		symm_on_device(c);
		graph_down(c, g);
		g->is_symm = 1;
	}
	return n_reduced;
}

/* ---------------------------------------------------------------------------------------------- cleaners (asg.c:238-433) */

int asg_cut_tip(asg_t *g, int max_ext)
{
	mahip_ctx_t *c = graph_up(g);
	uint32_t cnt = 0;
	GPU(mahip_asg_cut_tip(c, max_ext, &cnt));
	if (cnt) graph_down(c, g);
	fprintf(MA_LOG, "[M::%s] cut %d tips\n", __func__, cnt);
	return cnt;
}

int asg_cut_internal(asg_t *g, int max_ext)
{
	mahip_ctx_t *c = graph_up(g);
	uint32_t cnt = 0;
	GPU(mahip_asg_cut_internal(c, max_ext, &cnt));
	if (cnt) graph_down(c, g);
	fprintf(MA_LOG, "[M::%s] cut %d internal sequences\n", __func__, cnt);
	return cnt;
}

int asg_cut_biloop(asg_t *g, int max_ext)
{
	mahip_ctx_t *c = graph_up(g);
	uint32_t cnt = 0;
	GPU(mahip_asg_cut_biloop(c, max_ext, &cnt));
	if (cnt) graph_down(c, g);
	fprintf(MA_LOG, "[M::%s] cut %d small bi-loops\n", __func__, cnt);
	return cnt;
}

int asg_pop_bubble(asg_t *g, int max_dist)
{
	mahip_ctx_t *c;
	uint32_t n_pop = 0, n_tips = 0;
	if (!g->is_symm) asg_symm(g); /* asg.c:418 */
	c = graph_up(g);
	GPU(mahip_asg_pop_bubble(c, max_dist, &n_pop, &n_tips));
	if (n_pop) graph_down(c, g);
	fprintf(MA_LOG, "[M::%s] popped %d bubbles and trimmed %d tips\n", __func__, n_pop, n_tips);
	return (int)n_pop;
}

/* asg.c:225-236, an external other objects may bind: the walk of one unitig end, read-only.  Runs the same per-vertex function
 * the device cleaners use (clean_core.h) on the caller's graph with an empty stamp set. */
int asg_extend(const asg_t *g, uint32_t v, int max_ext, asg64_v *a)
{
	const uint32_t R = g->n_seq;
	uint32_t *av = (uint32_t*)malloc(((size_t)g->n_arc + 1) * 4), *alen = (uint32_t*)malloc(((size_t)g->n_arc + 1) * 4), *aol = (uint32_t*)malloc(((size_t)g->n_arc + 1) * 4);
	uint32_t *rst = (uint32_t*)malloc(((size_t)R + 1) * 4), *ast = (uint32_t*)malloc(((size_t)g->n_arc + 1) * 4), i, e = 0;
	uint8_t *sdel = (uint8_t*)malloc((size_t)R + 1);
	cl_view_t w;
	int kind;
	for (i = 0; i < g->n_arc; ++i) av[i] = g->arc[i].v, alen[i] = (uint32_t)g->arc[i].ul, aol[i] = g->arc[i].ol | (uint32_t)g->arc[i].del << 31, ast[i] = CL_NONE;
	for (i = 0; i < R; ++i) sdel[i] = g->seq[i].del, rst[i] = CL_NONE;
	w.av = av; w.alen = alen; w.aol = aol; w.idx = (const unsigned long long*)g->idx; w.sdel = sdel; w.rst = rst; w.ast = ast; w.n_vtx = 2 * R; w.no_stamps = 1;
#define EXT_PUSH(x) do { if (a->n == a->m) { a->m = a->m ? a->m << 1 : 2; a->a = (uint64_t*)realloc(a->a, a->m * 8); } a->a[a->n] = (x); ++a->n; } while (0)
	a->n = 0;
	EXT_PUSH((uint64_t)v);
	do {
		kind = cl_end_kind(&w, v ^ 1, 0, &e);
		if (kind != CL_MERGEABLE) break;
		v = av[e];
		EXT_PUSH((uint64_t)alen[e] << 32 | v);
	} while (--max_ext > 0);
#undef EXT_PUSH
	free(av); free(alen); free(aol); free(rst); free(ast); free(sdel);
	return kind;
}
