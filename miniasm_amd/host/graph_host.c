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
#include <pthread.h>
#include <unistd.h>
#include "ma_host.h"

#define GPU(call) do { if ((call) != 0) ma_gpu_fail(__func__); } while (0)

/* ---------------------------------------------------------------------------------------------- worker threads
 * The cleaners are order-dependent sweeps, but almost every vertex they look at is left alone.  On big graphs the
 * expensive read-only part (classifying a vertex end, probing for a bubble) runs SPECULATIVELY on worker threads over a
 * snapshot of the graph, recording which reads' state each probe looked at; the sweep itself then runs in the reference's
 * order and trusts a speculative "nothing to do here" only while none of those reads has been modified since the
 * snapshot (a dirty map fed by every mutation) -- anything else is re-evaluated on the spot by the sequential code.
 * The result is the sequential one by construction (ordered speculative commit). */
typedef void (*par_body_t)(void *arg, int tid, size_t lo, size_t hi);
typedef struct { par_body_t fn; void *arg; size_t n, chunk, *next; int tid; } par_job_t;

static void *par_thread(void *p)
{
	par_job_t *j = (par_job_t*)p;
	for (;;) {
		size_t lo = __atomic_fetch_add(j->next, j->chunk, __ATOMIC_RELAXED);
		if (lo >= j->n) break;
		j->fn(j->arg, j->tid, lo, lo + j->chunk < j->n ? lo + j->chunk : j->n);
	}
	return 0;
}

#define PAR_MAX_THREADS 64
static void par_run(size_t n, size_t chunk, int nth, par_body_t fn, void *arg)
{
	par_job_t job[PAR_MAX_THREADS];
	pthread_t th[PAR_MAX_THREADS];
	int started[PAR_MAX_THREADS], t;
	size_t next = 0;
	if (nth > PAR_MAX_THREADS) nth = PAR_MAX_THREADS;
	if (nth < 1) nth = 1;
	for (t = 0; t < nth; ++t) { job[t].fn = fn; job[t].arg = arg; job[t].n = n; job[t].chunk = chunk ? chunk : 1; job[t].next = &next; job[t].tid = t; }
	for (t = 1; t < nth; ++t) started[t] = pthread_create(&th[t], 0, par_thread, &job[t]) == 0;
	par_thread(&job[0]);
	for (t = 1; t < nth; ++t) if (started[t]) pthread_join(th[t], 0);
}

int ma_clean_threads(void)
{
	const char *s = getenv("MA_THREADS");
	long n = s ? atol(s) : sysconf(_SC_NPROCESSORS_ONLN);
	if (!s && n > 8) n = 8; /* more threads only add start-up and barrier time at these sizes (8: 144 ms, 16: 152 ms, 32: 178 ms per 220 k reads) */
	if (n < 1) n = 1;
	if (n > PAR_MAX_THREADS) n = PAR_MAX_THREADS;
	return (int)n;
}

/* graphs below this many vertices are swept by the plain sequential code (MA_CLEAN_PAR_MIN overrides: tests use 0) */
static int clean_par_at(const asg_t *g, long dflt_min_vtx)
{
	const char *s = getenv("MA_CLEAN_PAR_MIN");
	long min_vtx = s ? atol(s) : dflt_min_vtx;
	return ma_clean_threads() > 1 && (long)g->n_seq * 2 >= min_vtx;
}
/* measured on the MI355X host (2 x EPYC 9575F), 440 k vertices: bubble probes, short-overlap / symmetry filters and the
 * arc compaction gain 30-40 % from 8 threads on a noisy graph, the end classification of the tip / internal / bi-loop sweeps
 * does not; on a clean graph of that size (few sources, few deletions) the threads only cost their start-up: the default
 * thresholds keep the speculative sweeps for cfg5-class graphs */
static int clean_par(const asg_t *g) { return clean_par_at(g, 1000000); }
static int clean_par_ends(const asg_t *g) { return clean_par_at(g, 4000000); }

typedef struct { uint8_t *map; size_t n; } dirty_t; /* per read: state (seq.del or an arc's del on either strand) changed since the snapshot */
static inline void dirty_mark(dirty_t *d, uint32_t read) { if (d && !d->map[read]) d->map[read] = 1, ++d->n; }

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

/* delete read s with every arc touching it, in both directions (asg.h:64-77); dm (optional) learns what changed */
static inline void seq_drop_mark(asg_t *g, uint32_t s, dirty_t *dm)
{
	uint32_t k;
	g->seq[s].del = 1;
	dirty_mark(dm, s);
	for (k = 0; k < 2; ++k) {
		uint32_t i, v = s << 1 | k, nv = asg_arc_n(g, v);
		asg_arc_t *av = asg_arc_a(g, v);
		for (i = 0; i < nv; ++i) {
			av[i].del = 1;
			arc_flag(g, av[i].v ^ 1, v ^ 1, 1);
			dirty_mark(dm, av[i].v >> 1);
		}
	}
}
static inline void seq_drop(asg_t *g, uint32_t s) { seq_drop_mark(g, s, 0); }

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
	uint64_t *idx = (uint64_t*)calloc(max_seq ? max_seq * 2 : 1, 8);
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

typedef struct { const asg_t *g; uint8_t *keep; } rm_par_t;
static void rm_body(void *arg, int tid, size_t lo, size_t hi)
{ /* the two endpoint look-ups are random accesses into seq[]: that is what the threads are for */
	rm_par_t *p = (rm_par_t*)arg;
	size_t e;
	(void)tid;
	for (e = lo; e < hi; ++e) {
		const asg_arc_t *a = &p->g->arc[e];
		p->keep[e] = !a->del && !p->g->seq[a->ul >> 33].del && !p->g->seq[a->v >> 1].del;
	}
}

void asg_arc_rm(asg_t *g) /* asg.c:57-70 */
{
	uint32_t e, n = 0;
	if (clean_par(g) && (g->n_arc >= 100000 || getenv("MA_CLEAN_PAR_MIN"))) {
		rm_par_t p;
		p.g = g; p.keep = (uint8_t*)malloc(g->n_arc);
		par_run(g->n_arc, 16384, ma_clean_threads(), rm_body, &p);
		for (e = 0; e < g->n_arc; ++e) /* order-preserving compaction: a streaming pass */
			if (p.keep[e]) { if (n != e) g->arc[n] = g->arc[e]; ++n; }
		free(p.keep);
	} else {
		for (e = 0; e < g->n_arc; ++e) {
			const asg_arc_t *p = &g->arc[e];
			if (!p->del && !g->seq[p->ul >> 33].del && !g->seq[p->v >> 1].del) g->arc[n++] = *p;
		}
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

/* The three filters below look at one vertex (or one arc) at a time and write only that vertex's own arcs: on big graphs
 * the vertex range is split over the worker threads; counts are summed.  Same flags as the sequential loops. */
typedef struct { asg_t *g; float ratio; uint32_t cnt[PAR_MAX_THREADS]; } flt_par_t;

static uint32_t multi_range(asg_t *g, uint32_t lo, uint32_t hi)
{ /* asg.c:104-121: per vertex the first arc to a target survives, later ones go (deleted arcs count as occurrences too) */
	uint32_t v, n_multi = 0, *seen = 0;
	for (v = lo; v < hi; ++v) {
		asg_arc_t *av = asg_arc_a(g, v);
		int32_t i, j, nv = asg_arc_n(g, v);
		if (nv < 2) continue;
		if (nv <= 32) {
			for (i = 1; i < nv; ++i) {
				for (j = 0; j < i; ++j)
					if (av[j].v == av[i].v) break;
				if (j < i) av[i].del = 1, ++n_multi;
			}
		} else { /* long list: stamp array over all vertices, allocated on first need */
			if (seen == 0) seen = (uint32_t*)calloc((size_t)g->n_seq * 2 + 1, 4);
			for (i = 0; i < nv; ++i) {
				if (seen[av[i].v] == v + 1) av[i].del = 1, ++n_multi;
				else seen[av[i].v] = v + 1;
			}
		}
	}
	free(seen);
	return n_multi;
}
static void multi_body(void *arg, int tid, size_t lo, size_t hi) { flt_par_t *p = (flt_par_t*)arg; p->cnt[tid] += multi_range(p->g, (uint32_t)lo, (uint32_t)hi); }

int asg_arc_del_multi(asg_t *g)
{
	uint32_t n_vtx = g->n_seq * 2, n_multi = 0;
	if (clean_par(g)) {
		flt_par_t p;
		int t;
		memset(&p, 0, sizeof(p)); p.g = g;
		par_run(n_vtx, 8192, ma_clean_threads(), multi_body, &p);
		for (t = 0; t < PAR_MAX_THREADS; ++t) n_multi += p.cnt[t];
	} else n_multi = multi_range(g, 0, n_vtx);
	if (n_multi) asg_cleanup(g);
	fprintf(MA_LOG, "[M::%s] removed %d multi-arcs\n", __func__, n_multi);
	return n_multi;
}

static uint32_t asymm_range(asg_t *g, uint32_t lo, uint32_t hi)
{ /* asg.c:124-138: u->v without v'->u' goes (the search ignores del flags) */
	uint32_t e, n_asymm = 0;
	for (e = lo; e < hi; ++e) {
		uint32_t v = g->arc[e].v ^ 1, u = (uint32_t)(g->arc[e].ul >> 32) ^ 1;
		uint32_t i, nv = asg_arc_n(g, v);
		const asg_arc_t *av = asg_arc_a(g, v);
		for (i = 0; i < nv; ++i)
			if (av[i].v == u) break;
		if (i == nv) g->arc[e].del = 1, ++n_asymm;
	}
	return n_asymm;
}
static void asymm_body(void *arg, int tid, size_t lo, size_t hi) { flt_par_t *p = (flt_par_t*)arg; p->cnt[tid] += asymm_range(p->g, (uint32_t)lo, (uint32_t)hi); }

int asg_arc_del_asymm(asg_t *g)
{
	uint32_t n_asymm = 0;
	if (clean_par(g)) {
		flt_par_t p;
		int t;
		memset(&p, 0, sizeof(p)); p.g = g;
		par_run(g->n_arc, 8192, ma_clean_threads(), asymm_body, &p);
		for (t = 0; t < PAR_MAX_THREADS; ++t) n_asymm += p.cnt[t];
	} else n_asymm = asymm_range(g, 0, g->n_arc);
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

static uint32_t short_range(asg_t *g, float drop_ratio, uint32_t lo, uint32_t hi)
{ /* asg.c:83-101 */
	uint32_t v, n_short = 0;
	for (v = lo; v < hi; ++v) {
		asg_arc_t *av = asg_arc_a(g, v);
		uint32_t i, thres, nv = asg_arc_n(g, v);
		if (nv < 2) continue;
		thres = (uint32_t)(av[0].ol * drop_ratio + .499);
		for (i = nv - 1; i >= 1 && av[i].ol < thres; --i);
		for (i = i + 1; i < nv; ++i) av[i].del = 1, ++n_short;
	}
	return n_short;
}
static void short_body(void *arg, int tid, size_t lo, size_t hi) { flt_par_t *p = (flt_par_t*)arg; p->cnt[tid] += short_range(p->g, p->ratio, (uint32_t)lo, (uint32_t)hi); }

int asg_arc_del_short(asg_t *g, float drop_ratio)
{
	uint32_t n_vtx = g->n_seq * 2, n_short = 0;
	if (clean_par(g)) {
		flt_par_t p;
		int t;
		memset(&p, 0, sizeof(p)); p.g = g; p.ratio = drop_ratio;
		par_run(n_vtx, 8192, ma_clean_threads(), short_body, &p);
		for (t = 0; t < PAR_MAX_THREADS; ++t) n_short += p.cnt[t];
	} else n_short = short_range(g, drop_ratio, 0, n_vtx);
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

/* speculative classification of every vertex end on a snapshot: kind (UE_*, or UE_GONE for a deleted read) and the read
 * of the unique live neighbour whose arcs were looked at (NO_READ if none) -- together with v's own read that is all the
 * state utg_end_kind() reads */
#define UE_GONE 255
#define NO_READ 0xffffffffu
typedef struct { const asg_t *g; uint8_t *kind; uint32_t *nb; } end_spec_t;

static void end_spec_body(void *arg, int tid, size_t lo, size_t hi)
{
	end_spec_t *e = (end_spec_t*)arg;
	size_t v;
	(void)tid;
	for (v = lo; v < hi; ++v) {
		uint64_t lw = 0;
		int k;
		e->nb[v] = NO_READ;
		if (e->g->seq[v >> 1].del) { e->kind[v] = UE_GONE; continue; }
		k = utg_end_kind(e->g, (uint32_t)v, &lw);
		e->kind[v] = (uint8_t)k;
		if (k == UE_MERGEABLE || k == UE_MULTI_NEI) e->nb[v] = (uint32_t)lw >> 1;
	}
}

typedef struct { uint8_t *kind; uint32_t *nb; dirty_t dm; } end_sweep_t;

static end_sweep_t *end_sweep_begin(const asg_t *g)
{
	end_sweep_t *w;
	end_spec_t e;
	size_t n_vtx = (size_t)g->n_seq * 2;
	if (!clean_par_ends(g)) return 0;
	w = (end_sweep_t*)calloc(1, sizeof(end_sweep_t));
	w->kind = (uint8_t*)malloc(n_vtx ? n_vtx : 1);
	w->nb = (uint32_t*)malloc((n_vtx ? n_vtx : 1) * 4);
	w->dm.map = (uint8_t*)calloc(g->n_seq ? g->n_seq : 1, 1);
	e.g = g; e.kind = w->kind; e.nb = w->nb;
	par_run(n_vtx, 4096, ma_clean_threads(), end_spec_body, &e);
	return w;
}

/* 1 if the sweep may skip vertex v: its snapshot classification is still valid and is not the one the sweep acts on */
static inline int end_sweep_skip(const end_sweep_t *w, uint32_t v, int wanted)
{
	if (w->dm.n && (w->dm.map[v >> 1] || (w->nb[v] != NO_READ && w->dm.map[w->nb[v]]))) return 0; /* stale: evaluate for real */
	return w->kind[v] != wanted;
}

static void end_sweep_end(end_sweep_t *w)
{
	if (w == 0) return;
	free(w->kind); free(w->nb); free(w->dm.map); free(w);
}

int asg_cut_tip(asg_t *g, int max_ext) /* asg.c:238-254 */
{
	asg64_v a = {0, 0, 0};
	uint32_t v, n_vtx = g->n_seq * 2, cnt = 0;
	size_t i;
	end_sweep_t *w = end_sweep_begin(g);
	for (v = 0; v < n_vtx; ++v) {
		if (w && end_sweep_skip(w, v, UE_TIP)) continue;
		if (g->seq[v >> 1].del) continue;
		if (utg_end_kind(g, v, 0) != UE_TIP) continue;
		if (asg_extend(g, v, max_ext, &a) == UE_MERGEABLE) continue; /* the unitig is longer than max_ext reads */
		for (i = 0; i < a.n; ++i) seq_drop_mark(g, (uint32_t)a.a[i] >> 1, w ? &w->dm : 0);
		++cnt;
	}
	end_sweep_end(w);
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
	end_sweep_t *w = end_sweep_begin(g);
	for (v = 0; v < n_vtx; ++v) {
		if (w && end_sweep_skip(w, v, UE_MULTI_NEI)) continue;
		if (g->seq[v >> 1].del) continue;
		if (utg_end_kind(g, v, 0) != UE_MULTI_NEI) continue;
		if (asg_extend(g, v, max_ext, &a) != UE_MULTI_NEI) continue;
		for (i = 0; i < a.n; ++i) seq_drop_mark(g, (uint32_t)a.a[i] >> 1, w ? &w->dm : 0);
		++cnt;
	}
	end_sweep_end(w);
	free(a.a);
	if (cnt > 0) asg_cleanup(g);
	fprintf(MA_LOG, "[M::%s] cut %d internal sequences\n", __func__, cnt);
	return cnt;
}

int asg_cut_biloop(asg_t *g, int max_ext) /* asg.c:274-306 */
{
	asg64_v a = {0, 0, 0};
	uint32_t v, n_vtx = g->n_seq * 2, cnt = 0;
	end_sweep_t *sw = end_sweep_begin(g);
	for (v = 0; v < n_vtx; ++v) {
		uint32_t i, nv, nw, w = UINT32_MAX, x, ov = 0, ox = 0;
		const asg_arc_t *av, *aw;
		if (sw && end_sweep_skip(sw, v, UE_MULTI_NEI)) continue;
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
			if (sw) { dirty_mark(&sw->dm, w >> 1); dirty_mark(&sw->dm, x >> 1); }
			++cnt;
		}
	}
	end_sweep_end(sw);
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
	bub_info_t *info; /* indexed by vertex (the sweep's own scratch) ... */
	uint32_t *hkey, hcap, hn; bub_info_t *hval; u32_v hused; /* ... or a small open-addressing map (the probing threads: a probe touches a handful of vertices) */
	u32_v ready;   /* vertices whose in-arcs have all been visited */
	u32_v tips;    /* visited dead ends */
	u32_v touched; /* visited vertices */
	u32_v arcs;    /* visited arcs */
} bub_buf_t;

#define BUB_HEMPTY 0xffffffffu
static void bub_map_grow(bub_buf_t *b)
{
	uint32_t i, ocap = b->hcap, *okey = b->hkey;
	bub_info_t *oval = b->hval;
	b->hcap = ocap ? ocap << 1 : 256;
	b->hkey = (uint32_t*)malloc((size_t)b->hcap * 4);
	b->hval = (bub_info_t*)malloc((size_t)b->hcap * sizeof(bub_info_t));
	memset(b->hkey, 0xff, (size_t)b->hcap * 4);
	b->hused.n = 0; b->hn = 0;
	for (i = 0; i < ocap; ++i)
		if (okey[i] != BUB_HEMPTY) {
			uint32_t s = (okey[i] * 2654435761u) & (b->hcap - 1);
			while (b->hkey[s] != BUB_HEMPTY) s = (s + 1) & (b->hcap - 1);
			b->hkey[s] = okey[i]; b->hval[s] = oval[i]; u32_push(&b->hused, s); ++b->hn;
		}
	free(okey); free(oval);
}

/* scratch record of vertex v: zero (unseen) on first access */
static inline bub_info_t *bub_info(bub_buf_t *b, uint32_t v)
{
	uint32_t s;
	if (b->info) return &b->info[v];
	if (b->hcap == 0 || b->hn * 2 >= b->hcap) bub_map_grow(b);
	for (s = (v * 2654435761u) & (b->hcap - 1); b->hkey[s] != BUB_HEMPTY; s = (s + 1) & (b->hcap - 1))
		if (b->hkey[s] == v) return &b->hval[s];
	b->hkey[s] = v; memset(&b->hval[s], 0, sizeof(bub_info_t));
	u32_push(&b->hused, s); ++b->hn;
	return &b->hval[s];
}

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
		uint32_t u = bub_info(b, v)->parent;
		g->seq[v >> 1].del = 0;
		arc_flag(g, u, v, 0);
		arc_flag(g, v ^ 1, u ^ 1, 0);
		v = u;
	} while (v != v0);
}

/* try to pop one bubble rooted at v0 (asg.c:360-409); returns 1 | n_tips<<32 when popped */
static uint64_t bub_pop1(asg_t *g, uint32_t v0, int max_dist, bub_buf_t *b, int apply) /* apply = 0: probe only, g is not modified */
{
	uint32_t i, n_pending = 0;
	uint64_t ret = 0;
	size_t k;
	b->ready.n = b->tips.n = b->touched.n = b->arcs.n = 0;
	if (g->seq[v0 >> 1].del) return 0;
	if ((uint32_t)g->idx[v0] < 2) return 0;
	{ bub_info_t *t0 = bub_info(b, v0); t0->cnt = t0->dist = 0; }
	u32_push(&b->ready, v0);
	do {
		uint32_t v = b->ready.a[--b->ready.n], d, c;
		uint32_t nv = asg_arc_n(g, v);
		{ const bub_info_t *tv = bub_info(b, v); d = tv->dist; c = tv->cnt; } /* values, not the pointer: the map may grow below */
		const asg_arc_t *av = asg_arc_a(g, v);
		assert(nv > 0);
		for (i = 0; i < nv; ++i) {
			uint32_t w = av[i].v, l = (uint32_t)av[i].ul;
			bub_info_t *t;
			if (w == v0) goto reset; /* a cycle through the source */
			if (av[i].del) continue;
			u32_push(&b->arcs, (uint32_t)(g->idx[v] >> 32) + i);
			if (d + l > (uint32_t)max_dist) break; /* too far */
			t = bub_info(b, w);
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
	if (apply) bub_backtrack(g, v0, b);
	ret = 1 | (uint64_t)b->tips.n << 32;
reset:
	if (b->info) {
		for (k = 0; k < b->touched.n; ++k) {
			bub_info_t *t = &b->info[b->touched.a[k]];
			t->seen = 0, t->cnt = 0, t->dist = 0;
		}
	} else { /* map mode: forget every record of this probe */
		for (k = 0; k < b->hused.n; ++k) b->hkey[b->hused.a[k]] = BUB_HEMPTY;
		b->hused.n = 0; b->hn = 0;
	}
	return ret;
}

/* Bubble popping on big graphs.  The sources are taken in the reference's order in BLOCKS: the probes of a block run
 * speculatively on worker threads against the graph as the previous block left it (each thread with its own scratch),
 * recording the reads whose state they looked at (the touched vertices plus the source); then the block is committed in
 * order: a probe that found nothing is trusted unless a pop committed earlier IN THE SAME BLOCK modified one of its reads
 * (epoch-stamped dirty map); everything else -- stale probes and probes that found a bubble -- is evaluated for real by
 * the sequential routine.  Small blocks keep the stale fraction low where pops are dense; the pool is persistent
 * (two barriers per block). */
typedef struct { uint64_t off; uint32_t n; uint16_t tid; uint8_t pop; } bub_spec_t;
typedef struct { bub_buf_t buf; u32_v log; char pad[256]; } bub_thread_t; /* one per worker, padded: the counters inside are bumped on every push */
typedef struct {
	asg_t *g; int max_dist, nth; const uint32_t *cand; bub_spec_t *spec; bub_thread_t *th;
	size_t blk_lo, blk_hi, next; int quit;
	pthread_mutex_t gate; /* held by the main thread until the barriers are sized for the threads that really started */
	pthread_barrier_t start, done;
} bub_par_t;

static inline uint32_t bub_is_source(const asg_t *g, uint32_t v)
{
	uint32_t i, n_live = 0, nv = asg_arc_n(g, v);
	const asg_arc_t *av = asg_arc_a(g, v);
	if (nv < 2 || g->seq[v >> 1].del) return 0;
	for (i = 0; i < nv; ++i)
		if (!av[i].del) ++n_live;
	return n_live > 1;
}

static void bub_spec_range(bub_par_t *p, int tid)
{
	bub_buf_t *b = &p->th[tid].buf;
	u32_v *log = &p->th[tid].log;
	log->n = 0; /* b->info stays NULL: the probing threads keep their scratch in a small map (no page-fault storm on huge zeroed arrays) */
	for (;;) {
		size_t k, i, lo = __atomic_fetch_add(&p->next, 32, __ATOMIC_RELAXED), hi;
		if (lo >= p->blk_hi) break;
		hi = lo + 32 < p->blk_hi ? lo + 32 : p->blk_hi;
		for (k = lo; k < hi; ++k) {
			bub_spec_t *s = &p->spec[k - p->blk_lo];
			if (!bub_is_source(p->g, p->cand[k])) { s->pop = 0; s->tid = (uint16_t)tid; s->off = log->n; s->n = 0; continue; } /* reads only its own arcs */
			s->pop = bub_pop1(p->g, p->cand[k], p->max_dist, b, 0) != 0;
			s->tid = (uint16_t)tid; s->off = log->n; s->n = (uint32_t)b->touched.n;
			for (i = 0; i < b->touched.n; ++i) u32_push(log, b->touched.a[i]);
		}
	}
}

typedef struct { bub_par_t *p; int tid; } bub_worker_t;
static void *bub_worker(void *arg)
{
	bub_worker_t *w = (bub_worker_t*)arg;
	pthread_mutex_lock(&w->p->gate); pthread_mutex_unlock(&w->p->gate);
	for (;;) {
		pthread_barrier_wait(&w->p->start);
		if (w->p->quit) break;
		bub_spec_range(w->p, w->tid);
		pthread_barrier_wait(&w->p->done);
	}
	return 0;
}

#define BUB_BLOCK_MIN 256u
#define BUB_BLOCK_MAX 65536u
static uint64_t pop_bubble_par(asg_t *g, int max_dist, bub_buf_t *b0)
{
	const uint32_t n_vtx = g->n_seq * 2;
	int nth = ma_clean_threads(), t, n_started = 0;
	uint32_t v, *cand, nc = 0, *stamp, epoch = 0;
	uint64_t n_pop = 0;
	size_t k, i, blk = 2048, tot_stale = 0, tot_spec_pop = 0, n_blocks = 0, tot_touched = 0;
	const int timing = getenv("MA_PIPE_TIMING") != 0;
	double t_spec = 0, t_commit = 0, t0;
	bub_par_t p;
	bub_worker_t wk[PAR_MAX_THREADS];
	pthread_t th[PAR_MAX_THREADS];
	const double t_begin = sys_realtime();
	double t_ready;
	cand = (uint32_t*)malloc(((size_t)n_vtx + 1) * 4);
	for (v = 0; v < n_vtx; ++v) /* sources at entry; deletions only ever remove sources */
		if (asg_arc_n(g, v) >= 2 && !g->seq[v >> 1].del) cand[nc++] = v;
	if (nc < 100000 && !getenv("MA_CLEAN_PAR_MIN")) { /* few sources: not worth a thread pool */
		for (k = 0; k < nc; ++k)
			if (bub_is_source(g, cand[k])) n_pop += bub_pop1(g, cand[k], max_dist, b0, 1);
		free(cand);
		return n_pop;
	}
	memset(&p, 0, sizeof(p));
	p.g = g; p.max_dist = max_dist; p.cand = cand;
	p.spec = (bub_spec_t*)malloc((size_t)BUB_BLOCK_MAX * sizeof(bub_spec_t));
	p.th = (bub_thread_t*)calloc(nth, sizeof(bub_thread_t));
	stamp = (uint32_t*)calloc(g->n_seq ? g->n_seq : 1, 4); /* read r is dirty in the current block iff stamp[r] == epoch */
	pthread_mutex_init(&p.gate, 0);
	pthread_mutex_lock(&p.gate);
	for (t = 1; t < nth; ++t) {
		wk[t].p = &p; wk[t].tid = t;
		if (pthread_create(&th[t], 0, bub_worker, &wk[t]) != 0) break;
	}
	n_started = t - 1;
	nth = n_started + 1; /* the threads that really exist */
	pthread_barrier_init(&p.start, 0, (unsigned)nth);
	pthread_barrier_init(&p.done, 0, (unsigned)nth);
	pthread_mutex_unlock(&p.gate);
	t_ready = sys_realtime();
	p.nth = nth;
	for (k = 0; k < nc;) {
		size_t hi = k + blk < nc ? k + blk : nc, n_dirty = 0, n_stale = 0, n_blk_pop = 0;
		p.blk_lo = k; p.blk_hi = hi; p.next = k;
		t0 = timing ? sys_realtime() : 0;
		pthread_barrier_wait(&p.start);
		bub_spec_range(&p, 0);
		pthread_barrier_wait(&p.done);
		if (timing) { t_spec += sys_realtime() - t0; t0 = sys_realtime(); }
		++epoch; ++n_blocks;
		for (; k < hi; ++k) { /* commit in the reference's order */
			const bub_spec_t *s = &p.spec[k - p.blk_lo];
			uint64_t r;
			int stale = 0;
			v = cand[k];
			if (n_dirty) {
				const uint32_t *t_ = p.th[s->tid].log.a + s->off;
				stale = stamp[v >> 1] == epoch;
				for (i = 0; !stale && i < s->n; ++i) stale = stamp[t_[i] >> 1] == epoch;
				n_stale += stale;
			}
			tot_spec_pop += s->pop; tot_touched += s->n;
			if (!stale && !s->pop) continue;        /* the probe saw exactly today's state and found nothing */
			if (!bub_is_source(g, v)) continue;     /* asg.c:421-428 on the current state */
			r = bub_pop1(g, v, max_dist, b0, 1);
			if (timing && !stale && s->pop && !r) { static int warned = 0; if (warned++ < 5) fprintf(stderr, "[T::pop_bubble] BUG? probe of %u said bubble (touched %u), the sweep found none (touched %zu)\n", v, s->n, b0->touched.n); }
			if (r) {
				n_pop += r; ++n_blk_pop;
				stamp[v >> 1] = epoch;
				for (i = 0; i < b0->touched.n; ++i) stamp[b0->touched.a[i] >> 1] = epoch;
				n_dirty += b0->touched.n + 1;
			}
		}
		tot_stale += n_stale;
		if (timing) t_commit += sys_realtime() - t0;
		/* adapt the block size: grow while pops are rare, shrink when much of a block went stale */
		if (n_stale * 8 > (hi - p.blk_lo) && blk > BUB_BLOCK_MIN) blk >>= 1;
		else if (n_blk_pop <= 1 && blk < BUB_BLOCK_MAX) blk <<= 1;
	}
	if (timing) fprintf(stderr, "[T::pop_bubble] %u sources, %zu blocks, %zu stale, %zu probes found a bubble, %llu popped, %.1f touched/probe; setup %.2f speculation %.2f commit %.2f ms (%d threads)\n",
	                    nc, n_blocks, tot_stale, tot_spec_pop, (unsigned long long)(n_pop & 0xffffffffu), nc ? (double)tot_touched / nc : 0., (t_ready - t_begin) * 1e3, t_spec * 1e3, t_commit * 1e3, nth);
	p.quit = 1;
	pthread_barrier_wait(&p.start);
	for (t = 1; t <= n_started; ++t) pthread_join(th[t], 0);
	pthread_barrier_destroy(&p.start); pthread_barrier_destroy(&p.done); pthread_mutex_destroy(&p.gate);
	for (t = 0; t < nth; ++t) {
		bub_buf_t *b = &p.th[t].buf;
		free(b->hkey); free(b->hval); free(b->hused.a); free(b->ready.a); free(b->tips.a); free(b->touched.a); free(b->arcs.a); free(p.th[t].log.a);
	}
	free(p.th); free(p.spec); free(cand); free(stamp);
	return n_pop;
}

int asg_pop_bubble(asg_t *g, int max_dist) /* asg.c:412-433 */
{
	uint32_t v, n_vtx = g->n_seq * 2;
	uint64_t n_pop = 0;
	bub_buf_t b;
	if (!g->is_symm) asg_symm(g);
	memset(&b, 0, sizeof(b));
	b.info = (bub_info_t*)calloc(n_vtx ? n_vtx : 1, sizeof(bub_info_t));
	if (clean_par(g)) n_pop = pop_bubble_par(g, max_dist, &b);
	else
		for (v = 0; v < n_vtx; ++v) {
			uint32_t i, n_live = 0, nv = asg_arc_n(g, v);
			const asg_arc_t *av = asg_arc_a(g, v);
			if (nv < 2 || g->seq[v >> 1].del) continue;
			for (i = 0; i < nv; ++i)
				if (!av[i].del) ++n_live;
			if (n_live > 1) n_pop += bub_pop1(g, v, max_dist, &b, 1);
		}
	free(b.info); free(b.ready.a); free(b.tips.a); free(b.touched.a); free(b.arcs.a);
	if (n_pop) asg_cleanup(g);
	fprintf(MA_LOG, "[M::%s] popped %d bubbles and trimmed %d tips\n", __func__, (uint32_t)n_pop, (uint32_t)(n_pop >> 32));
	return (int)n_pop;
}
