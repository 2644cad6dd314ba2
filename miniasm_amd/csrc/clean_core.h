/* clean_core.h -- the order-dependent graph cleaners of the reference (asg.c:199-433: asg_cut_tip, asg_cut_internal,
 * asg_cut_biloop, asg_pop_bubble) as a data-parallel FIXPOINT over versioned state, written once for the device
 * kernels (csrc/clean.hip) and for the host test harness (tests/clean_host.cpp).
 *
 * The reference sweeps the vertices in id order and mutates the graph as it goes: vertex v sees the deletions made
 * for every vertex u < v.  Here every vertex is evaluated at once against a VERSIONED view of the graph:
 *
 *     a read / an arc is dead for vertex v   <=>   it is dead in the base graph, or its stamp is < v,
 *
 * where stamp[cell] = the smallest vertex whose action deletes that cell.  One iteration evaluates all vertices
 * under the stamps of the previous iteration and rebuilds the stamps from the actions it finds; the iteration is
 * repeated until the stamps do not change.  By induction on the vertex id the fixpoint is unique and equals the
 * sequential sweep: vertex 0 only sees the base graph, and the view of vertex v depends only on stamps < v, i.e. on
 * the (by induction final) actions of smaller vertices.  The number of iterations is the depth of the longest
 * chain of dependent actions (a handful), not the number of actions.
 *
 * The per-vertex rules (what is a tip, how far a unitig end is extended, what a bubble pop deletes and resurrects)
 * are the reference's definitions -- they are the specification -- re-expressed over the view.
 *
 * ASSUMPTION: a cell deleted by a smaller vertex stays deleted -- stamps only ever decrease.  One action of the reference can break it:
 * asg_bub_backtrack sets seq.del = 0 for every vertex of the best path (asg.c:352) whatever the read's state was.  "Dead, then alive again from v0 on"
 * is not a stamp.  When can a probe put a dead read on its path?  It only follows live arcs (asg.c:377), so the read must be dead AND own a live arc.
 *
 *   CLAIM.  On a graph that is symmetric (u->v live  <=>  v'->u' live) and clean (no live arc touches a read with seq.del set) at the start of the sweep
 *   -- what asg_symm + asg_cleanup leave, i.e. every call the pipeline makes (main.c:160-187; asg_arc_del_short re-establishes both, asg.c:95-98) --
 *   no probe of the sweep ever meets a dead read: the situation cannot arise.
 *   PROOF.  Induction over the successful pops of the sweep, invariant: the live arcs are symmetric and none touches a dead read.  Let the pop from v0
 *   succeed and let w be a visited vertex that is not on the best path.  (i) Arcs INTO w: a fresh visit sets r(w) = live arcs out of w' (asg.c:383), by
 *   symmetry the number of live arcs into w; the probe succeeds only when no vertex is pending (asg.c:400), i.e. r(w) went to 0, one decrement per arc
 *   walked into w from an expanded vertex: EVERY live arc u->w was walked, sits in b->e, and is deleted together with w'->u' (asg.c:345-348).
 *   (ii) Arcs OUT of w: either w was expanded -- then all its live arcs were walked (a walk that stops early, asg.c:379, makes the probe fail) and are
 *   deleted with their mirrors x'->w' -- or w is a tip with no arcs at all (asg.c:393 counts deleted arcs too), and then w' has no arc into it.
 *   So after the pop every arc with an end in w's read, on either strand, is deleted; the arcs it brings back (asg.c:353-354) join path vertices, whose
 *   reads it revives -- both ends alive; every deletion and revival is done to an arc and its mirror.  The invariant holds again, and under it a probe
 *   that follows live arcs reaches live reads only.  (The rules of cut_tip / cut_internal / cut_biloop delete reads with asg_seq_del, which deletes all
 *   their arcs and mirrors; nothing else in a sweep revives anything.)  QED.
 *
 * Outside that contract the reference still answers, and so must the drop-in: the per-symbol asg_pop_bubble can be handed a graph whose is_symm flag is
 * set although it is not symmetric (the extra arc into a popped read survives and a later pop can walk it), or one that was never cleaned (a read with
 * seq.del set and live arcs lies on a path).  cl_bubble_stamp reports a path vertex that is dead in the popper's view; if the FINAL view still holds
 * such a pop the fixpoint's answer is void and the call is run again as the reference's own sequential sweep on one lane (cl_bubble_sweep_seq below,
 * csrc/clean.hip: k_clean_bubble_seq) -- slow and exact.  tests/test_gpu_graph_api.py holds a witness of either kind.
 */
#ifndef CLEAN_CORE_H
#define CLEAN_CORE_H

#include <stdint.h>

#if defined(__HIPCC__)
#define CL_HD __host__ __device__ __forceinline__
#define CL_MIN_U32(p, x) atomicMin((p), (x))
#else
#define CL_HD static inline
#define CL_MIN_U32(p, x) do { if ((x) < *(p)) *(p) = (x); } while (0)
#endif

#define CL_NONE 0xffffffffu
#define CL_ADEL 0x80000000u

/* end kinds (reference asg.c:200-203) */
#define CL_MERGEABLE 0
#define CL_TIP       1
#define CL_MULTI_OUT 2
#define CL_MULTI_NEI 3

typedef struct {
	const uint32_t *av, *alen, *aol;   /* arcs: target vertex, length, overlap | del<<31 (base flags) */
	const unsigned long long *idx;     /* per vertex: first arc << 32 | arc count */
	const uint8_t *sdel;               /* per read: seq.del (base flag) */
	const uint32_t *rst, *ast;         /* stamps of the previous iteration, per read / per arc */
	uint32_t n_vtx;
	uint32_t no_stamps;                /* first iteration: there are no stamps yet (rst / ast are not read) */
} cl_view_t;

typedef struct { uint32_t *rst, *ast; } cl_stamps_t; /* stamps being built by this iteration */

CL_HD int cl_arc_dead(const cl_view_t *g, uint32_t e, uint32_t me) { return (g->aol[e] >> 31) || (!g->no_stamps && g->ast[e] < me); }
CL_HD int cl_seq_dead(const cl_view_t *g, uint32_t r, uint32_t me) { return g->sdel[r] || (!g->no_stamps && g->rst[r] < me); }
CL_HD uint32_t cl_first(const cl_view_t *g, uint32_t v) { return (uint32_t)(g->idx[v] >> 32); }
CL_HD uint32_t cl_count(const cl_view_t *g, uint32_t v) { return (uint32_t)g->idx[v]; }

CL_HD uint32_t cl_live_out(const cl_view_t *g, uint32_t v, uint32_t me)
{
	uint32_t st = cl_first(g, v), n = cl_count(g, v), i, live = 0;
	for (i = 0; i < n; ++i) live += !cl_arc_dead(g, st + i, me);
	return live;
}

/* What lies beyond v's far end, i.e. out of v^1 (asg.c:205-223): nothing (TIP), a fork (MULTI_OUT), one neighbour that has
 * other ways in (MULTI_NEI), or one neighbour reached only from here (MERGEABLE).  *e_one = the unique live arc. */
CL_HD int cl_end_kind(const cl_view_t *g, uint32_t v, uint32_t me, uint32_t *e_one)
{
	uint32_t st = cl_first(g, v ^ 1), n = cl_count(g, v ^ 1), i, live = 0, last = 0;
	for (i = 0; i < n; ++i)
		if (!cl_arc_dead(g, st + i, me)) last = st + i, ++live;
	if (live == 0) return CL_TIP;
	if (live > 1) return CL_MULTI_OUT;
	*e_one = last;
	return cl_live_out(g, g->av[last] ^ 1, me) != 1 ? CL_MULTI_NEI : CL_MERGEABLE;
}

/* asg_seq_del (asg.h:64-77) as stamps: the read, every arc of both of its vertices, and the arcs that mirror them */
CL_HD void cl_stamp_read(const cl_view_t *g, cl_stamps_t s, uint32_t r, uint32_t me)
{
	uint32_t k;
	CL_MIN_U32(&s.rst[r], me);
	for (k = 0; k < 2; ++k) {
		uint32_t v = r << 1 | k, st = cl_first(g, v), n = cl_count(g, v), i;
		for (i = 0; i < n; ++i) {
			uint32_t t = g->av[st + i] ^ 1, st2 = cl_first(g, t), n2 = cl_count(g, t), j;
			CL_MIN_U32(&s.ast[st + i], me);
			for (j = 0; j < n2; ++j)
				if (g->av[st2 + j] == (v ^ 1)) CL_MIN_U32(&s.ast[st2 + j], me);
		}
	}
}

/* asg_arc_del(g, v, w, 1) (asg.h:55-61) as stamps: every arc v -> w */
CL_HD void cl_stamp_arcs(const cl_view_t *g, cl_stamps_t s, uint32_t v, uint32_t w, uint32_t me)
{
	uint32_t st = cl_first(g, v), n = cl_count(g, v), i;
	for (i = 0; i < n; ++i)
		if (g->av[st + i] == w) CL_MIN_U32(&s.ast[st + i], me);
}

/* Walk the unitig that starts at v for at most max_ext steps (asg.c:225-236).  stamp != 0: delete every read on the way.
 * Returns the kind that stopped the walk (MERGEABLE: it did not stop); *last = the last vertex reached. */
CL_HD int cl_extend(const cl_view_t *g, uint32_t v, int max_ext, uint32_t me, int stamp, cl_stamps_t s, uint32_t *last)
{
	int kind;
	for (;;) {
		uint32_t e = 0;
		if (stamp) cl_stamp_read(g, s, v >> 1, me);
		kind = cl_end_kind(g, v ^ 1, me, &e);
		if (kind != CL_MERGEABLE) break;
		v = g->av[e];
		if (--max_ext <= 0) { if (stamp) cl_stamp_read(g, s, v >> 1, me); break; }
	}
	*last = v;
	return kind;
}

/* ---- the three short-unitig rules; each returns 1 if vertex v acts (and has then written its stamps) ---- */

/* asg.c:238-254: a unitig that starts at a dead end and is over within max_ext reads goes */
CL_HD int cl_rule_tip(const cl_view_t *g, cl_stamps_t s, uint32_t v, int max_ext)
{
	uint32_t e, last;
	if (cl_seq_dead(g, v >> 1, v)) return 0;
	if (cl_end_kind(g, v, v, &e) != CL_TIP) return 0;
	if (cl_extend(g, v, max_ext, v, 0, s, &last) == CL_MERGEABLE) return 0;
	cl_extend(g, v, max_ext, v, 1, s, &last);
	return 1;
}

/* asg.c:256-272: a short unitig wedged between two forks goes */
CL_HD int cl_rule_internal(const cl_view_t *g, cl_stamps_t s, uint32_t v, int max_ext)
{
	uint32_t e, last;
	if (cl_seq_dead(g, v >> 1, v)) return 0;
	if (cl_end_kind(g, v, v, &e) != CL_MULTI_NEI) return 0;
	if (cl_extend(g, v, max_ext, v, 0, s, &last) != CL_MULTI_NEI) return 0;
	cl_extend(g, v, max_ext, v, 1, s, &last);
	return 1;
}

/* asg.c:274-306: w -> v ... x' and w -> x with the weaker overlap on the x side: cut w -> x */
CL_HD int cl_rule_biloop(const cl_view_t *g, cl_stamps_t s, uint32_t v, int max_ext)
{
	uint32_t e = 0, last, w, x, st, n, i, ov = 0, ox = 0;
	if (cl_seq_dead(g, v >> 1, v)) return 0;
	if (cl_end_kind(g, v, v, &e) != CL_MULTI_NEI) return 0;
	if (cl_extend(g, v, max_ext, v, 0, s, &last) != CL_MULTI_OUT) return 0;
	x = last ^ 1;
	w = g->av[e] ^ 1; /* the one live arc out of v^1 */
	st = cl_first(g, w); n = cl_count(g, w);
	for (i = 0; i < n; ++i) {
		if (cl_arc_dead(g, st + i, v)) continue;
		if (g->av[st + i] == x) ox = g->aol[st + i] & 0x7fffffffu;
		if (g->av[st + i] == v) ov = g->aol[st + i] & 0x7fffffffu;
	}
	if (ov == 0 && ox == 0) return 0;
	if (ov > ox) {
		cl_stamp_arcs(g, s, w, x, v);
		cl_stamp_arcs(g, s, x ^ 1, w ^ 1, v);
		return 1;
	}
	return 0;
}

/* ---- bubble popping (asg.c:312-433) ----
 * One source at a time per thread, its traversal state in a private open-addressing table (a probe touches a handful of
 * vertices).  The traversal order (LIFO work list, arcs in list order) and the best-parent rule are the reference's. */
typedef struct {
	uint32_t key;      /* vertex, CL_NONE = empty slot */
	uint32_t p, d, c;  /* best predecessor, shortest distance from the source, most reads on a path */
	uint32_t r;        /* in-arcs not yet seen */
	uint32_t fl;       /* CL_B_* */
} cl_binfo_t;
#define CL_B_EXPANDED 1u
#define CL_B_TIP 2u
#define CL_B_PATH 4u

typedef struct {
	cl_binfo_t *tab;   /* [cap], cap a power of two */
	uint32_t *used;    /* [cap] slots in use */
	uint32_t *stack;   /* [cap] */
	uint32_t cap, n_used;
} cl_bscratch_t;

CL_HD uint32_t cl_bhash(uint32_t v, uint32_t cap) { return (v * 0x9E3779B1u) & (cap - 1); }

CL_HD cl_binfo_t *cl_bfind(const cl_bscratch_t *b, uint32_t v)
{
	uint32_t s = cl_bhash(v, b->cap);
	for (;;) {
		if (b->tab[s].key == v) return &b->tab[s];
		if (b->tab[s].key == CL_NONE) return 0;
		s = (s + 1) & (b->cap - 1);
	}
}

/* the record of v, created empty (never seen) on first access; 0 when the table is full */
CL_HD cl_binfo_t *cl_bget(cl_bscratch_t *b, uint32_t v, int *fresh)
{
	uint32_t s = cl_bhash(v, b->cap);
	for (;;) {
		if (b->tab[s].key == v) { *fresh = 0; return &b->tab[s]; }
		if (b->tab[s].key == CL_NONE) break;
		s = (s + 1) & (b->cap - 1);
	}
	if (b->n_used * 4 >= b->cap * 3) return 0;
	b->tab[s].key = v; b->tab[s].p = CL_NONE; b->tab[s].d = b->tab[s].c = b->tab[s].r = b->tab[s].fl = 0;
	b->used[b->n_used++] = s;
	*fresh = 1;
	return &b->tab[s];
}

CL_HD void cl_bclear(cl_bscratch_t *b)
{
	uint32_t i;
	for (i = 0; i < b->n_used; ++i) b->tab[b->used[i]].key = CL_NONE;
	b->n_used = 0;
}

/* Try to pop the bubble rooted at v0.  Returns 1 (popped: the table holds the traversal, *sink / *n_tips set), 0 (no bubble:
 * table cleared) or -1 (scratch too small: table cleared, the caller grows it and repeats the iteration). */
CL_HD int cl_bubble_probe(const cl_view_t *g, uint32_t v0, uint32_t max_dist, cl_bscratch_t *b, uint32_t *sink, uint32_t *n_tips)
{
	uint32_t n_stack = 0, n_pending = 0, tips = 0;
	if (cl_seq_dead(g, v0 >> 1, v0)) return 0;
	if (cl_count(g, v0) < 2 || cl_live_out(g, v0, v0) < 2) return 0; /* asg.c:421-427 */
	b->stack[n_stack++] = v0;
	do {
		const uint32_t v = b->stack[--n_stack];
		uint32_t d = 0, c = 0, st = cl_first(g, v), nv = cl_count(g, v), i;
		if (v != v0) { cl_binfo_t *tv = cl_bfind(b, v); d = tv->d; c = tv->c; tv->fl |= CL_B_EXPANDED; }
		for (i = 0; i < nv; ++i) {
			const uint32_t e = st + i, w = g->av[e], l = g->alen[e];
			cl_binfo_t *t;
			int fresh = 0;
			if (w == v0) goto fail;                 /* a cycle through the source (checked before the arc's own flag) */
			if (cl_arc_dead(g, e, v0)) continue;
			if (d + l > max_dist) goto fail;        /* too far: the reference leaves the list early and gives up */
			t = cl_bget(b, w, &fresh);
			if (t == 0) { cl_bclear(b); return -1; }
			if (fresh) {
				t->p = v; t->d = d + l;             /* c stays 0 on the first visit, as in the reference */
				t->r = cl_live_out(g, w ^ 1, v0);
				++n_pending;
			} else {
				if (c + 1 > t->c || (c + 1 == t->c && d + l > t->d)) t->p = v;
				if (c + 1 > t->c) t->c = c + 1;
				if (d + l < t->d) t->d = d + l;
			}
			if (--t->r == 0) {
				if (cl_count(g, w)) b->stack[n_stack++] = w; /* deleted arcs count here too (asg.c:393) */
				else t->fl |= CL_B_TIP, ++tips;
				--n_pending;
			}
		}
		if (n_stack == 0) goto fail;
	} while (n_stack > 1 || n_pending);
	*sink = b->stack[0];
	*n_tips = tips;
	return 1;
fail:
	cl_bclear(b);
	return 0;
}

/* The net effect of a pop (asg.c:338-357) as stamps: every touched read goes unless one of its vertices is on the best path;
 * every arc that was walked, and its mirror, goes unless it joins two consecutive vertices of the best path.  Clears the table. */
CL_HD int cl_bubble_stamp(const cl_view_t *g, cl_stamps_t s, uint32_t v0, uint32_t sink, cl_bscratch_t *b)
{ /* returns 1 if the pop would resurrect a read that is already dead in v0's view (see the ASSUMPTION at the top) */
	uint32_t i, v = sink;
	int resurrects = 0;
	while (v != v0) { cl_binfo_t *t = cl_bfind(b, v); t->fl |= CL_B_PATH; resurrects |= cl_seq_dead(g, v >> 1, v0); v = t->p; }
	for (i = 0; i <= b->n_used; ++i) { /* the source (i == n_used) and every expanded vertex: all their live arcs were walked */
		uint32_t u, st, n, k;
		if (i < b->n_used) {
			const cl_binfo_t *t = &b->tab[b->used[i]];
			const cl_binfo_t *o = cl_bfind(b, t->key ^ 1);
			if (!(t->fl & CL_B_PATH) && !(o && (o->fl & CL_B_PATH))) CL_MIN_U32(&s.rst[t->key >> 1], v0);
			if (!(t->fl & CL_B_EXPANDED)) continue;
			u = t->key;
		} else u = v0;
		st = cl_first(g, u); n = cl_count(g, u);
		for (k = 0; k < n; ++k) {
			const uint32_t e = st + k, w = g->av[e];
			const cl_binfo_t *tw, *tu;
			if (cl_arc_dead(g, e, v0)) continue;
			/* the pair {u -> w, w' -> u'} comes back when it joins consecutive path vertices, read in either direction */
			tw = cl_bfind(b, w); tu = cl_bfind(b, u ^ 1);
			if (tw && (tw->fl & CL_B_PATH) && tw->p == u) continue;
			if (tu && (tu->fl & CL_B_PATH) && tu->p == (w ^ 1)) continue;
			CL_MIN_U32(&s.ast[e], v0);
			cl_stamp_arcs(g, s, w ^ 1, u ^ 1, v0);
		}
	}
	cl_bclear(b);
	return resurrects;
}


/* ---- asg_pop_bubble as the reference runs it (asg.c:360-433): one sweep over the vertices that MUTATES the base flags as it goes.  The fallback for graphs
 * outside the fixpoint's contract (see the ASSUMPTION at the top); also what MA_BUBBLE_SEQ=1 forces, so that the tests can hold it against the reference
 * on every graph they have.  info[n_vtx] zeroed by the caller; stk / seen: n_vtx words each, walked: one word per arc.
 * Returns 0, or -1 where the reference's own assertion (asg.c:391: more walks into a vertex than it has arcs in) would end the process. */
typedef struct { uint32_t p, d, c, r; } cl_seqinfo_t; /* best parent, shortest distance, most reads, arcs still to come in | visited << 31 */

CL_HD void cl_seq_arc_set(const uint32_t *av, uint32_t *aol, const unsigned long long *idx, uint32_t v, uint32_t w, int del)
{ /* asg_arc_del (asg.h:53-60): every arc v -> w */
	const uint32_t st = (uint32_t)(idx[v] >> 32), n = (uint32_t)idx[v];
	uint32_t i;
	for (i = 0; i < n; ++i)
		if (av[st + i] == w) aol[st + i] = del ? aol[st + i] | CL_ADEL : aol[st + i] & ~CL_ADEL;
}
CL_HD uint32_t cl_seq_live_out(const uint32_t *aol, const unsigned long long *idx, uint32_t v)
{
	const uint32_t st = (uint32_t)(idx[v] >> 32), n = (uint32_t)idx[v];
	uint32_t i, live = 0;
	for (i = 0; i < n; ++i) live += !(aol[st + i] >> 31);
	return live;
}

CL_HD int cl_bubble_sweep_seq(const uint32_t *au, const uint32_t *av, const uint32_t *alen, uint32_t *aol, const unsigned long long *idx, uint8_t *sdel, uint32_t n_vtx,
                              uint32_t max_dist, cl_seqinfo_t *info, uint32_t *stk, uint32_t *seen, uint32_t *walked, unsigned long long *n_pop, unsigned long long *n_tips)
{
	uint32_t v0;
	*n_pop = *n_tips = 0;
	for (v0 = 0; v0 < n_vtx; ++v0) {
		uint32_t n_stk = 0, n_seen = 0, n_walk = 0, pending = 0, tips = 0, k;
		int popped = 0;
		if ((uint32_t)idx[v0] < 2 || sdel[v0 >> 1] || cl_seq_live_out(aol, idx, v0) < 2) continue; /* asg.c:421-427, 365-366 */
		info[v0].c = info[v0].d = 0;
		stk[n_stk++] = v0;
		for (;;) {
			const uint32_t v = stk[--n_stk], d = info[v].d, c = info[v].c, st = (uint32_t)(idx[v] >> 32), nv = (uint32_t)idx[v];
			uint32_t i;
			for (i = 0; i < nv; ++i) {
				const uint32_t w = av[st + i], l = alen[st + i];
				cl_seqinfo_t *t = &info[w];
				if (w == v0) goto reset;
				if (aol[st + i] >> 31) continue;
				walked[n_walk++] = st + i;
				if (d + l > max_dist) break;
				if (!(t->r >> 31)) { /* first visit; c keeps the zero it was reset to */
					seen[n_seen++] = w;
					t->p = v; t->d = d + l;
					t->r = 0x80000000u | cl_seq_live_out(aol, idx, w ^ 1);
					++pending;
				} else {
					if (c + 1 > t->c || (c + 1 == t->c && d + l > t->d)) t->p = v;
					if (c + 1 > t->c) t->c = c + 1;
					if (d + l < t->d) t->d = d + l;
				}
				if ((t->r & 0x7fffffffu) == 0) return -1; /* asg.c:391 */
				if ((--t->r & 0x7fffffffu) == 0) {
					if ((uint32_t)idx[w]) stk[n_stk++] = w; else ++tips; /* deleted arcs count here too (asg.c:393) */
					--pending;
				}
			}
			if (i < nv || n_stk == 0) goto reset;
			if (!(n_stk > 1 || pending)) break;
		}
		/* asg_bub_backtrack (asg.c:338-357): everything touched goes, the best path comes back */
		for (k = 0; k < n_seen; ++k) sdel[seen[k] >> 1] = 1;
		for (k = 0; k < n_walk; ++k) {
			const uint32_t e = walked[k];
			aol[e] |= CL_ADEL;
			cl_seq_arc_set(av, aol, idx, av[e] ^ 1, au[e] ^ 1, 1);
		}
		{
			uint32_t v = stk[0];
			do {
				const uint32_t u = info[v].p;
				sdel[v >> 1] = 0;
				cl_seq_arc_set(av, aol, idx, u, v, 0);
				cl_seq_arc_set(av, aol, idx, v ^ 1, u ^ 1, 0);
				v = u;
			} while (v != v0);
		}
		popped = 1;
reset:
		for (k = 0; k < n_seen; ++k) { cl_seqinfo_t *t = &info[seen[k]]; t->r = t->c = t->d = 0; }
		if (popped) { ++*n_pop; *n_tips += tips; }
	}
	return 0;
}

#endif
