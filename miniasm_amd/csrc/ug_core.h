/* ug_core.h -- per-vertex steps of the device-side unitig construction (csrc/ug.hip; reference asm.c:121-210), written once
 * for the kernels and for the host test harness (tests/clean_host.cpp).  See ug.hip for the method. */
#ifndef UG_CORE_H
#define UG_CORE_H

#include <stdint.h>

#if defined(__HIPCC__)
#define UG_HD __host__ __device__ __forceinline__
#define UG_MIN_U32(p, x) atomicMin((p), (x))
#define UG_ADD_U32(p, x) atomicAdd((p), (x))
#else
#define UG_HD static inline
#define UG_MIN_U32(p, x) do { if ((x) < *(p)) *(p) = (x); } while (0)
#define UG_ADD_U32(p, x) (*(p) += (x))
#endif

#define UG_NONE 0xffffffffu      /* no link: a chain head (prv) / a chain tail (nxt) */
#define UG_OUT  0xfffffffeu      /* prv of a vertex that belongs to no unitig */
#define UG_ADEL 0x80000000u

typedef struct {
	/* the string graph */
	const uint32_t *au, *av, *alen, *aol;
	const unsigned long long *idx;
	const uint8_t *sdel;
	const uint32_t *slen;
	uint32_t n_vtx;
	/* per vertex */
	uint32_t *nxt, *prv, *wt, *cm, *tail, *uid, *flag, *pos;
	uint8_t *circ;
	int32_t *mark;
	/* per unitig */
	uint32_t *u_head, *u_n, *u_len, *u_start, *u_end, *u_off;
	unsigned long long *ua;
} ug_t;

UG_HD uint32_t ug_deg(const ug_t *a, uint32_t v) { return (uint32_t)a->idx[v]; }
UG_HD uint32_t ug_first(const ug_t *a, uint32_t v) { return (uint32_t)(a->idx[v] >> 32); }

/* links in both directions; a vertex is a MEMBER of some unitig if its read is alive and has an arc on either side */
UG_HD void ugk_link(const ug_t *a, uint32_t w)
{
	uint32_t n = UG_NONE, p = UG_NONE;
	const int member = !a->sdel[w >> 1] && (ug_deg(a, w) > 0 || ug_deg(a, w ^ 1) > 0);
	if (member) {
		if (ug_deg(a, w) == 1) { uint32_t x = a->av[ug_first(a, w)]; if (ug_deg(a, x ^ 1) == 1) n = x; }             /* forward step, asm.c:140-142 */
		if (ug_deg(a, w ^ 1) == 1) { uint32_t t = a->av[ug_first(a, w ^ 1)] ^ 1; if (ug_deg(a, t) == 1) p = t; }      /* backward step, asm.c:155-157 */
	}
	a->nxt[w] = n; a->prv[w] = member ? p : UG_OUT;
	/* what w contributes to its unitig's length: the arc to the next read, or -- last read of a linear unitig -- its whole length (asm.c:144-153) */
	a->wt[w] = !member ? 0u : n != UG_NONE ? a->alen[ug_first(a, w)] : (a->slen[w >> 1] & 0x7fffffffu);
}

/* List ranking state per vertex: ptr (where the stretch covered so far begins), and over that stretch: mn = the smallest vertex
 * that has an arc (the reference discovers a unitig there), dist = links, ws = length contributions. */
typedef struct { uint32_t *ptr, *mn, *dist, *ws; } ug_rank_t;

UG_HD void ugk_jump_init(const ug_t *a, uint32_t w, ug_rank_t r)
{
	const uint32_t p = a->prv[w];
	const int linked = p < UG_OUT;
	r.ptr[w] = linked ? p : w;
	r.mn[w] = ug_deg(a, w) > 0 && p != UG_OUT ? w : UG_NONE;
	r.dist[w] = linked ? 1u : 0u;
	r.ws[w] = linked ? a->wt[w] : 0u; /* like dist, the sum leaves the head out (it is re-added every round once the pointer rests there): the head's share is added at the end */
}

/* one round of pointer jumping: the pointer doubles its reach; minimum, distance and length over the skipped stretch are folded in */
UG_HD void ugk_jump(uint32_t w, ug_rank_t i, ug_rank_t o)
{
	const uint32_t p = i.ptr[w];
	const uint32_t x = i.mn[w], y = i.mn[p];
	o.ptr[w] = i.ptr[p];
	o.mn[w] = x < y ? x : y;
	o.dist[w] = i.dist[w] + (p != w ? i.dist[p] : 0u);
	o.ws[w] = i.ws[w] + (p != w ? i.ws[p] : 0u);
}

/* a member whose chain has no head sits on a cycle; the cycle is cut in front of its smallest vertex (where the reference's
 * sweep enters it).  is_head[] is a snapshot of "prv == NONE" taken before this step (the step rewrites prv). */
UG_HD void ugk_cut(const ug_t *a, uint32_t w, const uint32_t *ptr, const uint32_t *mn, const uint8_t *is_head)
{
	const uint32_t p = a->prv[w];
	a->circ[w] = 0;
	if (p >= UG_OUT) return;              /* a head, or not a member */
	if (is_head[ptr[w]]) return;          /* the chain has a head: linear */
	if (mn[w] == w) { a->circ[w] = 1; a->prv[w] = UG_NONE; a->nxt[p] = UG_NONE; } /* (wt keeps the arc length of the closing link: a circular unitig has no last read) */
}

/* per chain (keyed by its head): its last vertex; the ranking state of the last vertex covers the whole chain, so the chain's
 * discovery vertex is simply its mn (one writer per chain: no atomics, however long the unitig) */
UG_HD void ugk_chain(const ug_t *a, uint32_t w, ug_rank_t r)
{
	uint32_t h;
	if (a->prv[w] == UG_OUT || a->nxt[w] != UG_NONE) return;
	h = r.ptr[w];
	a->tail[h] = w;
	a->cm[h] = r.mn[w];
}

/* The orientation the reference emits: it discovers a unitig at its smallest vertex that has an arc; that vertex lies in one of
 * the two complementary chains.  The twin chain is the one that holds the complement of this chain's last vertex. */
UG_HD int ug_emitted(const ug_t *a, uint32_t h, const uint32_t *ptr)
{
	uint32_t x, y;
	if (a->prv[h] != UG_NONE) return 0;   /* heads only */
	x = a->cm[h]; y = a->cm[ptr[a->tail[h] ^ 1]];
	return x != UG_NONE && x < y;
}
UG_HD void ugk_pick(const ug_t *a, uint32_t h, const uint32_t *ptr)
{
	if (ug_emitted(a, h, ptr)) a->flag[a->cm[h]] = 1;
}

/* unitig records in discovery order: number = rank of the discovery vertex among the flagged vertices (pos = scan of flag) */
UG_HD void ugk_units(const ug_t *a, uint32_t h, ug_rank_t r)
{
	const uint32_t *ptr = r.ptr;
	uint32_t k, t;
	a->uid[h] = UG_NONE;
	if (!ug_emitted(a, h, ptr)) return;
	k = a->pos[a->cm[h]]; t = a->tail[h];
	a->uid[h] = k;
	a->u_head[k] = h; a->u_n[k] = r.dist[t] + 1; a->u_len[k] = r.ws[t] + a->wt[h];
	a->u_start[k] = a->circ[h] ? UG_NONE : h; a->u_end[k] = a->circ[h] ? UG_NONE : (t ^ 1);
}

/* members: unitig k = [u_off[k], u_off[k] + u_n[k]) of ua, element = vertex << 32 | length to the next read; the last read of a
 * linear unitig contributes its whole length (asm.c:144-153) */
UG_HD void ugk_fill(const ug_t *a, uint32_t w, const uint32_t *ptr, const uint32_t *dist)
{
	uint32_t h, k, l;
	if (a->prv[w] == UG_OUT) return;
	h = ptr[w]; k = a->uid[h];
	if (k == UG_NONE) return;
	(void)h;
	l = a->wt[w];
	a->ua[a->u_off[k] + dist[w]] = (unsigned long long)w << 32 | l;
}

/* ---- graphs whose links are not mirror images of each other (asymmetric input: `-b -S 5 -p ug`, hand-made graphs through the
 * per-symbol ma_ug_gen) ------------------------------------------------------------------------------------------------------
 * Everything above rests on one property of a symmetric graph: w -> x is a link exactly when x^1 -> w^1 is, so the forward walk
 * (asm.c:139-151) and the backward walk (asm.c:160-171, which follows the COMPLEMENT strand's arcs and never looks at where w's own
 * arc goes) trace the same chain, every unitig exists as two complementary chains and nobody else marks their vertices.  Without it
 * the reference's result is a function of its sweep order: walks run through vertices other unitigs have marked, unitigs overlap,
 * a vertex can start a unitig only if no earlier walk touched it.  ugk_link_bad finds out (one test per link); if any link fails it,
 * the sweep itself runs on the device, in vertex order, on one lane (ug_seq_unitig per start vertex): slow and exact. */
UG_HD int ugk_link_bad(const ug_t *a, uint32_t w)
{
	const uint32_t n = a->nxt[w], p = a->prv[w];
	int bad = 0;
	if (n != UG_NONE) bad |= a->sdel[n >> 1] || n == (w ^ 1) || a->av[ug_first(a, n ^ 1)] != (w ^ 1);
	if (p < UG_OUT) bad |= a->sdel[p >> 1] || a->av[ug_first(a, p)] != w;
	return bad;
}

typedef struct {
	uint8_t *seen;                    /* the reference's mark[] of its first loop (asm.c:131) */
	unsigned long long cap, n_mem;    /* room in ua / members so far (keeps counting past cap: the caller grows ua and repeats) */
	uint32_t n_utg, err;              /* err: a walk that cannot end -- the reference does not return on such a graph */
} ug_seq_t;

/* one iteration of asm.c:133-178 for a start vertex v the caller found alive, with an arc and unmarked */
UG_HD void ug_seq_unitig(const ug_t *a, ug_seq_t *s, uint32_t v)
{
	const uint32_t lim = a->n_vtx;    /* a walk is a function iteration over the vertex set: more steps than vertices = it runs in a cycle it never leaves */
	const unsigned long long off = s->n_mem;
	uint32_t w, x, l, start = v, end = v ^ 1, len = 0, nf = 0, nb = 0, k, n;
	int circ;
	s->seen[v] = 1;
	for (w = v;;) {                   /* forward: marks and the count; the members are written once their place is known */
		if (ug_deg(a, w) != 1) break;
		x = a->av[ug_first(a, w)];
		if (ug_deg(a, x ^ 1) != 1) break;
		s->seen[x] = s->seen[w ^ 1] = 1;
		len += a->alen[ug_first(a, w)];
		end = x ^ 1; ++nf; w = x;
		if (x == v) break;
		if (nf > lim) { s->err = 1; return; }
	}
	circ = start == (end ^ 1) && nf != 0;
	if (!circ) {                      /* backward: written front to back behind `off`, turned round afterwards (kdq_unshift) */
		for (x = v;;) {
			if (ug_deg(a, x ^ 1) != 1) break;
			w = a->av[ug_first(a, x ^ 1)] ^ 1;
			if (ug_deg(a, w) != 1) break;
			s->seen[x] = s->seen[w ^ 1] = 1;
			l = a->alen[ug_first(a, w)];
			if (off + nb < s->cap) a->ua[off + nb] = (unsigned long long)w << 32 | l;
			start = w; len += l; ++nb; x = w;
			if (nb > lim) { s->err = 1; return; }
		}
		if (off + nb <= s->cap)
			for (k = 0; k < nb / 2; ++k) { const unsigned long long t = a->ua[off + k]; a->ua[off + k] = a->ua[off + nb - 1 - k]; a->ua[off + nb - 1 - k] = t; }
	}
	for (w = v, k = 0; k < nf; ++k) { /* forward again */
		const uint32_t e = ug_first(a, w);
		if (off + nb + k < s->cap) a->ua[off + nb + k] = (unsigned long long)w << 32 | a->alen[e];
		w = a->av[e];
	}
	n = nb + nf;
	if (!circ) {                      /* asm.c:152-155: the last read contributes its whole length */
		l = a->slen[end >> 1] & 0x7fffffffu;
		if (off + n < s->cap) a->ua[off + n] = (unsigned long long)(end ^ 1) << 32 | l;
		len += l; ++n;
		s->seen[start] = s->seen[end] = 1;
	}
	k = s->n_utg++;                   /* at most one unitig per vertex: the per-unitig arrays always have room */
	a->u_n[k] = n; a->u_len[k] = len; a->u_off[k] = (uint32_t)off;
	a->u_start[k] = circ ? UG_NONE : start; a->u_end[k] = circ ? UG_NONE : end;
	s->n_mem = off + n;
}

UG_HD void ugk_mark(const ug_t *a, uint32_t k) /* asm.c:180-184 */
{
	if (a->u_start[k] == UG_NONE) return;
	a->mark[a->u_start[k]] = (int32_t)(k << 1 | 0); a->mark[a->u_end[k]] = (int32_t)(k << 1 | 1);
}

UG_HD uint32_t ugk_arc_keep(const ug_t *a, size_t e) /* asm.c:187-190 */
{
	return !(a->aol[e] & UG_ADEL) && a->mark[a->au[e] ^ 1] >= 0 && a->mark[a->av[e]] >= 0;
}

/* the unitig arc of string-graph arc e as the four words of an asg_arc_t {len, u, v, ol} (asm.c:191-198) */
UG_HD void ugk_arc_emit(const ug_t *a, size_t e, uint32_t out[4])
{
	const uint32_t u = (uint32_t)a->mark[a->au[e] ^ 1] ^ 1, ol = a->aol[e] & 0x7fffffffu;
	int32_t l = (int32_t)(a->u_len[u >> 1] - ol);
	if (l < 0) l = 1;
	out[0] = (uint32_t)l; out[1] = u; out[2] = (uint32_t)a->mark[a->av[e]]; out[3] = ol;
}

#endif
