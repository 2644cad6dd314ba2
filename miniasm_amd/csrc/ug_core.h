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
	uint32_t *nxt, *prv, *cm, *tail, *uid, *flag, *pos;
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
}

UG_HD void ugk_jump_init(const ug_t *a, uint32_t w, uint32_t *ptr, uint32_t *mn, uint32_t *dist)
{
	const uint32_t p = a->prv[w];
	const int linked = p < UG_OUT;
	ptr[w] = linked ? p : w;
	if (mn) mn[w] = w;
	if (dist) dist[w] = linked ? 1u : 0u;
}

/* one round of pointer jumping: the pointer doubles its reach, the minimum / the distance over the skipped stretch is folded in */
UG_HD void ugk_jump(uint32_t w, const uint32_t *ptr, const uint32_t *mn, const uint32_t *dist, uint32_t *ptr2, uint32_t *mn2, uint32_t *dist2)
{
	const uint32_t p = ptr[w];
	ptr2[w] = ptr[p];
	if (mn) { uint32_t x = mn[w], y = mn[p]; mn2[w] = x < y ? x : y; }
	if (dist) dist2[w] = dist[w] + (p != w ? dist[p] : 0u);
}

/* a member whose chain has no head sits on a cycle; the cycle is cut in front of its smallest vertex (where the reference's
 * sweep enters it).  is_head[] is a snapshot of "prv == NONE" taken before this step (the step rewrites prv). */
UG_HD void ugk_cut(const ug_t *a, uint32_t w, const uint32_t *ptr, const uint32_t *mn, const uint8_t *is_head)
{
	const uint32_t p = a->prv[w];
	a->circ[w] = 0;
	if (p >= UG_OUT) return;              /* a head, or not a member */
	if (is_head[ptr[w]]) return;          /* the chain has a head: linear */
	if (mn[w] == w) { a->circ[w] = 1; a->prv[w] = UG_NONE; a->nxt[p] = UG_NONE; }
}

/* per chain (keyed by its head): last vertex, smallest vertex that has an arc */
UG_HD void ugk_chain(const ug_t *a, uint32_t w, const uint32_t *ptr)
{
	uint32_t h;
	if (a->prv[w] == UG_OUT) return;
	h = ptr[w];
	if (a->nxt[w] == UG_NONE) a->tail[h] = w;
	if (ug_deg(a, w) > 0) UG_MIN_U32(&a->cm[h], w);
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
UG_HD void ugk_units(const ug_t *a, uint32_t h, const uint32_t *ptr, const uint32_t *dist)
{
	uint32_t k, t;
	a->uid[h] = UG_NONE;
	if (!ug_emitted(a, h, ptr)) return;
	k = a->pos[a->cm[h]]; t = a->tail[h];
	a->uid[h] = k;
	a->u_head[k] = h; a->u_n[k] = dist[t] + 1;
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
	l = (a->nxt[w] != UG_NONE || a->circ[h]) ? a->alen[ug_first(a, w)] : (a->slen[w >> 1] & 0x7fffffffu);
	a->ua[a->u_off[k] + dist[w]] = (unsigned long long)w << 32 | l;
	UG_ADD_U32(&a->u_len[k], l);
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
