/* ma_core.h -- per-hit arithmetic of the hot path, written once for device (HIP) and host (unit tests).
 *
 * These are the places where the reference mixes int, uint32_t, 31-bit bit-fields (which promote to
 * SIGNED int) and float; the result is only bit-identical if every comparison keeps the signedness
 * and width C gives it there (SURVEY.md section 8a "type notes").  The functions below take SoA
 * scalars and spell every conversion out.  Compile with -ffp-contract=off.
 *
 *   mc_hit2arc : reference miniasm.h:86-104 (ma_hit2arc)
 *   mc_cut     : reference hit.c:168-188   (body of ma_hit_cut)
 *   mc_sub_ok  : reference hit.c:125-127   (which hits feed the coverage sweep of ma_hit_sub)
 */
#ifndef MA_CORE_H
#define MA_CORE_H

#include <stdint.h>

#if defined(__HIPCC__)
#define MA_HD __host__ __device__ __forceinline__
#else
#define MA_HD static inline
#endif

#define MC_HT_INT        (-1)
#define MC_HT_QCONT      (-2)
#define MC_HT_TCONT      (-3)
#define MC_HT_SHORT_OVLP (-4)

typedef struct {
	uint32_t u, v;   /* vertex ids: read<<1 | end */
	uint32_t len;    /* length from u to v */
	uint32_t ol;     /* 31-bit overlap length */
} mc_arc_t;

/* Classify one hit.  qs_u is the low word of qns; ql/tl are the (clipped) read lengths as C ints.
 * Returns MC_HT_* (<0) or (int)len with *a filled in. */
MA_HD int mc_hit2arc(uint32_t qid, uint32_t qs_u, uint32_t qe, uint32_t tn, uint32_t ts, uint32_t te, int rev,
                     int ql, int tl, int max_hang, float int_frac, int min_ovlp, mc_arc_t *a)
{
	int32_t tl5, tl3, ext5, ext3, qs = (int32_t)qs_u;
	uint32_t u, v, l, qrem, span, x;
	if (rev) tl5 = (int32_t)((uint32_t)tl - te), tl3 = (int32_t)ts;   /* int - uint32 -> uint32 -> int32 */
	else tl5 = (int32_t)ts, tl3 = (int32_t)((uint32_t)tl - te);
	ext5 = qs < tl5 ? qs : tl5;                                       /* signed compare */
	qrem = (uint32_t)ql - qe;                                         /* "ql - h->qe" is uint32 */
	ext3 = (int32_t)(qrem < (uint32_t)tl3 ? qrem : (uint32_t)tl3);    /* unsigned compare, uint32 result */
	span = qe - (uint32_t)qs;                                         /* "h->qe - qs" is uint32 */
	x = span + (uint32_t)ext5 + (uint32_t)ext3;
	if (ext5 > max_hang || ext3 > max_hang || (float)span < (float)x * int_frac)
		return MC_HT_INT;
	if (qs <= tl5 && qrem <= (uint32_t)tl3) return MC_HT_QCONT;
	else if (qs >= tl5 && qrem >= (uint32_t)tl3) return MC_HT_TCONT;
	else if (qs > tl5) u = 0, v = !!rev, l = (uint32_t)(qs - tl5);
	else u = 1, v = !rev, l = qrem - (uint32_t)tl3;
	if (x < (uint32_t)min_ovlp || te - ts + (uint32_t)ext5 + (uint32_t)ext3 < (uint32_t)min_ovlp)
		return MC_HT_SHORT_OVLP;
	u |= qid << 1, v |= tn << 1;
	a->u = u, a->v = v, a->len = l, a->ol = ((uint32_t)ql - l) & 0x7fffffffu;
	return (int)l;
}

/* Clip one hit to the kept intervals [qsub_s,qsub_e) of the query and [tsub_s,tsub_e) of the target and
 * re-base its coordinates.  *_s are the 31-bit fields (promote to int), *_e plain uint32.
 * Returns 1 if the clipped hit still spans min_span on both reads (coordinates written back). */
MA_HD int mc_cut(uint32_t *pqs, uint32_t *pqe, uint32_t *pts, uint32_t *pte, int rev,
                 int32_t rq_s, uint32_t rq_e, int32_t rt_s, uint32_t rt_e, int min_span)
{
	uint32_t oqs = *pqs, oqe = *pqe, ots = *pts, ote = *pte;
	int32_t qs, qe, ts, te;
	if (rev) {
		qs = (int32_t)(ote < rt_e ? oqs : oqs + (ote - rt_e));
		qe = (int32_t)(ots > (uint32_t)rt_s ? oqe : oqe - ((uint32_t)rt_s - ots));
		ts = (int32_t)(oqe < rq_e ? ots : ots + (oqe - rq_e));
		te = (int32_t)(oqs > (uint32_t)rq_s ? ote : ote - ((uint32_t)rq_s - oqs));
	} else {
		qs = (int32_t)(ots > (uint32_t)rt_s ? oqs : oqs + ((uint32_t)rt_s - ots));
		qe = (int32_t)(ote < rt_e ? oqe : oqe - (ote - rt_e));
		ts = (int32_t)(oqs > (uint32_t)rq_s ? ots : ots + ((uint32_t)rq_s - oqs));
		te = (int32_t)(oqe < rq_e ? ote : ote - (oqe - rq_e));
	}
	qs = (qs > rq_s ? qs : rq_s) - rq_s;                                              /* signed max */
	qe = (int32_t)(((uint32_t)qe < rq_e ? (uint32_t)qe : rq_e) - (uint32_t)rq_s);     /* unsigned min */
	ts = (ts > rt_s ? ts : rt_s) - rt_s;
	te = (int32_t)(((uint32_t)te < rt_e ? (uint32_t)te : rt_e) - (uint32_t)rt_s);
	if (qe - qs >= min_span && te - ts >= min_span) {
		*pqs = (uint32_t)qs, *pqe = (uint32_t)qe, *pts = (uint32_t)ts, *pte = (uint32_t)te;
		return 1;
	}
	return 0;
}

/* Does this hit contribute a [start,end) pair to the coverage sweep?  ml/bl are the 31-bit fields. */
MA_HD int mc_sub_ok(uint32_t qid, uint32_t qs, uint32_t qe, uint32_t tn, int32_t ml, int32_t bl,
                    float min_iden, int end_clip, uint32_t *ev_s, uint32_t *ev_e)
{
	uint32_t s, e;
	if (tn == qid || (float)ml < (float)bl * min_iden) return 0;
	s = qs + (uint32_t)end_clip, e = qe - (uint32_t)end_clip;
	if (e > s) { *ev_s = s << 1, *ev_e = e << 1 | 1; return 1; }
	return 0;
}

/* ma_hit_no_cont's per-line verdict (hit.c:52-64; the -R pre-filter): 0 nothing, 1 the TARGET read is clearly contained, 2 the QUERY
 * read is.  The columns are the reader's uint32_t fields; l5 / l3 are ints built from uint32 arithmetic as in the reference. */
MA_HD int mc_no_cont(uint32_t ql, uint32_t qs, uint32_t qe, uint32_t tl, uint32_t ts, uint32_t te, int rev, int max_hang, float int_frac)
{
	const int l5 = (int)(rev ? tl - te : ts), l3 = (int)(rev ? ts : tl - te);
	if (ql >> 1 > tl) {
		if (l5 > max_hang >> 2 || l3 > max_hang >> 2 || (float)(te - ts) < (float)tl * int_frac) return 0;
		if ((int)qs - l5 > max_hang << 1 && (int)(ql - qe) - l3 > max_hang << 1) return 1;
	} else if (ql < tl >> 1) {
		if (qs > (uint32_t)(max_hang >> 2) || ql - qe > (uint32_t)(max_hang >> 2) || (float)(qe - qs) < (float)ql * int_frac) return 0;
		if (l5 - (int)qs > max_hang << 1 && l3 - (int)(ql - qe) > max_hang << 1) return 2;
	}
	return 0;
}

#endif
