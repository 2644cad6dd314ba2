/* ma_oracle.c -- CPU ORACLE, TEST INFRASTRUCTURE ONLY (see ma_oracle.h for who may use it and how it is pinned).
 *
 * Straight-line C that follows the reference function by function; each block names the reference lines it
 * restates.  Where the reference relies on C's implicit conversions (31-bit bit-fields promote to signed
 * int, uint32_t stays unsigned, float vs double) the same declarations are used so the compiler applies the
 * same rules.  Sorting uses qsort with the input position as the last key: a total order (the reference's
 * in-place radix sort leaves equal keys in a data-dependent order; inputs used for parity are tie-free, or
 * compared after normalisation -- DESIGN.md "tie order").
 */
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include "ma_oracle.h"

#define HT_INT   (-1)
#define HT_QCONT (-2)
#define HT_TCONT (-3)
#define HT_SHORT (-4)

/* ------------------------------------------------------------------ sort (hit.c:12-22; total order) */
typedef struct { uint64_t key; size_t pos; } keypos_t;
static int cmp_keypos(const void *a, const void *b)
{
	const keypos_t *x = (const keypos_t*)a, *y = (const keypos_t*)b;
	if (x->key != y->key) return x->key < y->key ? -1 : 1;
	return x->pos < y->pos ? -1 : x->pos > y->pos;
}

void orc_hit_sort(size_t n, orc_hit_t *a)
{
	keypos_t *k = (keypos_t*)malloc((n ? n : 1) * sizeof(keypos_t));
	orc_hit_t *t = (orc_hit_t*)malloc((n ? n : 1) * sizeof(orc_hit_t));
	size_t i;
	for (i = 0; i < n; ++i) k[i].key = a[i].qns, k[i].pos = i;
	qsort(k, n, sizeof(keypos_t), cmp_keypos);
	for (i = 0; i < n; ++i) t[i] = a[k[i].pos];
	memcpy(a, t, n * sizeof(orc_hit_t));
	free(k); free(t);
}

/* ------------------------------------------------------------------ miniasm.h:86-104 */
int orc_hit2arc(const orc_hit_t *h, int ql, int tl, int max_hang, float int_frac, int min_ovlp, orc_arc_t *p)
{
	int32_t tl5, tl3, ext5, ext3, qs = (int32_t)h->qns;
	uint32_t u, v, l;
	if (h->rev) tl5 = tl - h->te, tl3 = h->ts;          /* overhangs of the target, in query orientation */
	else tl5 = h->ts, tl3 = tl - h->te;
	ext5 = qs < tl5 ? qs : tl5;
	ext3 = ql - h->qe < tl3 ? ql - h->qe : tl3;
	if (ext5 > max_hang || ext3 > max_hang || h->qe - qs < (h->qe - qs + ext5 + ext3) * int_frac)
		return HT_INT;
	if (qs <= tl5 && ql - h->qe <= tl3) return HT_QCONT;
	else if (qs >= tl5 && ql - h->qe >= tl3) return HT_TCONT;
	else if (qs > tl5) u = 0, v = !!h->rev, l = qs - tl5;
	else u = 1, v = !h->rev, l = (ql - h->qe) - tl3;
	if (h->qe - qs + ext5 + ext3 < min_ovlp || h->te - h->ts + ext5 + ext3 < min_ovlp) return HT_SHORT;
	u |= h->qns >> 32 << 1, v |= h->tn << 1;
	p->ul = (uint64_t)u << 32 | l, p->v = v, p->ol = ql - l, p->del = 0;
	return l;
}

/* ------------------------------------------------------------------ hit.c:109-160 */
static int cmp_u32(const void *a, const void *b)
{
	uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
	return x < y ? -1 : x > y;
}

size_t orc_hit_sub(int min_dp, float min_iden, int end_clip, size_t n, const orc_hit_t *a, size_t n_sub, orc_sub_t *sub)
{
	size_t i, j, last, n_remained = 0, m_ev = 0;
	uint32_t *ev = 0;
	for (i = 1, last = 0; i <= n; ++i) {
		if (i != n && a[i].qns >> 32 == a[i-1].qns >> 32) continue;
		{ /* hits [last,i) share one query */
			int qid = a[i-1].qns >> 32, dp = 0;
			size_t n_ev = 0, start = 0;
			orc_sub_t best;
			if (2 * (i - last) > m_ev) { m_ev = 2 * (i - last); ev = (uint32_t*)realloc(ev, m_ev * 4); }
			for (j = last; j < i; ++j) {
				uint32_t qs, qe;
				if (a[j].tn == qid || a[j].ml < a[j].bl * min_iden) continue; /* self hit or low identity (hit.c:125) */
				qs = (uint32_t)a[j].qns + end_clip, qe = a[j].qe - end_clip;
				if (qe > qs) ev[n_ev++] = qs << 1, ev[n_ev++] = qe << 1 | 1;
			}
			qsort(ev, n_ev, 4, cmp_u32);
			best.s = best.e = best.del = 0;
			for (j = 0; j < n_ev; ++j) { /* depth sweep; first longest run at depth >= min_dp (hit.c:134-145) */
				int old_dp = dp;
				if (ev[j] & 1) --dp; else ++dp;
				if (old_dp < min_dp && dp >= min_dp) start = ev[j] >> 1;
				else if (old_dp >= min_dp && dp < min_dp) {
					int len = (ev[j] >> 1) - start;
					if (len > best.e - best.s) best.s = start, best.e = ev[j] >> 1;
				}
			}
			if (best.e - best.s > 0) {
				assert((size_t)qid < n_sub);
				sub[qid].s = best.s - end_clip, sub[qid].e = best.e + end_clip, sub[qid].del = 0;
				++n_remained;
			} else sub[qid].del = 1;
			last = i;
		}
	}
	free(ev);
	return n_remained;
}

/* ------------------------------------------------------------------ hit.c:162-193 */
size_t orc_hit_cut(const orc_sub_t *reg, int min_span, size_t n, orc_hit_t *a)
{
	size_t i, m = 0;
	for (i = 0; i < n; ++i) {
		orc_hit_t *p = &a[i];
		const orc_sub_t *rq = &reg[p->qns >> 32], *rt = &reg[p->tn];
		int qs, qe, ts, te;
		if (rq->del || rt->del) continue;
		if (p->rev) { /* a clip on one read moves the far end of the hit on the other read */
			qs = p->te < rt->e ? (uint32_t)p->qns : (uint32_t)p->qns + (p->te - rt->e);
			qe = p->ts > rt->s ? p->qe : p->qe - (rt->s - p->ts);
			ts = p->qe < rq->e ? p->ts : p->ts + (p->qe - rq->e);
			te = (uint32_t)p->qns > rq->s ? p->te : p->te - (rq->s - (uint32_t)p->qns);
		} else {
			qs = p->ts > rt->s ? (uint32_t)p->qns : (uint32_t)p->qns + (rt->s - p->ts);
			qe = p->te < rt->e ? p->qe : p->qe - (p->te - rt->e);
			ts = (uint32_t)p->qns > rq->s ? p->ts : p->ts + (rq->s - (uint32_t)p->qns);
			te = p->qe < rq->e ? p->te : p->te - (p->qe - rq->e);
		}
		qs = (qs > rq->s ? qs : rq->s) - rq->s;
		qe = (qe < rq->e ? qe : rq->e) - rq->s;
		ts = (ts > rt->s ? ts : rt->s) - rt->s;
		te = (te < rt->e ? te : rt->e) - rt->s;
		if (qe - qs >= min_span && te - ts >= min_span) {
			p->qns = p->qns >> 32 << 32 | qs, p->qe = qe, p->ts = ts, p->te = te;
			a[m++] = *p;
		}
	}
	return m;
}

/* ------------------------------------------------------------------ hit.c:195-216 */
size_t orc_hit_flt(const orc_sub_t *sub, int max_hang, int min_ovlp, size_t n, orc_hit_t *a, float *cov)
{
	size_t i, m = 0;
	uint64_t tot_dp = 0, tot_len = 0;
	orc_arc_t t;
	for (i = 0; i < n; ++i) {
		orc_hit_t *h = &a[i];
		const orc_sub_t *sq = &sub[h->qns >> 32], *st = &sub[h->tn];
		int r;
		if (sq->del || st->del) continue;
		r = orc_hit2arc(h, sq->e - sq->s, st->e - st->s, max_hang, .5, min_ovlp, &t);
		if (r >= 0 || r == HT_QCONT || r == HT_TCONT)
			a[m++] = *h, tot_dp += r >= 0 ? r : r == HT_QCONT ? sq->e - sq->s : st->e - st->s;
	}
	for (i = 1; i <= m; ++i)
		if (i == m || a[i].qns >> 32 != a[i-1].qns >> 32)
			tot_len += sub[a[i-1].qns >> 32].e - sub[a[i-1].qns >> 32].s;
	*cov = (double)tot_dp / tot_len;
	return m;
}

/* ------------------------------------------------------------------ hit.c:218-223 */
void orc_sub_merge(size_t n_sub, orc_sub_t *a, const orc_sub_t *b)
{
	size_t i;
	for (i = 0; i < n_sub; ++i) a[i].e = a[i].s + b[i].e, a[i].s += b[i].s;
}

/* ------------------------------------------------------------------ hit.c:225-256 (+ :24-36, sdict.c:69-86) */
/* first loop of ma_hit_contained + the marking loop of ma_hit_mark_unused, as pure flag computation (no renumbering):
 * r_cont[x] = read x is contained by the final thresholds, r_used[x] = some hit touches x */
void orc_contained_flags(const orc_opt_t *opt, const orc_sub_t *sub, size_t n, const orc_hit_t *a, uint8_t *r_cont, uint8_t *r_used)
{
	size_t i;
	orc_arc_t t;
	for (i = 0; i < n; ++i) {
		const orc_hit_t *h = &a[i];
		const orc_sub_t *sq = &sub[h->qns >> 32], *st = &sub[h->tn];
		int r = orc_hit2arc(h, sq->e - sq->s, st->e - st->s, opt->max_hang, opt->int_frac, opt->min_ovlp, &t);
		if (r == HT_QCONT) r_cont[h->qns >> 32] = 1;
		else if (r == HT_TCONT) r_cont[h->tn] = 1;
		r_used[h->qns >> 32] = r_used[h->tn] = 1;
	}
}

size_t orc_hit_contained(const orc_opt_t *opt, uint32_t n_seq, uint8_t *seq_del, orc_sub_t *sub, size_t n, orc_hit_t *a, int32_t *map, uint32_t *n_seq_new)
{
	size_t i, m = 0;
	uint32_t j = 0;
	uint8_t *used = (uint8_t*)calloc(n_seq ? n_seq : 1, 1), *cont = (uint8_t*)calloc(n_seq ? n_seq : 1, 1);
	orc_contained_flags(opt, sub, n, a, cont, used);
	for (i = 0; i < n_seq; ++i) if (cont[i]) sub[i].del = 1;                    /* hit.c:234-235 write into sub */
	for (i = 0; i < n_seq; ++i) if (sub[i].del) seq_del[i] = 1;                 /* hit.c:237-238 */
	for (i = 0; i < n_seq; ++i) if (!used[i]) seq_del[i] = 1;                   /* hit.c:24-36: reads no hit touches go too */
	for (i = 0; i < n_seq; ++i) map[i] = seq_del[i] ? -1 : (int32_t)j++;        /* sdict.c:75-81 */
	for (i = 0; i < n_seq; ++i) if (map[i] >= 0) sub[map[i]] = sub[i];
	for (i = 0; i < n; ++i) {
		int32_t qn = map[a[i].qns >> 32], tn = map[a[i].tn];
		if (qn >= 0 && tn >= 0) {
			a[i].qns = (uint64_t)qn << 32 | (uint32_t)a[i].qns, a[i].tn = tn;
			a[m++] = a[i];
		}
	}
	free(used); free(cont);
	*n_seq_new = j;
	return m;
}

/* ------------------------------------------------------------------ asm.c:9-39 (+ asg.c:57-80) */
/* asm.c:14-35 without the cleanup: seq arrays, the pushed arcs in push order, seq.del side effects; returns #pushed */
size_t orc_sg_candidates(const orc_opt_t *opt, uint32_t n_seq, const orc_sub_t *sub, const uint32_t *len_in, const uint8_t *del_in,
                         size_t n, const orc_hit_t *a, orc_arc_t *arcs, uint32_t *seq_len, uint8_t *seq_del)
{
	size_t i, n_arc = 0;
	for (i = 0; i < n_seq; ++i) {
		if (sub) seq_len[i] = (sub[i].e - sub[i].s) & 0x7fffffffu, seq_del[i] = sub[i].del || (del_in && del_in[i]);
		else seq_len[i] = len_in[i] & 0x7fffffffu, seq_del[i] = del_in ? del_in[i] : 0;
	}
	for (i = 0; i < n; ++i) {
		const orc_hit_t *h = &a[i];
		uint32_t qn = h->qns >> 32;
		int ql = sub ? sub[qn].e - sub[qn].s : len_in[qn];
		int tl = sub ? sub[h->tn].e - sub[h->tn].s : len_in[h->tn];
		orc_arc_t t;
		int r = orc_hit2arc(h, ql, tl, opt->max_hang, opt->int_frac, opt->min_ovlp, &t);
		if (r >= 0) {
			if (qn == h->tn) { /* self overlap: only the palindromic artefact matters (asm.c:27-31) */
				if ((uint32_t)h->qns == h->ts && h->qe == h->te && h->rev) seq_del[qn] = 1;
				continue;
			}
			arcs[n_arc++] = t;
		} else if (r == HT_QCONT) seq_del[qn] = 1;
	}
	return n_arc;
}

/* asg_cleanup of ma_sg_gen (asm.c:36): arc_rm, then sort by ul with ties in push order */
size_t orc_sg_finish(size_t n_arc, orc_arc_t *arcs, const uint8_t *seq_del)
{
	size_t m;
	keypos_t *k;
	orc_arc_t *tmp;
	n_arc = orc_arc_rm(n_arc, arcs, seq_del);
	k = (keypos_t*)malloc((n_arc ? n_arc : 1) * sizeof(keypos_t));
	tmp = (orc_arc_t*)malloc((n_arc ? n_arc : 1) * sizeof(orc_arc_t));
	for (m = 0; m < n_arc; ++m) k[m].key = arcs[m].ul, k[m].pos = m;
	qsort(k, n_arc, sizeof(keypos_t), cmp_keypos);
	for (m = 0; m < n_arc; ++m) tmp[m] = arcs[k[m].pos];
	memcpy(arcs, tmp, n_arc * sizeof(orc_arc_t));
	free(k); free(tmp);
	return n_arc;
}

size_t orc_sg_gen(const orc_opt_t *opt, uint32_t n_seq, const orc_sub_t *sub, const uint32_t *len_in, const uint8_t *del_in,
                  size_t n, const orc_hit_t *a, orc_arc_t *arcs, uint32_t *seq_len, uint8_t *seq_del)
{
	size_t n_arc = orc_sg_candidates(opt, n_seq, sub, len_in, del_in, n, a, arcs, seq_len, seq_del);
	return orc_sg_finish(n_arc, arcs, seq_del);
}

/* ------------------------------------------------------------------ asg.c:27-36 */
void orc_arc_index(uint32_t n_seq, size_t n_arc, const orc_arc_t *a, uint64_t *idx)
{
	size_t i, first = 0;
	memset(idx, 0, (size_t)n_seq * 2 * 8);
	for (i = 1; i <= n_arc; ++i)
		if (i == n_arc || a[i-1].ul >> 32 != a[i].ul >> 32)
			idx[a[i-1].ul >> 32] = (uint64_t)first << 32 | (i - first), first = i;
}

/* ------------------------------------------------------------------ asg.c:57-70 */
size_t orc_arc_rm(size_t n_arc, orc_arc_t *a, const uint8_t *seq_del)
{
	size_t e, n = 0;
	for (e = 0; e < n_arc; ++e) {
		uint32_t u = a[e].ul >> 32, v = a[e].v;
		if (!a[e].del && !seq_del[u >> 1] && !seq_del[v >> 1]) a[n++] = a[e];
	}
	return n;
}

#define ARC_N(idx, v) ((uint32_t)(idx)[(v)])
#define ARC_A(a, idx, v) (&(a)[(idx)[(v)] >> 32])

/* ------------------------------------------------------------------ asg.c:148-186 */
uint32_t orc_arc_del_trans(uint32_t n_seq, size_t n_arc, orc_arc_t *a, const uint64_t *idx, const uint8_t *seq_del, int fuzz, uint64_t *n_inner)
{
	return orc_arc_del_trans_range(n_seq, n_arc, a, idx, seq_del, fuzz, 0, n_seq * 2, n_inner);
}

/* the same sweep restricted to the vertices [v_beg, v_end): every vertex only writes its own arcs, so ranges are independent */
uint32_t orc_arc_del_trans_range(uint32_t n_seq, size_t n_arc, orc_arc_t *a, const uint64_t *idx, const uint8_t *seq_del, int fuzz,
                                 uint32_t v_beg, uint32_t v_end, uint64_t *n_inner)
{
	uint32_t v, n_vtx = n_seq * 2, n_reduced = 0;
	uint8_t *mark = (uint8_t*)calloc(n_vtx ? n_vtx : 1, 1);
	uint64_t inner = 0;
	(void)n_arc;
	if (v_end > n_vtx) v_end = n_vtx;
	for (v = v_beg; v < v_end; ++v) {
		uint32_t L, i, nv = ARC_N(idx, v);
		orc_arc_t *av = ARC_A(a, idx, v);
		if (nv == 0) continue;
		if (seq_del[v >> 1]) {
			for (i = 0; i < nv; ++i) av[i].del = 1, ++n_reduced;
			continue;
		}
		for (i = 0; i < nv; ++i) mark[av[i].v] = 1;                 /* every neighbour "in play" */
		L = (uint32_t)av[nv-1].ul + fuzz;                           /* longest arc + fuzz */
		for (i = 0; i < nv; ++i) {
			uint32_t w = av[i].v, j, nw = ARC_N(idx, w);
			const orc_arc_t *aw = ARC_A(a, idx, w);
			if (mark[av[i].v] != 1) continue;                       /* already reduced: do not expand it */
			for (j = 0; j < nw && (uint32_t)aw[j].ul + (uint32_t)av[i].ul <= L; ++j, ++inner)
				if (mark[aw[j].v]) mark[aw[j].v] = 2;
		}
		for (i = 0; i < nv; ++i) {
			if (mark[av[i].v] == 2) av[i].del = 1, ++n_reduced;
			mark[av[i].v] = 0;
		}
	}
	free(mark);
	if (n_inner) *n_inner = inner;
	return n_reduced;
}

/* ------------------------------------------------------------------ asg.c:104-118 */
uint32_t orc_arc_del_multi(uint32_t n_seq, size_t n_arc, orc_arc_t *a, const uint64_t *idx)
{
	uint32_t *cnt, n_vtx = n_seq * 2, n_multi = 0, v;
	(void)n_arc;
	cnt = (uint32_t*)calloc(n_vtx ? n_vtx : 1, 4);
	for (v = 0; v < n_vtx; ++v) {
		orc_arc_t *av = ARC_A(a, idx, v);
		int32_t i, nv = ARC_N(idx, v);
		if (nv < 2) continue;
		for (i = nv - 1; i >= 0; --i) ++cnt[av[i].v];
		for (i = nv - 1; i >= 0; --i)
			if (--cnt[av[i].v] != 0) av[i].del = 1, ++n_multi;
	}
	free(cnt);
	return n_multi;
}

/* ------------------------------------------------------------------ asg.c:124-135 */
uint32_t orc_arc_del_asymm(uint32_t n_seq, size_t n_arc, orc_arc_t *a, const uint64_t *idx)
{
	size_t e;
	uint32_t n_asymm = 0;
	(void)n_seq;
	for (e = 0; e < n_arc; ++e) {
		uint32_t v = a[e].v ^ 1, u = a[e].ul >> 32 ^ 1, i, nv = ARC_N(idx, v);
		const orc_arc_t *av = ARC_A(a, idx, v);
		for (i = 0; i < nv; ++i)
			if (av[i].v == u) break;
		if (i == nv) a[e].del = 1, ++n_asymm;
	}
	return n_asymm;
}

/* ------------------------------------------------------------------ asg.c:83-96 */
uint32_t orc_arc_del_short(uint32_t n_seq, size_t n_arc, orc_arc_t *a, const uint64_t *idx, float drop_ratio)
{
	uint32_t v, n_vtx = n_seq * 2, n_short = 0;
	(void)n_arc;
	for (v = 0; v < n_vtx; ++v) {
		orc_arc_t *av = ARC_A(a, idx, v);
		uint32_t i, thres, nv = ARC_N(idx, v);
		if (nv < 2) continue;
		thres = (uint32_t)(av[0].ol * drop_ratio + .499);
		for (i = nv - 1; i >= 1 && av[i].ol < thres; --i);
		for (i = i + 1; i < nv; ++i) av[i].del = 1, ++n_short;
	}
	return n_short;
}
