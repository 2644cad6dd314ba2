/* core_host.c -- test shim: the per-hit device arithmetic (miniasm_amd/csrc/ma_core.h) compiled for the
 * host, exported with plain C symbols so the CPU tests can compare it with the reference's ma_hit2arc /
 * ma_hit_cut on millions of random inputs without a GPU. */
#include "ma_core.h"

int core_hit2arc(uint32_t qid, uint32_t qs, uint32_t qe, uint32_t tn, uint32_t ts, uint32_t te, int rev, int ql, int tl,
                 int max_hang, float int_frac, int min_ovlp, uint32_t *out4)
{
	mc_arc_t a = {0, 0, 0, 0};
	int r = mc_hit2arc(qid, qs, qe, tn, ts, te, rev, ql, tl, max_hang, int_frac, min_ovlp, &a);
	out4[0] = a.u, out4[1] = a.v, out4[2] = a.len, out4[3] = a.ol;
	return r;
}

int core_cut(uint32_t *c4, int rev, int32_t rq_s, uint32_t rq_e, int32_t rt_s, uint32_t rt_e, int min_span)
{
	return mc_cut(&c4[0], &c4[1], &c4[2], &c4[3], rev, rq_s, rq_e, rt_s, rt_e, min_span);
}

int core_sub_ok(uint32_t qid, uint32_t qs, uint32_t qe, uint32_t tn, int32_t ml, int32_t bl, float min_iden, int end_clip, uint32_t *ev2)
{
	return mc_sub_ok(qid, qs, qe, tn, ml, bl, min_iden, end_clip, &ev2[0], &ev2[1]);
}
