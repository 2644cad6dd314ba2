/* ma_oracle.h -- CPU ORACLE, TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the data-parallel part of the miniasm hot path (every pass that has a HIP
 * kernel in miniasm_amd/csrc).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product never links it and has no CPU fallback.
 *
 * Pinning: the reference ships no tests or golden vectors, so this oracle is pinned
 * against the reference itself: tests/test_oracle_vs_ref.py calls the unmodified reference functions in
 * oracle/_ref/libminiasm_ref.so on the same inputs, and tests/golden/ holds stage dumps produced by
 * oracle/_ref/miniasm_ref (script: tests/golden/make_golden.py).  The sequential graph cleaners, unitig
 * construction and GFA text are host code in the product; their parity is pinned end-to-end against
 * oracle/_ref/miniasm_ref and the golden GFA files.
 *
 * Records are byte-compatible with the reference's ma_hit_t / ma_sub_t / asg_arc_t.
 */
#ifndef MA_ORACLE_H
#define MA_ORACLE_H
#include <stdint.h>
#include <stddef.h>

typedef struct { uint64_t qns; uint32_t qe, tn, ts, te; uint32_t ml:31, rev:1; uint32_t bl:31, del:1; } orc_hit_t; /* miniasm.h:29-34 */
typedef struct { uint32_t s:31, del:1, e; } orc_sub_t;                                                              /* miniasm.h:38-40 */
typedef struct { uint64_t ul; uint32_t v; uint32_t ol:31, del:1; } orc_arc_t;                                       /* asg.h:7-11 */
typedef struct { int min_span, min_match, min_dp; float min_iden; int max_hang, min_ovlp; float int_frac;
                 int gap_fuzz, n_rounds, bub_dist, max_ext; float r0, r1, r2; } orc_opt_t;                          /* miniasm.h:12-27 */

#ifdef __cplusplus
extern "C" {
#endif

/* hit.c:19-22 with a TOTAL order: (qns, input position) -- the product's documented tie rule */
void orc_hit_sort(size_t n, orc_hit_t *a);
/* miniasm.h:86-104 ; returns MA_HT_* or len, arc in *p */
int orc_hit2arc(const orc_hit_t *h, int ql, int tl, int max_hang, float int_frac, int min_ovlp, orc_arc_t *p);
/* hit.c:109-160 ; sub must hold n_sub zeroed entries; returns #reads that keep an interval */
size_t orc_hit_sub(int min_dp, float min_iden, int end_clip, size_t n, const orc_hit_t *a, size_t n_sub, orc_sub_t *sub);
/* hit.c:162-193 */
size_t orc_hit_cut(const orc_sub_t *reg, int min_span, size_t n, orc_hit_t *a);
/* hit.c:195-216 */
size_t orc_hit_flt(const orc_sub_t *sub, int max_hang, int min_ovlp, size_t n, orc_hit_t *a, float *cov);
/* hit.c:218-223 */
void orc_sub_merge(size_t n_sub, orc_sub_t *a, const orc_sub_t *b);
/* hit.c:225-256 + hit.c:24-36 + sdict.c:69-86 ; seq_del[n_seq] in/out, map[n_seq] out; returns #hits kept, *n_seq_new */
size_t orc_hit_contained(const orc_opt_t *opt, uint32_t n_seq, uint8_t *seq_del, orc_sub_t *sub, size_t n, orc_hit_t *a, int32_t *map, uint32_t *n_seq_new);
/* asm.c:9-39 + asg.c:57-80 ; arcs must hold n entries; seq_len/seq_del [n_seq] out (sub may be NULL -> len_in);
 * arcs come out sorted by (ul, push order) ; returns n_arc */
size_t orc_sg_gen(const orc_opt_t *opt, uint32_t n_seq, const orc_sub_t *sub, const uint32_t *len_in, const uint8_t *del_in,
                  size_t n, const orc_hit_t *a, orc_arc_t *arcs, uint32_t *seq_len, uint8_t *seq_del);
/* the pieces of the two passes above, split where the sharded multi-GPU mode exchanges flags (tests/test_dist_gloo.py) */
void orc_contained_flags(const orc_opt_t *opt, const orc_sub_t *sub, size_t n, const orc_hit_t *a, uint8_t *r_cont, uint8_t *r_used);
size_t orc_sg_candidates(const orc_opt_t *opt, uint32_t n_seq, const orc_sub_t *sub, const uint32_t *len_in, const uint8_t *del_in,
                         size_t n, const orc_hit_t *a, orc_arc_t *arcs, uint32_t *seq_len, uint8_t *seq_del);
size_t orc_sg_finish(size_t n_arc, orc_arc_t *arcs, const uint8_t *seq_del);
uint32_t orc_arc_del_trans_range(uint32_t n_seq, size_t n_arc, orc_arc_t *a, const uint64_t *idx, const uint8_t *seq_del, int fuzz,
                                 uint32_t v_beg, uint32_t v_end, uint64_t *n_inner);
/* asg.c:27-36 ; idx must hold 2*n_seq entries */
void orc_arc_index(uint32_t n_seq, size_t n_arc, const orc_arc_t *a, uint64_t *idx);
/* asg.c:57-70 ; returns new n_arc */
size_t orc_arc_rm(size_t n_arc, orc_arc_t *a, const uint8_t *seq_del);
/* asg.c:148-186 (marking only) ; returns n_reduced ; *n_inner = iterations of the loop at asg.c:169 (roofline figure) */
uint32_t orc_arc_del_trans(uint32_t n_seq, size_t n_arc, orc_arc_t *a, const uint64_t *idx, const uint8_t *seq_del, int fuzz, uint64_t *n_inner);
/* asg.c:104-118 / 124-135 / 83-96 (marking only) */
uint32_t orc_arc_del_multi(uint32_t n_seq, size_t n_arc, orc_arc_t *a, const uint64_t *idx);
uint32_t orc_arc_del_asymm(uint32_t n_seq, size_t n_arc, orc_arc_t *a, const uint64_t *idx);
uint32_t orc_arc_del_short(uint32_t n_seq, size_t n_arc, orc_arc_t *a, const uint64_t *idx, float drop_ratio);

#ifdef __cplusplus
}
#endif
#endif
