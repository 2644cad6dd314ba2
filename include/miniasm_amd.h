/* miniasm_amd.h -- the drop-in C link interface of the MI355X-native overlap-graph hot path.
 *
 * This header declares, with identical memory layout and identical signatures, the data shapes and
 * entry points that the reference driver (reference main.c) binds to; a program written against the
 * reference headers links against libminiasm_amd.so unchanged.  Each item cites the reference
 * declaration it replaces (file:line under the reference tree).  Nothing here is device-specific:
 * plain pointers and sizes, libc-heap ownership exactly as in the reference.
 *
 * The HIP side sits one level below, behind include/mahip.h.
 */
#ifndef MINIASM_AMD_H
#define MINIASM_AMD_H

#include <stdio.h>
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ read-name dictionary (sdict.h:6-25) */

typedef struct {            /* sdict.h:6-9 : 16 bytes */
	char *name;             /* strdup'ed, owned by the dictionary */
	uint32_t len;           /* first-seen read length */
	uint32_t aux:31, del:1; /* scratch / "drop this read" */
} sd_seq_t;

typedef struct {            /* sdict.h:11-15 */
	uint32_t n_seq, m_seq;
	sd_seq_t *seq;
	void *h;                /* opaque name->id index (rebuilt by sd_squeeze) */
} sdict_t;

sdict_t *sd_init(void);                                         /* sdict.h:21 */
void     sd_destroy(sdict_t *d);                                /* sdict.h:22 */
int32_t  sd_put(sdict_t *d, const char *name, uint32_t len);    /* sdict.h:23 : ids dense, in first-appearance order */
int32_t  sd_get(const sdict_t *d, const char *name);            /* sdict.h:24 : -1 if absent */
int32_t *sd_squeeze(sdict_t *d);                                /* sdict.h:25 : returns calloc'ed old->new map (-1 dropped) */
void sd_hash(sdict_t *d);                                       /* sdict.c:55 (no header declares it): build the name index if it is not there */

/* ------------------------------------------------------------------ PAF reader (paf.h:20-32) */

typedef struct { size_t l, m; char *s; } ma_kstring_t;          /* same shape as paf.h:9-12 kstring_t */
typedef struct { void *fp; ma_kstring_t buf; } paf_file_t;      /* paf.h:15-18 */
typedef struct {                                                /* paf.h:20-24 */
	const char *qn, *tn;    /* point into the reader's line buffer */
	uint32_t ql, qs, qe, tl, ts, te;
	uint32_t ml:31, rev:1, bl;
} paf_rec_t;

paf_file_t *paf_open(const char *fn);                           /* paf.h:30 : plain, gz, or "-" for stdin */
int paf_close(paf_file_t *pf);                                  /* paf.h:31 */
int paf_read(paf_file_t *pf, paf_rec_t *r);                     /* paf.h:32 : <0 at EOF; lines with <10 fields skipped */
int paf_parse(int l, char *s, paf_rec_t *pr);                   /* paf.c:34 (no header declares it): one NUL-terminated line, split in place; <0 = fewer than 10 columns */

/* ------------------------------------------------------------------ timers (sys.h:8-11) */

double sys_cputime(void);
double sys_realtime(void);
void   sys_init(void);
const char *sys_timestamp(void);
void   sys_liftrlimit(void);                                    /* sys.c:22 (no header declares it): RLIMIT_AS soft limit := hard limit */

/* ------------------------------------------------------------------ string graph (asg.h:7-42) */

typedef struct {            /* asg.h:7-11 : 16 bytes */
	uint64_t ul;            /* u<<32 | len ; u = read<<1|end */
	uint32_t v;
	uint32_t ol:31, del:1;
} asg_arc_t;

typedef struct { uint32_t len:31, del:1; } asg_seq_t;           /* asg.h:13-15 */

typedef struct {            /* asg.h:17-23 */
	uint32_t m_arc, n_arc:31, is_srt:1;
	asg_arc_t *arc;
	uint32_t m_seq, n_seq:31, is_symm:1;
	asg_seq_t *seq;
	uint64_t *idx;          /* idx[v] = first_arc<<32 | n_arcs */
} asg_t;

typedef struct { size_t n, m; uint64_t *a; } asg64_v;           /* asg.h:25 */

#define asg_arc_len(arc) ((uint32_t)(arc).ul)                   /* asg.h:27 */
#define asg_arc_n(g, v) ((uint32_t)(g)->idx[(v)])               /* asg.h:28 */
#define asg_arc_a(g, v) (&(g)->arc[(g)->idx[(v)]>>32])          /* asg.h:29 */

asg_t *asg_init(void);                                          /* asg.h:31 */
void asg_destroy(asg_t *g);                                     /* asg.h:32 */
void asg_seq_set(asg_t *g, int sid, int len, int del);          /* asg.h:33 */
void asg_symm(asg_t *g);                                        /* asg.h:34 */
void asg_cleanup(asg_t *g);                                     /* asg.h:35 */
int asg_arc_del_short(asg_t *g, float drop_ratio);              /* asg.h:37 */
int asg_arc_del_trans(asg_t *g, int fuzz);                      /* asg.h:38 : HIP */
int asg_cut_tip(asg_t *g, int max_ext);                         /* asg.h:39 */
int asg_cut_internal(asg_t *g, int max_ext);                    /* asg.h:40 */
int asg_cut_biloop(asg_t *g, int max_ext);                      /* asg.h:41 */
int asg_pop_bubble(asg_t *g, int max_dist);                     /* asg.h:42 */
/* non-header externals of the reference's asg.c that other objects may bind (asg.c:22,27,38,57,104,124,217) */
void asg_arc_sort(asg_t *g);
uint64_t *asg_arc_index_core(size_t max_seq, size_t n, const asg_arc_t *a);
void asg_arc_index(asg_t *g);
void asg_arc_rm(asg_t *g);
int asg_arc_del_multi(asg_t *g);
int asg_arc_del_asymm(asg_t *g);
int asg_extend(const asg_t *g, uint32_t v, int max_ext, asg64_v *a);

/* ------------------------------------------------------------------ overlap hits (miniasm.h:10-75) */

extern int ma_verbose;                                          /* miniasm.h:10 */

typedef struct {            /* miniasm.h:12-27 : 56 bytes */
	int min_span, min_match, min_dp;
	float min_iden;
	int max_hang, min_ovlp;
	float int_frac;
	int gap_fuzz, n_rounds, bub_dist, max_ext;
	float min_ovlp_drop_ratio, max_ovlp_drop_ratio, final_ovlp_drop_ratio;
} ma_opt_t;

typedef struct {            /* miniasm.h:29-34 : 32 bytes */
	uint64_t qns;           /* query id<<32 | query start */
	uint32_t qe, tn, ts, te;
	uint32_t ml:31, rev:1;
	uint32_t bl:31, del:1;
} ma_hit_t;

typedef struct { size_t n, m; ma_hit_t *a; } ma_hit_v;          /* miniasm.h:36 */

typedef struct { uint32_t s:31, del:1, e; } ma_sub_t;           /* miniasm.h:38-40 : kept interval [s,e) */

typedef struct {            /* miniasm.h:42-48 */
	uint32_t len:31, circ:1;
	uint32_t start, end;
	uint32_t m, n;
	uint64_t *a;            /* (vertex<<32 | arc length) per read on the unitig */
	char *s;
} ma_utg_t;
typedef struct { size_t n, m; ma_utg_t *a; } ma_utg_v;          /* miniasm.h:50 */
typedef struct { ma_utg_v u; asg_t *g; } ma_ug_t;               /* miniasm.h:52-55 */

#define MA_HT_INT        (-1)                                   /* miniasm.h:81-84 */
#define MA_HT_QCONT      (-2)
#define MA_HT_TCONT      (-3)
#define MA_HT_SHORT_OVLP (-4)

void ma_opt_init(ma_opt_t *opt);                                                                               /* miniasm.h:61 */
sdict_t *ma_hit_no_cont(const char *fn, int min_span, int min_match, int max_hang, float int_frac);            /* :62 */
ma_hit_t *ma_hit_read(const char *fn, int min_span, int min_match, sdict_t *d, size_t *n, int bi_dir,
                      const sdict_t *excl);                                                                    /* :63 ; sort on HIP */
ma_sub_t *ma_hit_sub(int min_dp, float min_iden, int end_clip, size_t n, const ma_hit_t *a, size_t n_sub);     /* :64 ; HIP */
size_t ma_hit_cut(const ma_sub_t *reg, int min_span, size_t n, ma_hit_t *a);                                   /* :65 ; HIP */
size_t ma_hit_flt(const ma_sub_t *sub, int max_hang, int min_ovlp, size_t n, ma_hit_t *a, float *cov);         /* :66 ; HIP */
void ma_sub_merge(size_t n_sub, ma_sub_t *a, const ma_sub_t *b);                                               /* :67 */
size_t ma_hit_contained(const ma_opt_t *opt, sdict_t *d, ma_sub_t *sub, size_t n, ma_hit_t *a);                /* :68 ; HIP */
asg_t *ma_sg_gen(const ma_opt_t *opt, const sdict_t *d, const ma_sub_t *sub, size_t n_hits,
                 const ma_hit_t *hit);                                                                         /* :70 ; HIP */
void ma_sg_print(const asg_t *g, const sdict_t *d, const ma_sub_t *sub, FILE *fp);                             /* :71 */
ma_ug_t *ma_ug_gen(asg_t *g);                                                                                  /* :72 */
int ma_ug_seq(ma_ug_t *g, const sdict_t *d, const ma_sub_t *sub, const char *fn);                              /* :73 ; host: FASTA/FASTQ (gz) -> unitig strings */
void ma_ug_print(const ma_ug_t *ug, const sdict_t *d, const ma_sub_t *sub, FILE *fp);                          /* :74 */
void ma_ug_destroy(ma_ug_t *ug);                                                                               /* :75 */
/* non-header externals of the reference's hit.c (hit.c:19,24) */
void ma_hit_sort(size_t n, ma_hit_t *a);                                                                       /* HIP */
void ma_hit_mark_unused(sdict_t *d, size_t n, const ma_hit_t *a);

/* ------------------------------------------------------------------ additions (not in the reference) */

/* Whole path in one call, hits resident in HBM between passes (what the CLI and bench.py use).
 * fn: PAF path; outfmt: "bed" | "paf" | "sg" | "ug"; stage: the reference's -S gate (main.c:121-182);
 * flags: bit0 = skip 1-pass selection (-1), bit1 = skip 2-pass selection (-2), bit2 = bi_dir off (-b),
 * bit3 = prefilter contained reads (-R).  Returns 0 on success; exits like the reference on I/O errors. */
int ma_pipeline_run(const ma_opt_t *opt, const char *fn, const char *outfmt, int stage, int flags, FILE *out);

#ifdef __cplusplus
}
#endif

#endif
