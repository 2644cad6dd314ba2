/* asm.c -- hits -> string graph bridge (GPU), unitig construction (GPU) and the GFA / string-graph writers (host).
 * Reference: asm.c:9-39 (ma_sg_gen), :41-55 (ma_sg_print), :64-116 (ma_ug_destroy, ma_ug_print),
 * :121-210 (ma_ug_gen), :216-290 (ma_ug_seq).
 * The output line formats are the contract with downstream tools and are reproduced byte for byte.
 */
#include <stdio.h>
#include <stdlib.h>
#include <pthread.h>
#include <string.h>
#include "ma_host.h"

#define GPU(call) do { if ((call) != 0) ma_gpu_fail(__func__); } while (0)

asg_arc_t *ma_asg_arc_pushp(asg_t *g);

asg_t *ma_sg_gen(const ma_opt_t *opt, const sdict_t *d, const ma_sub_t *sub, size_t n_hits, const ma_hit_t *hit)
{
	mahip_ctx_t *c = ma_gpu();
	asg_t *g = asg_init();
	uint32_t i, n_arc = 0, R = d->n_seq;
	uint8_t *del = (uint8_t*)malloc(R ? R : 1);
	uint32_t *len = 0;
	for (i = 0; i < R; ++i) del[i] = d->seq[i].del;
	if (!sub) {
		len = (uint32_t*)malloc((R ? R : 1) * 4);
		for (i = 0; i < R; ++i) len[i] = d->seq[i].len;
	}
	GPU(mahip_set_shard(c, 0, 0xffffffffu));
	GPU(mahip_hits_upload(c, hit, n_hits, R));
	GPU(mahip_hits_index(c));
	if (sub) GPU(mahip_sub_upload(c, 0, sub, R));
	GPU(mahip_sg_gen(c, opt, sub != 0, len, del, &n_arc));
	GPU(mahip_asg_download(c, g));
	free(del); free(len);
	fprintf(MA_LOG, "[M::%s] read %d arcs\n", __func__, g->n_arc);
	return g;
}

/* ---- buffered text output: the writers below produce tens of thousands of short lines; formatting them with
 * fprintf costs more than every GPU pass together, so lines are assembled in a buffer with hand-rolled integer
 * formatting and written once.  Byte-for-byte the text the reference's fprintf calls produce. */
typedef struct { char *s; size_t n, m; FILE *fp; } obuf_t;

static inline void ob_need(obuf_t *o, size_t k)
{
	if (o->n + k > o->m) {
		o->m = (o->n + k) * 2 + 4096;
		o->s = (char*)realloc(o->s, o->m);
	}
}
static inline void ob_chr(obuf_t *o, char c) { ob_need(o, 1); o->s[o->n++] = c; }
static inline void ob_mem(obuf_t *o, const char *p, size_t l) { ob_need(o, l); memcpy(o->s + o->n, p, l); o->n += l; }
static inline void ob_str(obuf_t *o, const char *p) { ob_mem(o, p, strlen(p)); }
static inline void ob_int(obuf_t *o, int64_t x) /* %d */
{
	char t[24]; int k = 0; uint64_t u = x < 0 ? (uint64_t)(-x) : (uint64_t)x;
	do { t[k++] = (char)('0' + u % 10); u /= 10; } while (u);
	ob_need(o, (size_t)k + 1);
	if (x < 0) o->s[o->n++] = '-';
	while (k) o->s[o->n++] = t[--k];
}
static inline void ob_int6(obuf_t *o, uint32_t x) /* %.6d of a non-negative value */
{
	char t[24]; int k = 0;
	do { t[k++] = (char)('0' + x % 10); x /= 10; } while (x);
	while (k < 6) t[k++] = '0';
	ob_need(o, (size_t)k);
	while (k) o->s[o->n++] = t[--k];
}
static inline void ob_flush(obuf_t *o) { if (o->n) fwrite(o->s, 1, o->n, o->fp); free(o->s); o->s = 0; o->n = o->m = 0; }
/* "name:s+1-e" (or just the name without sub) */
static inline void ob_read(obuf_t *o, const sdict_t *d, const ma_sub_t *sub, uint32_t x)
{
	ob_str(o, d->seq[x].name);
	if (sub) { ob_chr(o, ':'); ob_int(o, (int)sub[x].s + 1); ob_chr(o, '-'); ob_int(o, (int)sub[x].e); }
}

void ma_sg_print(const asg_t *g, const sdict_t *d, const ma_sub_t *sub, FILE *fp) /* asm.c:41-55 */
{
	uint32_t i;
	obuf_t o = {0, 0, 0, fp};
	for (i = 0; i < g->n_arc; ++i) { /* "L\t%s:%d-%d\t%c\t%s:%d-%d\t%c\t%d:\tL1:i:%d\n" */
		const asg_arc_t *p = &g->arc[i];
		ob_mem(&o, "L\t", 2); ob_read(&o, d, sub, (uint32_t)(p->ul >> 33)); ob_chr(&o, '\t'); ob_chr(&o, "+-"[p->ul >> 32 & 1]); ob_chr(&o, '\t');
		ob_read(&o, d, sub, p->v >> 1); ob_chr(&o, '\t'); ob_chr(&o, "+-"[p->v & 1]); ob_chr(&o, '\t');
		ob_int(&o, (int)p->ol); ob_mem(&o, ":\tL1:i:", 7); ob_int(&o, (int)(uint32_t)p->ul); ob_chr(&o, '\n');
	}
	ob_flush(&o);
}

/* ---------------------------------------------------------------------------------------------- unitigs */

void ma_ug_destroy(ma_ug_t *ug)
{
	size_t i;
	if (ug == 0) return;
	for (i = 0; i < ug->u.n; ++i) { free(ug->u.a[i].a); free(ug->u.a[i].s); }
	free(ug->u.a);
	asg_destroy(ug->g);
	free(ug);
}

/* asm.c:121-210.  The unitigs are computed on the device (csrc/ug.hip: links, list ranking by pointer jumping, orientation and
 * numbering by the discovery vertex, segmented gather of the members); the host only unpacks the arrays into the reference's
 * ma_ug_t and finishes the (small) unitig graph: asg_cleanup = the reference's sort order + index (asm.c:208). */
ma_ug_t *ma_ug_from_device(mahip_ctx_t *c)
{
	ma_ug_t *ug = (ma_ug_t*)calloc(1, sizeof(ma_ug_t));
	uint32_t U = 0, M = 0, NA = 0, *u_n, *u_len, *u_start, *u_end, *u_off, k;
	uint64_t *mem;
	ug->g = asg_init();
	GPU(mahip_ug_gen(c, &U, &M, &NA));
	u_n = (uint32_t*)malloc(((size_t)U + 1) * 4 * 5);
	u_len = u_n + U; u_start = u_len + U; u_end = u_start + U; u_off = u_end + U;
	mem = (uint64_t*)malloc(((size_t)M + 1) * 8);
	ug->g->arc = (asg_arc_t*)malloc(((size_t)NA + 1) * sizeof(asg_arc_t));
	ug->g->m_arc = NA + 1; ug->g->n_arc = NA;
	GPU(mahip_ug_download(c, u_n, u_len, u_start, u_end, u_off, mem, ug->g->arc));
	ug->u.n = ug->u.m = U;
	ug->u.a = (ma_utg_t*)calloc(U ? U : 1, sizeof(ma_utg_t));
	for (k = 0; k < U; ++k) {
		ma_utg_t *p = &ug->u.a[k];
		uint32_t m = u_n[k];
		p->len = u_len[k]; p->circ = u_start[k] == UINT32_MAX;
		p->start = u_start[k]; p->end = u_end[k];
		p->n = m;
		if (m) { --m; m |= m >> 1; m |= m >> 2; m |= m >> 4; m |= m >> 8; m |= m >> 16; ++m; }
		p->m = m;
		p->a = (uint64_t*)malloc(8 * (size_t)(p->m ? p->m : 1));
		memcpy(p->a, mem + u_off[k], (size_t)p->n * 8);
		asg_seq_set(ug->g, (int)k, (int)p->len, 0);
	}
	free(u_n); free(mem);
	asg_cleanup(ug->g);
	return ug;
}

ma_ug_t *ma_ug_gen(asg_t *g) /* per-symbol form: the caller's host graph goes up first */
{
	mahip_ctx_t *c = ma_gpu();
	if (!g->is_srt || g->idx == 0) asg_cleanup(g);
	GPU(mahip_asg_upload(c, g));
	return ma_ug_from_device(c);
}

static inline void ob_utg(obuf_t *o, uint32_t id1, int circ) { ob_mem(o, "utg", 3); ob_int6(o, id1); ob_chr(o, "lc"[circ]); }

/* The three blocks of a unitig GFA (asm.c:77-116).  Formatting is bound by cache misses on names and intervals of reads
 * scattered over the dictionary, so big outputs are formatted by several threads, each into its own buffer over a
 * contiguous range of unitigs / links, and written in order: the text is the sequential one byte for byte. */
/* a segment = reads [j0, j1) of unitig u, l0 = offset of read j0 on the unitig; the segment with j0 == 0 also carries the
 * S line (and the circularising L lines): long unitigs are cut into several segments so that they can be formatted in parallel */
typedef struct { uint32_t u, j0, j1, l0; } fmt_seg_t;

static void fmt_units(obuf_t *o, const ma_ug_t *ug, const sdict_t *d, const ma_sub_t *sub, const fmt_seg_t *seg, size_t lo, size_t hi)
{
	size_t k;
	uint32_t j, l;
	for (k = lo; k < hi; ++k) {
		const uint32_t i = seg[k].u;
		const ma_utg_t *p = &ug->u.a[i];
		if (seg[k].j0 == 0) { /* S line, circularising L lines */
			ob_mem(o, "S\t", 2); ob_utg(o, i + 1, p->circ); ob_chr(o, '\t'); ob_str(o, p->s ? p->s : "*"); ob_mem(o, "\tLN:i:", 6); ob_int(o, (int)p->len); ob_chr(o, '\n');
			if (p->circ) {
				for (j = 0; j < 2; ++j) { /* "L\t%s\t+\t%s\t+\t0M\n" and the '-' twin */
					char c = "+-"[j];
					ob_mem(o, "L\t", 2); ob_utg(o, i + 1, p->circ); ob_chr(o, '\t'); ob_chr(o, c); ob_chr(o, '\t');
					ob_utg(o, i + 1, p->circ); ob_chr(o, '\t'); ob_chr(o, c); ob_mem(o, "\t0M\n", 4);
				}
			}
		}
		for (j = seg[k].j0, l = seg[k].l0; j < seg[k].j1; l += (uint32_t)p->a[j++]) { /* "a\t%s\t%d\t%s:%d-%d\t%c\t%d\n" */
			if (j + 4 < p->n) __builtin_prefetch(&d->seq[p->a[j + 4] >> 33]);
			if (j + 2 < p->n) __builtin_prefetch(d->seq[p->a[j + 2] >> 33].name);
			ob_mem(o, "a\t", 2); ob_utg(o, i + 1, p->circ); ob_chr(o, '\t'); ob_int(o, (int)l); ob_chr(o, '\t');
			ob_read(o, d, sub, (uint32_t)(p->a[j] >> 33)); ob_chr(o, '\t'); ob_chr(o, "+-"[p->a[j] >> 32 & 1]); ob_chr(o, '\t');
			ob_int(o, (int)(uint32_t)p->a[j]); ob_chr(o, '\n');
		}
	}
}

static void fmt_links(obuf_t *o, const ma_ug_t *ug, uint32_t lo, uint32_t hi)
{
	uint32_t i;
	for (i = lo; i < hi; ++i) { /* "L\tutg%.6d%c\t%c\tutg%.6d%c\t%c\t%dM\tSD:i:%d\n" */
		const asg_arc_t *e = &ug->g->arc[i];
		uint32_t u = (uint32_t)(e->ul >> 32), v = e->v;
		ob_mem(o, "L\t", 2); ob_utg(o, (u >> 1) + 1, ug->u.a[u >> 1].circ); ob_chr(o, '\t'); ob_chr(o, "+-"[u & 1]); ob_chr(o, '\t');
		ob_utg(o, (v >> 1) + 1, ug->u.a[v >> 1].circ); ob_chr(o, '\t'); ob_chr(o, "+-"[v & 1]); ob_chr(o, '\t');
		ob_int(o, (int)e->ol); ob_mem(o, "M\tSD:i:", 7); ob_int(o, (int)asg_arc_len(*e)); ob_chr(o, '\n');
	}
}

static void fmt_summary(obuf_t *o, const ma_ug_t *ug, const sdict_t *d, const ma_sub_t *sub, uint32_t lo, uint32_t hi)
{
	uint32_t i;
	for (i = lo; i < hi; ++i) { /* x lines: unitig summary */
		const ma_utg_t *u = &ug->u.a[i];
		if (u->start == UINT32_MAX) { /* "x\tutg%.6dc\t%d\t%d\n" */
			ob_mem(o, "x\t", 2); ob_utg(o, i + 1, 1); ob_chr(o, '\t'); ob_int(o, (int)u->len); ob_chr(o, '\t'); ob_int(o, (int)u->n); ob_chr(o, '\n');
		} else { /* "x\tutg%.6dl\t%d\t%d\t%d\t%d\t%s:%d-%d\t%c\t%s:%d-%d\t%c\n" */
			uint32_t c0 = asg_arc_n(ug->g, i << 1 | 0), c1 = asg_arc_n(ug->g, i << 1 | 1);
			ob_mem(o, "x\t", 2); ob_utg(o, i + 1, 0); ob_chr(o, '\t'); ob_int(o, (int)u->len); ob_chr(o, '\t'); ob_int(o, (int)u->n); ob_chr(o, '\t');
			ob_int(o, (int)c1); ob_chr(o, '\t'); ob_int(o, (int)c0); ob_chr(o, '\t');
			ob_read(o, d, sub, u->start >> 1); ob_chr(o, '\t'); ob_chr(o, "+-"[u->start & 1]); ob_chr(o, '\t');
			ob_read(o, d, sub, u->end >> 1); ob_chr(o, '\t'); ob_chr(o, "+-"[u->end & 1]); ob_chr(o, '\n');
		}
	}
}

typedef struct { const ma_ug_t *ug; const sdict_t *d; const ma_sub_t *sub; const fmt_seg_t *seg; size_t s_lo, s_hi; uint32_t u_lo, u_hi, l_lo, l_hi; obuf_t units, links, summary; } fmt_job_t;

static void *fmt_worker(void *arg)
{
	fmt_job_t *j = (fmt_job_t*)arg;
	{ /* room for the whole share up front (an `a` line is about 40 bytes, an `L` line 45, an `x` line 70): a buffer that grows by doubling copies its text again and again and touches twice the pages */
		size_t k, reads = 0;
		for (k = j->s_lo; k < j->s_hi; ++k) reads += (size_t)(j->seg[k].j1 - j->seg[k].j0) + 2;
		ob_need(&j->units, reads * 52 + 4096);
		ob_need(&j->links, (size_t)(j->l_hi - j->l_lo) * 56 + 4096);
		ob_need(&j->summary, (size_t)(j->u_hi - j->u_lo) * 96 + 4096);
	}
	fmt_units(&j->units, j->ug, j->d, j->sub, j->seg, j->s_lo, j->s_hi);
	fmt_links(&j->links, j->ug, j->l_lo, j->l_hi);
	fmt_summary(&j->summary, j->ug, j->d, j->sub, j->u_lo, j->u_hi);
	return 0;
}

#define FMT_MAX_THREADS 16
#define FMT_SEG_READS 2048u
extern double g_fmt_laps[2]; /* pipeline.c */
typedef struct { fmt_job_t *job; int T, t; char *dst; } fmt_copy_t;
static void *fmt_copy_worker(void *arg) /* thread t copies ITS three pieces to their places: the pages of a fresh block are first touched by sixteen threads, not one */
{
	fmt_copy_t *q = (fmt_copy_t*)arg;
	size_t off_u = 0, off_l = 0, off_s = 0;
	int t;
	for (t = 0; t < q->T; ++t) off_l += q->job[t].units.n;
	off_s = off_l;
	for (t = 0; t < q->T; ++t) off_s += q->job[t].links.n;
	for (t = 0; t < q->t; ++t) { off_u += q->job[t].units.n; off_l += q->job[t].links.n; off_s += q->job[t].summary.n; }
	memcpy(q->dst + off_u, q->job[q->t].units.s, q->job[q->t].units.n);
	memcpy(q->dst + off_l, q->job[q->t].links.s, q->job[q->t].links.n);
	memcpy(q->dst + off_s, q->job[q->t].summary.s, q->job[q->t].summary.n);
	return 0;
}
/* asm.c:77-116: the text goes to fp, or (fp == 0) into one malloc'ed block *buf of *len bytes */
static void ug_print_to(const ma_ug_t *ug, const sdict_t *d, const ma_sub_t *sub, FILE *fp, char **buf, size_t *len)
{
	const uint32_t nu = (uint32_t)ug->u.n, nl = ug->g->n_arc;
	uint64_t tot = 0, acc = 0;
	uint32_t i, j, l;
	size_t n_seg = 0, m_seg = (size_t)nu + 16, si;
	int T, t, k;
	fmt_seg_t *seg = (fmt_seg_t*)malloc(m_seg * sizeof(fmt_seg_t));
	const char *e_seg = getenv("MA_FMT_SEG"); /* reads per segment (tests shrink it to cut short unitigs too) */
	const uint32_t seg_reads = e_seg && atol(e_seg) > 0 ? (uint32_t)atol(e_seg) : FMT_SEG_READS;
	fmt_job_t job[FMT_MAX_THREADS];
	pthread_t th[FMT_MAX_THREADS];
	int started[FMT_MAX_THREADS];
	for (i = 0; i < nu; ++i) { /* segments in output order */
		const ma_utg_t *p = &ug->u.a[i];
		tot += p->n + 2;
		j = 0; l = 0;
		do {
			uint32_t j1 = p->n - j > seg_reads ? j + seg_reads : p->n, x;
			if (n_seg == m_seg) { m_seg <<= 1; seg = (fmt_seg_t*)realloc(seg, m_seg * sizeof(fmt_seg_t)); }
			seg[n_seg].u = i; seg[n_seg].j0 = j; seg[n_seg].j1 = j1; seg[n_seg].l0 = l; ++n_seg;
			for (x = j; x < j1; ++x) l += (uint32_t)p->a[x];
			j = j1;
		} while (j < p->n);
	}
	{
		const char *e = getenv("MA_FMT_GRAIN"); /* lines per thread at least: about 0.3 ms of formatting */
		long grain = e ? atol(e) : 6000;
		T = (int)(tot / (uint64_t)(grain > 0 ? grain : 1));
	}
	k = ma_ingest_threads();
	if (T > k) T = k;
	if (T > FMT_MAX_THREADS) T = FMT_MAX_THREADS;
	if (T < 1) T = 1;
	memset(job, 0, sizeof(job));
	for (t = 0, si = 0; t < T; ++t) { /* segment ranges of about equal read count; summary lines by unitig range, links by equal share */
		uint64_t want = tot * (uint64_t)(t + 1) / (uint64_t)T;
		job[t].ug = ug; job[t].d = d; job[t].sub = sub; job[t].seg = seg;
		job[t].s_lo = si;
		while (si < n_seg && (acc < want || t == T - 1)) { acc += (seg[si].j1 - seg[si].j0) + (seg[si].j0 == 0 ? 2 : 0); ++si; }
		job[t].s_hi = si;
		job[t].u_lo = (uint32_t)((uint64_t)nu * (uint64_t)t / (uint64_t)T);
		job[t].u_hi = (uint32_t)((uint64_t)nu * (uint64_t)(t + 1) / (uint64_t)T);
		job[t].l_lo = (uint32_t)((uint64_t)nl * (uint64_t)t / (uint64_t)T);
		job[t].l_hi = (uint32_t)((uint64_t)nl * (uint64_t)(t + 1) / (uint64_t)T);
	}
	const int timing = getenv("MA_PIPE_TIMING") != 0;
	const double t_begin = sys_realtime();
	double t_fmt;
	for (t = 1; t < T; ++t) {
		started[t] = pthread_create(&th[t], 0, fmt_worker, &job[t]) == 0;
		if (!started[t]) fmt_worker(&job[t]);
	}
	fmt_worker(&job[0]);
	for (t = 1; t < T; ++t) if (started[t]) pthread_join(th[t], 0);
	t_fmt = sys_realtime();
	if (fp) {
		for (t = 0; t < T; ++t) { job[t].units.fp = fp; ob_flush(&job[t].units); }
		for (t = 0; t < T; ++t) { job[t].links.fp = fp; ob_flush(&job[t].links); }
		for (t = 0; t < T; ++t) { job[t].summary.fp = fp; ob_flush(&job[t].summary); }
	} else { /* one block, every thread copies what it formatted */
		fmt_copy_t cp[FMT_MAX_THREADS];
		size_t total = 0;
		char *dst;
		for (t = 0; t < T; ++t) total += job[t].units.n + job[t].links.n + job[t].summary.n;
		dst = (char*)ma_big_alloc(total + 1);
		for (t = 0; t < T; ++t) { cp[t].job = job; cp[t].T = T; cp[t].t = t; cp[t].dst = dst; }
		for (t = 1; t < T; ++t) {
			started[t] = pthread_create(&th[t], 0, fmt_copy_worker, &cp[t]) == 0;
			if (!started[t]) fmt_copy_worker(&cp[t]);
		}
		fmt_copy_worker(&cp[0]);
		for (t = 1; t < T; ++t) if (started[t]) pthread_join(th[t], 0);
		for (t = 0; t < T; ++t) { free(job[t].units.s); free(job[t].links.s); free(job[t].summary.s); }
		dst[total] = 0;
		*buf = dst; *len = total;
	}
	free(seg);
	g_fmt_laps[0] = (t_fmt - t_begin) * 1e3; g_fmt_laps[1] = (sys_realtime() - t_fmt) * 1e3;
	if (timing) fprintf(stderr, "[T::ug_print] %d threads: format %.3f ms, %s %.3f ms\n", T, (t_fmt - t_begin) * 1e3, fp ? "write" : "copy into one block", (sys_realtime() - t_fmt) * 1e3);
}

void ma_ug_print(const ma_ug_t *ug, const sdict_t *d, const ma_sub_t *sub, FILE *fp) { ug_print_to(ug, d, sub, fp, 0, 0); }
/* the same text in one malloc'ed block (free() it): what a caller that wants the GFA in memory gets without a stream in between */
void ma_ug_print_mem(const ma_ug_t *ug, const sdict_t *d, const ma_sub_t *sub, char **buf, size_t *len) { ug_print_to(ug, d, sub, 0, buf, len); }

/* ---------------------------------------------------------------------------------------------- unitig sequences
 * reference asm.c:216-290 (ma_ug_seq): every read placed on a unitig contributes the first `len` bases of its kept
 * interval (forward) or the reverse complement of its last `len` bases (reverse).  FASTA/FASTQ, plain or gzip, read
 * with the record rules of the reference's reader (kseq.h:172-247): a record starts at a line beginning with '>' or
 * '@', the name ends at the first white space, sequence lines run until a line starting with '>', '@' or '+', a '+'
 * line is followed by as many quality characters as there are bases. */
#include <zlib.h>
#include <ctype.h>
#include <assert.h>

typedef struct { gzFile fp; unsigned char *buf; int beg, end, eof; } fq_stream_t;
#define FQ_BUF (1 << 18)

static inline int fq_getc(fq_stream_t *f)
{
	if (f->beg >= f->end) {
		if (f->eof) return -1;
		f->beg = 0; f->end = gzread(f->fp, f->buf, FQ_BUF);
		if (f->end < FQ_BUF) f->eof = 1;
		if (f->end <= 0) { f->end = 0; return -1; }
	}
	return f->buf[f->beg++];
}

typedef struct { char *s; size_t l, m; } fq_str_t;
static inline void fq_push(fq_str_t *t, int c)
{
	if (t->l + 2 > t->m) { t->m = t->m ? t->m << 1 : 256; t->s = (char*)realloc(t->s, t->m); }
	t->s[t->l++] = (char)c; t->s[t->l] = 0;
}
/* append the rest of the current line (without the line end, CR dropped like kseq.h:146 when the line is longer than 1) */
static int fq_line(fq_stream_t *f, fq_str_t *t)
{
	int c;
	size_t l0 = t->l;
	while ((c = fq_getc(f)) != -1 && c != '\n') fq_push(t, c);
	if (t->l - l0 > 1 && t->s[t->l - 1] == '\r') t->s[--t->l] = 0;
	return c;
}

/* next record: name and sequence (qualities are skipped); *last is the look-ahead marker; returns <0 at end */
static int fq_read(fq_stream_t *f, int *last, fq_str_t *name, fq_str_t *seq)
{
	int c;
	if (*last == 0) {
		while ((c = fq_getc(f)) != -1 && c != '>' && c != '@');
		if (c == -1) return -1;
		*last = c;
	}
	name->l = seq->l = 0;
	if (name->s) name->s[0] = 0;
	while ((c = fq_getc(f)) != -1 && !isspace(c)) fq_push(name, c); /* name up to the first white space */
	if (c == -1 && name->l == 0) return -1;
	if (c != '\n' && c != -1) while ((c = fq_getc(f)) != -1 && c != '\n'); /* comment */
	if (seq->s == 0) { seq->m = 256; seq->s = (char*)malloc(seq->m); }
	seq->s[0] = 0;
	while ((c = fq_getc(f)) != -1 && c != '>' && c != '+' && c != '@') {
		if (c == '\n') continue; /* empty line */
		fq_push(seq, c);
		fq_line(f, seq);
	}
	if (c == '>' || c == '@') *last = c; else *last = 0;
	if (c == '+') { /* FASTQ: skip the '+' line and as many quality characters as there are bases */
		size_t ql = 0;
		while ((c = fq_getc(f)) != -1 && c != '\n');
		while (ql < seq->l && (c = fq_getc(f)) != -1) if (c != '\n' && c != '\r') ++ql; else if (c == '\r') { /* CR inside qualities: kseq keeps line semantics */ }
		*last = 0;
	}
	return (int)seq->l;
}

typedef struct { uint32_t utg:31, ori:1, start, len; } utg_place_t;

/* The file is read here (one inflate stream, the reference's record rules); the bases of every read that sits on a unitig go to
 * the device in batches and a byte-gather kernel places them (forward copy / reverse complement, csrc/useq.hip); the unitig
 * strings come back once, at the end. */
#define USEQ_BATCH_BYTES ((size_t)256 << 20)
int ma_ug_seq(ma_ug_t *g, const sdict_t *d, const ma_sub_t *sub, const char *fn)
{
	mahip_ctx_t *c;
	fq_stream_t f;
	fq_str_t name = {0, 0, 0}, seq = {0, 0, 0};
	utg_place_t *pl;
	uint64_t *uoff, tot = 0;
	char *batch, *arena;
	mahip_useq_job_t *jobs;
	size_t n_jobs = 0, m_jobs = 1 << 16, n_batch = 0, m_batch = USEQ_BATCH_BYTES;
	uint32_t i, j;
	int last = 0;
	memset(&f, 0, sizeof(f));
	f.fp = fn && strcmp(fn, "-") ? gzopen(fn, "r") : gzdopen(fileno(stdin), "r");
	if (f.fp == 0) return -1;
	c = ma_gpu();
	f.buf = (unsigned char*)malloc(FQ_BUF);
	pl = (utg_place_t*)calloc(d->n_seq ? d->n_seq : 1, sizeof(utg_place_t));
	uoff = (uint64_t*)malloc((g->u.n + 1) * 8);
	for (i = 0; i < g->u.n; ++i) { /* where every read lands; unitig strings back to back, each with its terminator */
		const ma_utg_t *u = &g->u.a[i];
		uint32_t l = 0;
		uoff[i] = tot; tot += (uint64_t)u->len + 1;
		for (j = 0; j < u->n; ++j) {
			utg_place_t *t = &pl[u->a[j] >> 33];
			assert(t->len == 0);
			t->utg = i, t->ori = u->a[j] >> 32 & 1;
			t->start = l, t->len = (uint32_t)u->a[j];
			l += t->len;
		}
	}
	GPU(mahip_useq_begin(c, (size_t)tot));
	batch = (char*)malloc(m_batch);
	jobs = (mahip_useq_job_t*)malloc(m_jobs * sizeof(mahip_useq_job_t));
	while (fq_read(&f, &last, &name, &seq) >= 0) {
		int32_t id = name.s ? sd_get(d, name.s) : -1;
		const utg_place_t *t;
		const char *rs = seq.s;
		size_t rl = seq.l;
		if (id < 0 || pl[id].len == 0) continue;
		t = &pl[id];
		if (sub) {
			assert(sub[id].e - sub[id].s <= rl);
			rs += sub[id].s; rl = sub[id].e - sub[id].s;
		}
		if (n_batch + rl > m_batch || n_jobs == m_jobs) { /* batch full: place what is there */
			GPU(mahip_useq_batch(c, batch, n_batch, jobs, n_jobs));
			n_batch = 0; n_jobs = 0;
			if (rl > m_batch) { m_batch = rl; batch = (char*)realloc(batch, m_batch); }
		}
		memcpy(batch + n_batch, rs, rl);
		jobs[n_jobs].src_off = n_batch; jobs[n_jobs].dst_off = uoff[t->utg] + t->start;
		jobs[n_jobs].src_len = (uint32_t)rl; jobs[n_jobs].len = t->len; jobs[n_jobs].rev = t->ori; jobs[n_jobs].pad = 0;
		++n_jobs; n_batch += rl;
	}
	GPU(mahip_useq_batch(c, batch, n_batch, jobs, n_jobs));
	arena = (char*)malloc(tot ? tot : 1);
	GPU(mahip_useq_end(c, arena));
	for (i = 0; i < g->u.n; ++i) { /* ma_ug_destroy frees every string on its own */
		ma_utg_t *u = &g->u.a[i];
		u->s = (char*)malloc((size_t)u->len + 1);
		memcpy(u->s, arena + uoff[i], u->len);
		u->s[u->len] = 0;
	}
	free(arena); free(batch); free(jobs); free(uoff);
	free(pl); free(name.s); free(seq.s); free(f.buf);
	gzclose(f.fp);
	return 0;
}
