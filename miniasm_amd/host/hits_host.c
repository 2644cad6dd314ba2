/* hit.c -- host side of the hit ingest/filter entry points (reference hit.c:19-256, miniasm.h:61-68).
 *
 * The text ingest (PAF parse + name dictionary) is host work; every data-parallel pass is a call into the
 * HIP library behind include/mahip.h.  These per-symbol entry points keep the reference's contract exactly
 * (libc-heap arrays, in-place compaction, same return values, same log lines) so that the reference's own
 * driver links against them unchanged; each call therefore pays one H2D and one D2H.  The resident path
 * that keeps the hits in HBM across passes is ma_pipeline_run() in pipeline.c.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ma_host.h"
#include "ma_core.h"

int ma_verbose = 3; /* reference common.c:3 */

void ma_opt_init(ma_opt_t *opt) /* reference common.c:5-23 */
{
	memset(opt, 0, sizeof(*opt));
	opt->min_span = 2000, opt->min_match = 100, opt->min_dp = 3, opt->min_iden = .05f;
	opt->max_hang = 1000, opt->min_ovlp = opt->min_span, opt->int_frac = .8f;
	opt->gap_fuzz = 1000, opt->n_rounds = 2, opt->bub_dist = 50000, opt->max_ext = 4;
	opt->min_ovlp_drop_ratio = .5f, opt->max_ovlp_drop_ratio = .7f, opt->final_ovlp_drop_ratio = .8f;
}

/* ---------------------------------------------------------------------------------------------- GPU context */

static mahip_ctx_t *g_ctx;

void ma_gpu_fail(const char *where)
{
	fprintf(stderr, "[E::%s] GPU path failed: %s\n", where, mahip_strerror());
	exit(1);
}

/* orderly teardown before the HIP runtime's own exit handlers run (a process that ends with a live stream and pinned
 * buffers was seen to crash in the runtime's teardown about once in 200 runs) */
static void ma_gpu_shutdown(void)
{
	if (g_ctx) { mahip_destroy(g_ctx); g_ctx = 0; }
}

mahip_ctx_t *ma_gpu(void)
{
	if (g_ctx == 0) {
		const char *s = getenv("MA_GPU_DEVICE");
		if (s == 0) s = getenv("LOCAL_RANK");
		const int timing = getenv("MA_PIPE_TIMING") != 0;
		double t0 = sys_realtime(), t1, t2, t3;
		int n_dev = mahip_device_count(); /* the first HIP call: the runtime comes up here (driver, queues, the code objects are registered) */
		t1 = sys_realtime();
		g_ctx = mahip_create(s ? atoi(s) : 0, 0);
		if (g_ctx == 0) ma_gpu_fail("ma_gpu");
		t2 = sys_realtime();
		if (timing) { /* the first launch loads the library's code object onto the device: made visible here, otherwise it hides in the first pass that launches */
			mahip_first_launch(g_ctx);
			t3 = sys_realtime();
			fprintf(stderr, "[T::init] at %.3f s after main: HIP runtime %.3f s (%d devices)  context (stream, counters, pinned mailbox) %.3f s  first launch + sync %.3f s\n", t0, t1 - t0, n_dev, t2 - t1, t3 - t2);
		}
		atexit(ma_gpu_shutdown);
	}
	return g_ctx;
}

#define GPU(call) do { if ((call) != 0) ma_gpu_fail(__func__); } while (0)

static uint32_t max_read_id(size_t n, const ma_hit_t *a)
{
	size_t i;
	uint32_t m = 0;
	for (i = 0; i < n; ++i) {
		uint32_t q = (uint32_t)(a[i].qns >> 32);
		if (q > m) m = q;
		if (a[i].tn > m) m = a[i].tn;
	}
	return n ? m + 1 : 0;
}

/* ---------------------------------------------------------------------------------------------- ingest */

/* the span / match gate every reader applies to a line before anything else (hit.c:52,85) */
static inline int line_passes(const paf_rec_t *r, int min_span, int min_match)
{
	return r->qe - r->qs >= (uint32_t)min_span && r->te - r->ts >= (uint32_t)min_span && (int)r->ml >= min_match;
}

static paf_file_t *open_or_die(const char *fn, const char *who)
{
	paf_file_t *fp = paf_open(fn);
	if (fp) return fp;
	fprintf(stderr, "[E::%s] could not open PAF file %s\n", who, fn);
	exit(1);
}

/* hit.c:38-68 behind the per-symbol ABI (and MA_HOST_PARSE=1): the names of reads that some line shows clearly inside a read more than
 * twice their length.  The per-line verdict is mc_no_cont (csrc/ma_core.h) -- the same function the device parser's k_paf_nocont runs --
 * so there is ONE statement of the rule; this loop only feeds it lines and collects names in order of first appearance. */
sdict_t *ma_hit_no_cont(const char *fn, int min_span, int min_match, int max_hang, float int_frac)
{
	paf_file_t *fp = open_or_die(fn, __func__);
	sdict_t *excl = sd_init();
	paf_rec_t r;
	memset(&r, 0, sizeof(r));
	while (paf_read(fp, &r) >= 0) {
		int who;
		if (!line_passes(&r, min_span, min_match)) continue;
		who = mc_no_cont(r.ql, r.qs, r.qe, r.tl, r.ts, r.te, r.rev, max_hang, int_frac); /* 1: the target is inside the query, 2: the query inside the target */
		if (who == 1) sd_put(excl, r.tn, r.tl);
		else if (who == 2) sd_put(excl, r.qn, r.ql);
	}
	paf_close(fp);
	if (ma_verbose >= 3) fprintf(MA_LOG, "[M::%s::%s] dropped %d contained reads\n", __func__, sys_timestamp(), excl->n_seq);
	return excl;
}

static uint32_t g_ingest_max_qs; /* largest query start stored by the last ingest: lets the device sort plan its digits */
uint32_t ma_ingest_max_qs(void) { return g_ingest_max_qs; }

ma_hit_t *ma_hit_ingest(const char *fn, int min_span, int min_match, sdict_t *d, size_t *n, int bi_dir, const sdict_t *excl)
{ /* hit.c:70-103: filter, name -> id in first-appearance order (query before target), store hit + mirrored hit */
	uint32_t max_qs = 0;
	paf_file_t *fp;
	paf_rec_t r;
	ma_hit_t *a = 0;
	size_t na = 0, ma = 0, i, tot = 0, tot_len = 0;
	a = ma_hit_ingest_mt(fn, min_span, min_match, d, &na, bi_dir, excl, &tot, &max_qs); /* plain files: chunk-parallel parse */
	if (a) goto done;
	fp = open_or_die(fn, "ma_hit_read"); /* gzip / stdin: one line at a time */
	memset(&r, 0, sizeof(r));
	for (; paf_read(fp, &r) >= 0; ++tot) {
		uint32_t id[2];
		int k, n_rec;
		if (!line_passes(&r, min_span, min_match)) continue;
		if (excl && (sd_get(excl, r.qn) >= 0 || sd_get(excl, r.tn) >= 0)) continue;
		id[0] = (uint32_t)sd_put(d, r.qn, r.ql); /* ids in order of first appearance, the query column first (sdict.c:27-45) */
		id[1] = (uint32_t)sd_put(d, r.tn, r.tl);
		n_rec = bi_dir && id[0] != id[1] ? 2 : 1;   /* the line as it stands, then seen from the target (hit.c:87-98) */
		if (na + 2 > ma) {
			ma = ma ? ma << 1 : 1u << 16;
			a = (ma_hit_t*)realloc(a, ma * sizeof(ma_hit_t));
		}
		for (k = 0; k < n_rec; ++k) {
			const uint32_t beg[2] = { r.qs, r.ts }, end[2] = { r.qe, r.te };
			ma_hit_t *p = &a[na++];
			p->qns = (uint64_t)id[k] << 32 | beg[k]; p->qe = end[k];
			p->tn = id[k ^ 1]; p->ts = beg[k ^ 1]; p->te = end[k ^ 1];
			p->rev = r.rev; p->ml = r.ml; p->bl = r.bl; p->del = 0;
			if (beg[k] > max_qs) max_qs = beg[k];
		}
	}
	paf_close(fp);
done:
	for (i = 0; i < d->n_seq; ++i) tot_len += d->seq[i].len;
	if (ma_verbose >= 3)
		fprintf(MA_LOG, "[M::%s::%s] read %ld hits; stored %ld hits and %d sequences (%ld bp)\n", "ma_hit_read", sys_timestamp(), (long)tot, (long)na, d->n_seq, (long)tot_len);
	if (a == 0) a = (ma_hit_t*)malloc(sizeof(ma_hit_t));
	*n = na;
	g_ingest_max_qs = max_qs;
	return a;
}

void ma_hit_sort(size_t n, ma_hit_t *a) /* hit.c:19-22 */
{
	mahip_ctx_t *c = ma_gpu();
	size_t m = 0;
	if (n < 2) return;
	GPU(mahip_set_shard(c, 0, 0xffffffffu));
	GPU(mahip_hits_upload(c, a, n, 0));
	GPU(mahip_hits_sort(c));
	GPU(mahip_hits_download(c, a, &m));
}

ma_hit_t *ma_hit_read(const char *fn, int min_span, int min_match, sdict_t *d, size_t *n, int bi_dir, const sdict_t *excl)
{
	ma_hit_t *a;
	if (excl == 0 && d->n_seq == 0 && ma_gpu_parse_enabled()) { /* text -> sorted records on the device, one download */
		mahip_ctx_t *c = ma_gpu();
		size_t m = 0;
		int rc = strcmp(fn, "-") != 0 ? ma_hit_ingest_gpu(c, fn, min_span, min_match, d, n, bi_dir) : -2;
		if (rc == -1) {
			fprintf(stderr, "[E::%s] could not open PAF file %s\n", __func__, fn);
			exit(1);
		}
		if (rc == 0) {
			a = (ma_hit_t*)malloc((*n ? *n : 1) * sizeof(ma_hit_t));
			if (*n) {
				GPU(mahip_hits_sort(c));
				GPU(mahip_hits_download(c, a, &m));
			}
			return a;
		} /* -2: stdin, or the text stage does not fit the device: host reader */
	}
	a = ma_hit_ingest(fn, min_span, min_match, d, n, bi_dir, excl);
	ma_hit_sort(*n, a);
	return a;
}

/* ---------------------------------------------------------------------------------------------- passes */

ma_sub_t *ma_hit_sub(int min_dp, float min_iden, int end_clip, size_t n, const ma_hit_t *a, size_t n_sub) /* hit.c:109-160 */
{
	mahip_ctx_t *c = ma_gpu();
	ma_sub_t *sub = (ma_sub_t*)calloc(n_sub ? n_sub : 1, sizeof(ma_sub_t));
	size_t n_remained = 0;
	GPU(mahip_set_shard(c, 0, 0xffffffffu));
	GPU(mahip_hits_upload(c, a, n, (uint32_t)n_sub));
	GPU(mahip_hits_index(c));
	GPU(mahip_hits_sub(c, min_dp, min_iden, end_clip, 0, &n_remained));
	if (n_sub) GPU(mahip_sub_download(c, 0, sub, 0));
	if (ma_verbose >= 3)
		fprintf(MA_LOG, "[M::%s::%s] %ld query sequences remain after sub\n", __func__, sys_timestamp(), (long)n_remained);
	return sub;
}

size_t ma_hit_cut(const ma_sub_t *reg, int min_span, size_t n, ma_hit_t *a) /* hit.c:162-193 */
{
	mahip_ctx_t *c = ma_gpu();
	size_t m = 0;
	uint32_t n_seq = max_read_id(n, a);
	GPU(mahip_hits_upload(c, a, n, n_seq));
	GPU(mahip_hits_index(c));
	GPU(mahip_sub_upload(c, 0, reg, n_seq));
	GPU(mahip_hits_cut(c, 0, min_span, &m));
	GPU(mahip_hits_download(c, a, &m));
	if (ma_verbose >= 3)
		fprintf(MA_LOG, "[M::%s::%s] %ld hits remain after cut\n", __func__, sys_timestamp(), (long)m);
	return m;
}

size_t ma_hit_flt(const ma_sub_t *sub, int max_hang, int min_ovlp, size_t n, ma_hit_t *a, float *cov) /* hit.c:195-216 */
{
	mahip_ctx_t *c = ma_gpu();
	size_t m = 0;
	uint32_t n_seq = max_read_id(n, a);
	GPU(mahip_hits_upload(c, a, n, n_seq));
	GPU(mahip_hits_index(c));
	GPU(mahip_sub_upload(c, 0, sub, n_seq));
	GPU(mahip_hits_flt(c, 0, max_hang, min_ovlp, &m, cov));
	GPU(mahip_hits_download(c, a, &m));
	if (ma_verbose >= 3)
		fprintf(MA_LOG, "[M::%s::%s] %ld hits remain after filtering; crude coverage after filtering: %.2f\n", __func__, sys_timestamp(), (long)m, *cov);
	return m;
}

void ma_sub_merge(size_t n_sub, ma_sub_t *a, const ma_sub_t *b) /* hit.c:218-223: R x 8 bytes, not worth a launch from the per-symbol path */
{
	size_t i;
	for (i = 0; i < n_sub; ++i)
		a[i].e = a[i].s + b[i].e, a[i].s += b[i].s;
}

void ma_hit_mark_unused(sdict_t *d, size_t n, const ma_hit_t *a) /* hit.c:24-36 (kept for link compatibility; the GPU path folds it into contained) */
{
	size_t i;
	for (i = 0; i < d->n_seq; ++i) d->seq[i].aux = 0;
	for (i = 0; i < n; ++i) d->seq[a[i].qns >> 32].aux = d->seq[a[i].tn].aux = 1;
	for (i = 0; i < d->n_seq; ++i) {
		if (!d->seq[i].aux) d->seq[i].del = 1;
		else d->seq[i].aux = 0;
	}
}

size_t ma_hit_contained(const ma_opt_t *opt, sdict_t *d, ma_sub_t *sub, size_t n, ma_hit_t *a) /* hit.c:225-256 */
{
	mahip_ctx_t *c = ma_gpu();
	size_t m = 0, i;
	uint32_t n_new = 0, old_n_seq = d->n_seq;
	uint8_t *del = (uint8_t*)malloc(old_n_seq ? old_n_seq : 1);
	int32_t *map;
	for (i = 0; i < old_n_seq; ++i) del[i] = d->seq[i].del;
	GPU(mahip_hits_upload(c, a, n, old_n_seq));
	GPU(mahip_hits_index(c));
	GPU(mahip_sub_upload(c, 0, sub, old_n_seq));
	GPU(mahip_hits_contained(c, opt, del, &n_new, &m));
	GPU(mahip_seqdel_download(c, del));
	for (i = 0; i < old_n_seq; ++i) d->seq[i].del = del[i], d->seq[i].aux = 0;
	map = sd_squeeze(d); /* same monotone map as the device's: frees dropped names, rebuilds the index */
	if (d->n_seq != n_new) { fprintf(stderr, "[E::%s] squeeze mismatch: host %u vs device %u\n", __func__, d->n_seq, n_new); exit(1); }
	GPU(mahip_sub_download(c, 0, sub, 1));
	GPU(mahip_hits_download(c, a, &m));
	free(map); free(del);
	if (ma_verbose >= 3)
		fprintf(MA_LOG, "[M::%s::%s] %d sequences and %ld hits remain after containment removal\n", __func__, sys_timestamp(), d->n_seq, (long)m);
	return m;
}
