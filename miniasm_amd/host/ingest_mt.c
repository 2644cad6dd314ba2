/* ingest_mt.c -- multi-threaded PAF ingest: the same result as the sequential ma_hit_ingest (reference
 * hit.c:70-101 + paf.c:34-67 + sdict.c:27-45), byte for byte and id for id, from T threads.
 *
 * The reference's semantics are sequential: read ids are handed out in order of first appearance (query before
 * target within a line, only for lines that pass the span/match filter), and a 10-column line inherits `bl` from
 * the previous line.  Both survive a three-phase split of a plain (uncompressed, seekable) file:
 *   1. parallel  : the file is read and cut into T chunks at line starts; every thread parses its chunk without
 *                  modifying the text, keeps the records that pass the filter with THREAD-LOCAL name ids (local
 *                  first-appearance order) and remembers which records saw no `bl` yet in this chunk;
 *   2. sequential: chunks in file order feed their local names, in local first-appearance order, to sd_put():
 *                  the global first appearance of a name is its local first appearance in the earliest chunk that
 *                  has it, so the global ids come out exactly as in a sequential pass (T x R dictionary look-ups);
 *                  the inherited `bl` values are patched from the previous chunk's final state;
 *   3. parallel  : local ids -> global ids, hits (+ mirrored hits) written at prefix-summed offsets.
 * gzip / stdin input and small files take the sequential path in hits_host.c.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <fcntl.h>
#include <unistd.h>
#include <pthread.h>
#include <sys/stat.h>
#include <sys/mman.h>
#include "ma_host.h"

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
/* Pre-fault the pages of [p, p+bytes) that lie entirely inside it, in the calling thread.  Fresh malloc memory costs
 * one page fault per 4 KiB on first touch; with the faults interleaved into the emit loop that was 60 % of the
 * whole ingest.  Populating each thread's slice up front lets the kernel do it in bulk; failure is harmless. */
static void prefault(void *p, size_t bytes)
{
	uintptr_t a = ((uintptr_t)p + 4095) & ~(uintptr_t)4095, e = ((uintptr_t)p + bytes) & ~(uintptr_t)4095;
	if (e > a) (void)madvise((void*)a, e - a, MADV_POPULATE_WRITE);
}

/* large arrays are plain malloc (the returned hit array must be ordinary libc heap for the drop-in contract);
 * MADV_HUGEPAGE was tried and was slower on virtualised hosts (direct compaction in the fault path) */
static void *big_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }

#define MT_MIN_BYTES (8u << 20)
#define MT_MAX_THREADS 64

typedef struct { uint32_t q, t, qs, qe, ts, te, mlrev, bl; } lrec_t; /* parsed line with local name ids */
typedef struct { const char *p; uint32_t l, len; } lname_t;          /* name bytes (in the file buffer), read length */

typedef struct {
	/* input */
	const char *buf; size_t beg, end; /* lines starting in [beg,end) */
	int min_span, min_match, bi_dir; const sdict_t *excl;
	/* phase 1 output */
	lrec_t *rec; size_t n_rec, m_rec;
	lname_t *name; uint32_t n_name, m_name;
	uint32_t *hslot, n_hslot; /* open addressing: name index + 1 */
	size_t tot_lines, n_out;
	uint32_t first_bl_rec; /* records [0, first_bl_rec) inherit bl from the previous chunk (UINT32_MAX: all of them) */
	int bl_seen; uint32_t last_bl;
	/* phase 2 output */
	uint32_t *l2g;
	/* phase 3 */
	ma_hit_t *out; size_t out_off; uint32_t max_qs;
} chunk_t;

static inline uint32_t hash_bytes(const char *s, uint32_t l)
{
	uint32_t h = 2166136261u, i;
	for (i = 0; i < l; ++i) h = (h ^ (uint8_t)s[i]) * 16777619u;
	return h;
}

static uint32_t local_id(chunk_t *c, const char *p, uint32_t l, uint32_t len)
{
	uint32_t h = hash_bytes(p, l), m, s;
	if ((c->n_name + 1) * 2 > c->n_hslot) { /* grow + rehash */
		uint32_t i, nn = c->n_hslot ? c->n_hslot << 1 : 1u << 16, *ns = (uint32_t*)calloc(nn, 4);
		for (i = 0; i < c->n_name; ++i) {
			uint32_t k = hash_bytes(c->name[i].p, c->name[i].l) & (nn - 1);
			while (ns[k]) k = (k + 1) & (nn - 1);
			ns[k] = i + 1;
		}
		free(c->hslot); c->hslot = ns; c->n_hslot = nn;
	}
	m = c->n_hslot - 1;
	for (s = h & m; c->hslot[s]; s = (s + 1) & m) {
		const lname_t *q = &c->name[c->hslot[s] - 1];
		if (q->l == l && memcmp(q->p, p, l) == 0) return c->hslot[s] - 1;
	}
	if (c->n_name == c->m_name) {
		c->m_name = c->m_name ? c->m_name << 1 : 1u << 14;
		c->name = (lname_t*)realloc(c->name, (size_t)c->m_name * sizeof(lname_t));
	}
	c->name[c->n_name].p = p; c->name[c->n_name].l = l; c->name[c->n_name].len = len;
	c->hslot[s] = c->n_name + 1;
	return c->n_name++;
}

static inline uint32_t field_num(const char *p, uint32_t l)
{ /* strtol(field) truncated to 32 bits; plain short digit strings take the fast path */
	uint32_t x = 0, i;
	char tmp[64];
	if (l > 0 && l <= 9) {
		for (i = 0; i < l && (unsigned)(p[i] - '0') < 10u; ++i) x = x * 10 + (uint32_t)(p[i] - '0');
		if (i == l) return x;
	}
	if (l >= sizeof(tmp)) l = sizeof(tmp) - 1; /* strtol stops long before this on any sane input */
	memcpy(tmp, p, l); tmp[l] = 0;
	return (uint32_t)strtol(tmp, 0, 10);
}

static void *phase1(void *arg)
{
	chunk_t *c = (chunk_t*)arg;
	const char *p = c->buf + c->beg, *end = c->buf + c->end;
	char nmq[4096], nmt[4096];
	c->first_bl_rec = UINT32_MAX;
	c->m_rec = (c->end - c->beg) / 48 + 1024; /* PAF lines are rarely shorter than this: avoids regrowing in the common case */
	c->rec = (lrec_t*)big_alloc(c->m_rec * sizeof(lrec_t));
	prefault(c->rec, c->m_rec * sizeof(lrec_t));
	while (p < end) {
		const char *nl = (const char*)memchr(p, '\n', (size_t)(end - p)), *le = nl ? nl : end, *f[12];
		uint32_t fl[12], nf = 0, ql, qs, qe, tl, ts, te, ml, rev;
		const char *s = p, *q;
		size_t l = (size_t)(le - p);
		if (l > 1 && le[-1] == '\r') --le, --l; /* kseq.h:146 */
		for (q = s;; ++q) { /* split on TAB, keep the first 11 columns */
			if (q == le || *q == '\t') {
				if (nf < 11) f[nf] = s, fl[nf] = (uint32_t)(q - s);
				++nf; s = q + 1;
				if (q == le) break;
			}
		}
		p = nl ? nl + 1 : end;
		if (nf < 10) continue; /* paf.c:54 */
		++c->tot_lines;
		if (nf >= 11) { c->last_bl = field_num(f[10], fl[10]); if (!c->bl_seen) c->bl_seen = 1, c->first_bl_rec = (uint32_t)c->n_rec; }
		ql = field_num(f[1], fl[1]); qs = field_num(f[2], fl[2]); qe = field_num(f[3], fl[3]);
		rev = fl[4] > 0 && f[4][0] == '-';
		tl = field_num(f[6], fl[6]); ts = field_num(f[7], fl[7]); te = field_num(f[8], fl[8]);
		ml = field_num(f[9], fl[9]) & 0x7fffffffu;
		if (qe - qs < (uint32_t)c->min_span || te - ts < (uint32_t)c->min_span || (int)ml < c->min_match) continue; /* hit.c:85 */
		if (c->excl) { /* hit.c:86 (names must be NUL-terminated for sd_get) */
			if (fl[0] < sizeof(nmq) && fl[5] < sizeof(nmt)) {
				memcpy(nmq, f[0], fl[0]); nmq[fl[0]] = 0; memcpy(nmt, f[5], fl[5]); nmt[fl[5]] = 0;
				if (sd_get(c->excl, nmq) >= 0 || sd_get(c->excl, nmt) >= 0) continue;
			}
		}
		if (c->n_rec == c->m_rec) {
			c->m_rec = c->m_rec + (c->m_rec >> 1) + (1u << 16);
			c->rec = (lrec_t*)realloc(c->rec, c->m_rec * sizeof(lrec_t));
		}
		{
			lrec_t *r = &c->rec[c->n_rec++];
			r->q = local_id(c, f[0], fl[0], ql); /* query before target (hit.c:88,90) */
			r->t = local_id(c, f[5], fl[5], tl);
			r->qs = qs, r->qe = qe, r->ts = ts, r->te = te, r->mlrev = ml | rev << 31, r->bl = c->last_bl;
			c->n_out += 1 + (c->bi_dir && r->q != r->t);
		}
	}
	return 0;
}

static void *phase3(void *arg)
{
	chunk_t *c = (chunk_t*)arg;
	ma_hit_t *o = c->out + c->out_off;
	size_t i;
	uint32_t mx = 0;
	prefault(o, c->n_out * sizeof(ma_hit_t));
	for (i = 0; i < c->n_rec; ++i) {
		const lrec_t *r = &c->rec[i];
		uint32_t qid = c->l2g[r->q], tid = c->l2g[r->t];
		o->qns = (uint64_t)qid << 32 | r->qs; o->qe = r->qe; o->tn = tid; o->ts = r->ts; o->te = r->te;
		o->ml = r->mlrev & 0x7fffffffu; o->rev = r->mlrev >> 31; o->bl = r->bl; o->del = 0;
		if (r->qs > mx) mx = r->qs;
		++o;
		if (c->bi_dir && qid != tid) {
			o->qns = (uint64_t)tid << 32 | r->ts; o->qe = r->te; o->tn = qid; o->ts = r->qs; o->te = r->qe;
			o->ml = r->mlrev & 0x7fffffffu; o->rev = r->mlrev >> 31; o->bl = r->bl; o->del = 0;
			if (r->ts > mx) mx = r->ts;
			++o;
		}
	}
	c->max_qs = mx;
	return 0;
}

typedef struct { int fd; char *buf; size_t beg, end; } rd_t;
static void *reader(void *arg)
{
	rd_t *r = (rd_t*)arg;
	size_t off = r->beg;
	prefault(r->buf + r->beg, r->end - r->beg);
	while (off < r->end) {
		ssize_t k = pread(r->fd, r->buf + off, r->end - off, (off_t)off);
		if (k <= 0) break;
		off += (size_t)k;
	}
	return 0;
}

/* How many CPUs this process can keep busy at once: the online CPUs, cut by a CPU quota of the control group it runs in (cgroup v2 cpu.max, v1 cfs_quota_us).  A
 * container is often given a machine's 256 CPUs to look at and 16 CPUs' worth of time: threads beyond the quota only make the scheduler stop ALL of the process's
 * threads for the rest of each 100 ms period -- the one that feeds the GPU included (round 6: the tie walk's bucket phase took 1.2 s on 16, 32, 64 and 128 threads alike,
 * cpu.stat showed the throttling). */
static long quota_cpus(const char *path_max, const char *path_quota, const char *path_period)
{
	char buf[128];
	FILE *f;
	long q = -1, per = 100000;
	if (path_max && (f = fopen(path_max, "r")) != 0) { /* "max 100000" or "1600000 100000" */
		if (fgets(buf, sizeof(buf), f) && buf[0] != 'm') { if (sscanf(buf, "%ld %ld", &q, &per) < 1) q = -1; }
		fclose(f);
	} else if (path_quota && (f = fopen(path_quota, "r")) != 0) {
		if (fscanf(f, "%ld", &q) != 1) q = -1;
		fclose(f);
		if ((f = fopen(path_period, "r")) != 0) { if (fscanf(f, "%ld", &per) != 1) per = 100000; fclose(f); }
	}
	if (q <= 0 || per <= 0) return 0;
	return (q + per - 1) / per;
}

int ma_cpu_budget(void)
{
	static int cached;
	long n, q;
	char line[512], path[640];
	FILE *f;
	if (cached) return cached;
	n = sysconf(_SC_NPROCESSORS_ONLN);
	if (n < 1) n = 1;
	if ((q = quota_cpus("/sys/fs/cgroup/cpu.max", 0, 0)) > 0 && q < n) n = q;
	if ((q = quota_cpus(0, "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us")) > 0 && q < n) n = q;
	if ((f = fopen("/proc/self/cgroup", "r")) != 0) { /* v2 without a private cgroup namespace: "0::/path" -- the group's own file and its ancestors' */
		while (fgets(line, sizeof(line), f))
			if (strncmp(line, "0::/", 4) == 0) {
				char *e = line + strlen(line);
				while (e > line && (e[-1] == '\n' || e[-1] == '/')) *--e = 0;
				while (strlen(line) > 4) {
					snprintf(path, sizeof(path), "/sys/fs/cgroup%s/cpu.max", line + 3);
					if ((q = quota_cpus(path, 0, 0)) > 0 && q < n) n = q;
					e = strrchr(line + 3, '/');
					if (!e || e == line + 3) break;
					*e = 0;
				}
			}
		fclose(f);
	}
	cached = (int)n;
	return cached;
}

int ma_ingest_threads(void)
{
	const char *s = getenv("MA_THREADS");
	long n = s ? atol(s) : ma_cpu_budget();
	if (!s && n > 16) n = 16; /* the sequential id-merge grows with the chunk count: 16 is the measured sweet spot (EPYC 9575F) */
	if (n < 1) n = 1;
	if (n > MT_MAX_THREADS) n = MT_MAX_THREADS;
	return (int)n;
}

/* returns NULL when the input is not eligible (gzip, stdin, small, one thread): the caller then parses sequentially */
ma_hit_t *ma_hit_ingest_mt(const char *fn, int min_span, int min_match, sdict_t *d, size_t *n, int bi_dir, const sdict_t *excl,
                           size_t *tot_lines, uint32_t *max_qs)
{
	int fd, T = ma_ingest_threads(), t;
	struct stat st;
	size_t size, n_out = 0, tot = 0;
	char *buf;
	chunk_t *ch;
	pthread_t tid[MT_MAX_THREADS];
	rd_t rd[MT_MAX_THREADS];
	ma_hit_t *out;
	uint32_t mx = 0, prev_bl = 0;
	char namebuf[65536];

	const int timing = getenv("MA_PIPE_TIMING") != 0;
	double t0 = sys_realtime(), t1, t2, t3;
	if (T < 2 || fn == 0 || strcmp(fn, "-") == 0) return 0;
	fd = open(fn, O_RDONLY);
	if (fd < 0) return 0; /* the sequential path reports the error */
	if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || (size_t)st.st_size < MT_MIN_BYTES) { close(fd); return 0; }
	size = (size_t)st.st_size;
	buf = (char*)big_alloc(size + 1);
	if (buf == 0) { close(fd); return 0; }
	for (t = 0; t < T; ++t) { rd[t].fd = fd; rd[t].buf = buf; rd[t].beg = size / T * t; rd[t].end = t == T - 1 ? size : size / T * (t + 1); pthread_create(&tid[t], 0, reader, &rd[t]); }
	for (t = 0; t < T; ++t) pthread_join(tid[t], 0);
	close(fd);
	if (size >= 2 && (uint8_t)buf[0] == 0x1f && (uint8_t)buf[1] == 0x8b) { free(buf); return 0; } /* gzip */

	t1 = sys_realtime();
	ch = (chunk_t*)calloc(T, sizeof(chunk_t));
	for (t = 0; t < T; ++t) { /* chunk t owns the lines that START in [beg,end): move nominal cuts to the next line start */
		size_t b = size / T * t;
		if (t > 0) { const char *nl = (const char*)memchr(buf + b - 1, '\n', size - (b - 1)); b = nl ? (size_t)(nl - buf) + 1 : size; }
		ch[t].beg = b;
		if (t > 0) ch[t-1].end = b;
		ch[t].buf = buf; ch[t].min_span = min_span; ch[t].min_match = min_match; ch[t].bi_dir = bi_dir; ch[t].excl = excl;
	}
	ch[T-1].end = size;
	for (t = 0; t < T; ++t) pthread_create(&tid[t], 0, phase1, &ch[t]);
	for (t = 0; t < T; ++t) pthread_join(tid[t], 0);

	t2 = sys_realtime();
	for (t = 0; t < T; ++t) { /* phase 2: global ids in first-appearance order; inherited bl */
		chunk_t *c = &ch[t];
		uint32_t i, lim = c->first_bl_rec == UINT32_MAX ? (uint32_t)c->n_rec : c->first_bl_rec;
		c->l2g = (uint32_t*)malloc((c->n_name ? c->n_name : 1) * 4);
		for (i = 0; i < c->n_name; ++i) {
			const lname_t *q = &c->name[i];
			if (q->l < sizeof(namebuf)) { memcpy(namebuf, q->p, q->l); namebuf[q->l] = 0; c->l2g[i] = (uint32_t)sd_put(d, namebuf, q->len); }
			else { char *tmp = (char*)malloc((size_t)q->l + 1); memcpy(tmp, q->p, q->l); tmp[q->l] = 0; c->l2g[i] = (uint32_t)sd_put(d, tmp, q->len); free(tmp); }
		}
		for (i = 0; i < lim; ++i) c->rec[i].bl = prev_bl;
		if (c->bl_seen) prev_bl = c->last_bl;
		c->out_off = n_out;
		n_out += c->n_out; tot += c->tot_lines;
	}
	t3 = sys_realtime();
	out = (ma_hit_t*)big_alloc((n_out ? n_out : 1) * sizeof(ma_hit_t));
	for (t = 0; t < T; ++t) { ch[t].out = out; pthread_create(&tid[t], 0, phase3, &ch[t]); }
	for (t = 0; t < T; ++t) pthread_join(tid[t], 0);
	for (t = 0; t < T; ++t) {
		if (ch[t].max_qs > mx) mx = ch[t].max_qs;
		free(ch[t].rec); free(ch[t].name); free(ch[t].hslot); free(ch[t].l2g);
	}
	free(ch); free(buf);
	if (timing) fprintf(stderr, "[T::ingest_mt] %d threads: read %.3f  parse %.3f  ids %.3f  emit %.3f s\n", T, t1 - t0, t2 - t1, t3 - t2, sys_realtime() - t3);
	*n = n_out; *tot_lines = tot; *max_qs = mx;
	return out;
}
