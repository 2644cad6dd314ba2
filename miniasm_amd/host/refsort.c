/* refsort.c -- the ORDER the reference's sorts produce, as a permutation (exact-tie mode).
 *
 * Both reference sorts (radix_sort_hit hit.c:13, radix_sort_asg asg.c:9) are instances of one in-place MSD radix
 * sort (ksort.h:134-183): 8-bit digits from the top byte of a 64-bit key, a cycle-leader ("American flag")
 * permutation per level, insertion sort for runs of <= 64.  It is not stable: records with equal keys end in an
 * order that is a deterministic function of the whole input order, and that order is observable downstream
 * (arc push order, order inside a vertex's arc list).  The GPU sorts are stable, so on inputs with equal keys they
 * realise a different (documented) tie order.  Exact-tie mode replaces the order, not the data movement: this
 * file runs the reference procedure on (key, input index) pairs -- 16 bytes instead of 32-byte records -- and
 * the device gathers through the resulting permutation.
 *
 * The walk of one level is inherently sequential (it is a greedy Euler walk over the bucket graph: the slot an
 * element lands in depends on how many arrivals its bucket has seen), so the top level runs on one thread;
 * the buckets below it are independent and are spread over worker threads.  A level where every key has the
 * same digit moves nothing (each element is already "home" when the head reaches it) and is skipped by looking
 * at the bits that vary in the range.  Own implementation (index based); pinned against the reference build by
 * tests/test_host_vs_ref.py::test_refsort_perm_matches_reference_sort.
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include "ma_host.h"
#include <time.h>
#include <unistd.h>
#include <stdio.h>
static double rs_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
static int rs_timing = -1;
#define RS_T0 double t0_ = rs_now()
#define RS_LAP(what) do { if (rs_timing < 0) rs_timing = getenv("MA_REFSORT_TIMING") != 0; if (rs_timing) { double t1_ = rs_now(); fprintf(stderr, "[T::refsort] %-12s %.3f s\n", what, t1_ - t0_); t0_ = t1_; } } while (0)

static int rs_late_flag = -1;
static int rs_late(void) { if (rs_late_flag < 0) rs_late_flag = getenv("MA_REFSORT_LATE") != 0; return rs_late_flag; }

#define RS_MAX_THREADS 128
#define RS_SMALL 64           /* RS_MIN_SIZE ksort.h:132 */
/* The walk touches the 256 bucket heads in an order the hardware prefetchers cannot follow (they track a few dozen streams), so every
 * new cache line of a bucket used to be a miss on the walk's dependent chain; each store now asks for the line a few ahead of its head. */
#define RS_PREFETCH(p) __builtin_prefetch((const char*)(p) + 256, 1, 3)
#define RS_SHORT 4096u        /* ranges up to this length take the short-range form of a level (refsort_body.h: level_small) */
#define TASK_MIN (1u << 11)   /* buckets smaller than this are finished by the thread that made them */
#define RS_DIG_MIN ((size_t)RS_SHORT + 1) /* ranges at least this long below the top level also walk on a byte array of their digits (refsort_body.h) */

/* Element layouts.  wide: {key, index} (16 bytes).  packed: when the bits of the key's high word (bh), of its low word (bl) and of the
 * index (bi) fit into 64, one word (hi << bl | lo) << bi | index -- half the memory traffic of a walk that is bound by it.  The reference
 * takes its 8-bit digits from the ORIGINAL key hi << 32 | lo; RS_LEVEL maps such a digit to a position and mask inside the word. */
typedef struct { int bi, bl; uint64_t lomask; } rs_cfg_t;

typedef struct { void *a; size_t n; int shift; } rs_task_t;

/* Two stacks: a range of TASK_BIG elements or more starts with a walk that one thread does alone (tens of ms) before it has anything to share, so those are taken
 * first -- on one stack with their own children they sat at the bottom until every small task was gone, and the sort ended on a handful of lone walks. */
#define TASK_BIG ((size_t)1 << 20)
typedef struct rs_pool_s {
	pthread_mutex_t mu;
	pthread_cond_t cv;
	rs_task_t *q[2]; /* [0] small, [1] big */
	size_t nq[2], mq[2], elem;
	int busy, n_threads;
	rs_cfg_t cfg;
	void (*run)(struct rs_pool_s*, void*, size_t, int);
	double st_busy, st_max, st_big, st_t0, st_last_big; /* MA_REFSORT_TIMING: seconds inside tasks (all / the longest / the big ones), when the last big one ended */
	size_t st_n, st_nbig;
	/* The RESTRICTED sort (ma_refsort_packed_wanted).  The caller needs the reference's order only inside the hit groups of some reads (key's high word = read id):
	 * wcum[id] = wanted ids below id.  A level's walk is always done whole -- where an element lands depends on every element of the range -- but a bucket it leaves
	 * behind is only sorted further if a wanted read's hits are in it; the others stay as the walk left them (NOT sorted: the caller must not look there).
	 * top_start / top_base / top_shift: the packed words do not carry the top level's digit (refsort.c: ma_refsort_packed): it is the top-level bucket an element stands in. */
	const uint32_t *wcum;
	uint64_t n_ids;
	const size_t *top_start;
	const void *top_base;
	int top_shift;
} rs_pool_t;

static void pool_push(rs_pool_t *p, void *a, size_t n, int shift)
{
	const int w = n >= TASK_BIG;
	pthread_mutex_lock(&p->mu);
	if (p->nq[w] == p->mq[w]) {
		p->mq[w] = p->mq[w] ? p->mq[w] << 1 : 1024;
		p->q[w] = (rs_task_t*)realloc(p->q[w], p->mq[w] * sizeof(rs_task_t));
	}
	p->q[w][p->nq[w]].a = a, p->q[w][p->nq[w]].n = n, p->q[w][p->nq[w]].shift = shift;
	++p->nq[w];
	pthread_cond_signal(&p->cv);
	pthread_mutex_unlock(&p->mu);
}

/* The digit array of a range below the top level (refsort_body.h: level): a worker thread keeps ONE block for all the ranges it is handed (a range's array is dead
 * when its walk is over, and everything a range calls afterwards is shorter) -- 20 000 blocks of 50 KB allocated and freed by 64 threads at once was time spent in
 * the allocator's locks and in page faults, not in the sort.  Outside a worker thread (one thread sorts alone): malloc / free as before. */
static __thread const uint32_t *tl_wcum; /* the restriction of the call in progress on this thread (ma_refsort_packed_wanted; handed to the pool where one is set up) */
static __thread uint64_t tl_n_ids;
static __thread int tl_top_apart;        /* the call in progress sorts words that do not carry the top level's digit */
static __thread uint8_t *tl_dig;
static __thread size_t tl_dig_cap;
static __thread int tl_worker;
static uint8_t *rs_dig_get(size_t n)
{
	if (!tl_worker) return (uint8_t*)malloc(n + 16);
	if (tl_dig_cap < n + 16) {
		free(tl_dig);
		tl_dig_cap = n + 16 < ((size_t)1 << 16) ? (size_t)1 << 16 : n + 16;
		tl_dig = (uint8_t*)malloc(tl_dig_cap);
		if (tl_dig == 0) tl_dig_cap = 0;
	}
	return tl_dig;
}
static void rs_dig_put(uint8_t *d) { if (!tl_worker) free(d); }

static void *pool_worker(void *arg)
{
	rs_pool_t *p = (rs_pool_t*)arg;
	tl_worker = 1;
	pthread_mutex_lock(&p->mu);
	for (;;) {
		while (p->nq[0] + p->nq[1] == 0 && p->busy > 0) pthread_cond_wait(&p->cv, &p->mu);
		if (p->nq[0] + p->nq[1] == 0) break; /* nothing queued, nobody running: done */
		{
			const int w = p->nq[1] != 0;
			rs_task_t t = p->q[w][--p->nq[w]];
			const double ts = rs_timing > 0 ? rs_now() : 0;
			++p->busy;
			pthread_mutex_unlock(&p->mu);
			p->run(p, t.a, t.n, t.shift);
			pthread_mutex_lock(&p->mu);
			--p->busy;
			if (rs_timing > 0) {
				const double te = rs_now(), dt = te - ts;
				p->st_busy += dt; ++p->st_n;
				if (dt > p->st_max) p->st_max = dt;
				if (w) { p->st_big += dt; ++p->st_nbig; p->st_last_big = te - p->st_t0; }
			}
			if (p->busy == 0 && p->nq[0] + p->nq[1] == 0) pthread_cond_broadcast(&p->cv);
		}
	}
	pthread_mutex_unlock(&p->mu);
	free(tl_dig); tl_dig = 0; tl_dig_cap = 0; tl_worker = 0;
	return 0;
}

typedef struct { const void *a; const rs_cfg_t *cfg; size_t beg, end; int shift; uint64_t diff; size_t cnt[256]; uint8_t *dig; } sweep_t;

static uint64_t sweep_run(void *(*worker)(void*), const void *a, size_t n, int shift, size_t *cnt, const rs_cfg_t *cfg, int n_threads, uint8_t *dig)
{
	sweep_t *w;
	pthread_t th[RS_MAX_THREADS];
	uint64_t diff = 0;
	int t, k;
	if (n_threads > RS_MAX_THREADS) n_threads = RS_MAX_THREADS;
	if (n_threads < 1) n_threads = 1;
	w = (sweep_t*)malloc(sizeof(sweep_t) * n_threads);
	for (t = 0; t < n_threads; ++t) w[t].a = a, w[t].cfg = cfg, w[t].shift = shift, w[t].dig = dig, w[t].diff = 0, w[t].beg = n / n_threads * t, w[t].end = t == n_threads - 1 ? n : n / n_threads * (t + 1);
	for (t = 1; t < n_threads; ++t) pthread_create(&th[t], 0, worker, &w[t]);
	worker(&w[0]);
	for (t = 1; t < n_threads; ++t) pthread_join(th[t], 0);
	if (cnt) memset(cnt, 0, 256 * sizeof(size_t));
	for (t = 0; t < n_threads; ++t) {
		diff |= w[t].diff;
		if (cnt) for (k = 0; k < 256; ++k) cnt[k] += w[t].cnt[k];
	}
	free(w);
	return diff;
}

/* ---- wide elements ---- */
#define RS_T ma_ki_t
#define RS_NAME(x) wide_##x
#define RS_WORD(e) ((e).key)
#define RS_ORIG(e, cfg) ((e).key)
#define RS_CMPKEY(e, cfg) ((e).key)
#define RS_LEVEL(cfg, shift, sh, m) do { (void)(cfg); (sh) = (shift); (m) = 0xffu; } while (0)
#include "refsort_body.h"
#undef RS_T
#undef RS_NAME
#undef RS_WORD
#undef RS_ORIG
#undef RS_CMPKEY
#undef RS_LEVEL

/* ---- packed elements ---- */
#define RS_T uint64_t
#define RS_NAME(x) packed_##x
#define RS_WORD(e) (e)
#define RS_ORIG(e, cfg) (((e) >> ((cfg)->bl + (cfg)->bi)) << 32 | (((e) >> (cfg)->bi) & (cfg)->lomask))
#define RS_CMPKEY(e, cfg) ((e) >> (cfg)->bi)
/* digit at `shift` of hi << 32 | lo: the high word's digits sit above the low word's bl bits; a digit of the low word may be cut off by bl */
#define RS_LEVEL(cfg, shift, sh, m) do { \
		if ((shift) >= 32) { (sh) = (cfg)->bl + (cfg)->bi + ((shift) - 32), (m) = 0xffu; if ((sh) >= 64) (sh) = 0, (m) = 0u; /* a digit above the word's bits is 0 in every key */ } \
		else (sh) = (cfg)->bi + (shift), (m) = (shift) + 8 <= (cfg)->bl ? 0xffu : (shift) < (cfg)->bl ? (1u << ((cfg)->bl - (shift))) - 1u : 0u; \
	} while (0)
#include "refsort_body.h"
#undef RS_T
#undef RS_NAME
#undef RS_WORD
#undef RS_ORIG
#undef RS_CMPKEY
#undef RS_LEVEL

void ma_refsort_ki(ma_ki_t *a, size_t n, int n_threads)
{
	rs_cfg_t cfg = { 0, 32, 0xffffffffull };
	wide_sort(a, n, &cfg, n_threads);
}

/* perm[i] = input position of the record the reference's sort leaves at position i */
typedef struct { const uint64_t *keys; ma_ki_t *a; uint64_t *pk; uint32_t *perm; size_t beg, end; int phase; rs_cfg_t cfg; uint64_t mhi, mlo, diff, himask; uint8_t *dig; int shift_top; size_t cnt[256]; } fill_t;

static void *fill_worker(void *arg)
{
	fill_t *f = (fill_t*)arg;
	size_t i;
	if (f->phase == 0) { /* bounds of the two key words */
		uint64_t mhi = 0, mlo = 0;
		for (i = f->beg; i < f->end; ++i) { const uint64_t k = f->keys[i], hi = k >> 32, lo = k & 0xffffffffull; mhi = hi > mhi ? hi : mhi; mlo = lo > mlo ? lo : mlo; }
		f->mhi = mhi, f->mlo = mlo;
	} else if (f->phase == 1) for (i = f->beg; i < f->end; ++i) f->a[i].key = f->keys[i], f->a[i].idx = (uint32_t)i, f->a[i].pad = 0;
	else if (f->phase == 2) for (i = f->beg; i < f->end; ++i) f->perm[i] = f->a[i].idx;
	else if (f->phase == 3) for (i = f->beg; i < f->end; ++i) { const uint64_t k = f->keys[i]; f->pk[i] = ((k >> 32) << f->cfg.bl | (k & 0xffffffffull)) << f->cfg.bi | i; }
	else if (f->phase == 5) { uint64_t d = 0; const uint64_t k0 = f->keys[0]; for (i = f->beg; i < f->end; ++i) d |= f->keys[i] ^ k0; f->diff = d; } /* which bits vary */
	else if (f->phase == 6) { /* pack WITHOUT the bits the top level consumes (the bucket implies them); the top level's digits and their counts on the way */
		memset(f->cnt, 0, sizeof(f->cnt));
		for (i = f->beg; i < f->end; ++i) {
			const uint64_t k = f->keys[i];
			const unsigned d = (unsigned)(k >> f->shift_top) & 0xffu;
			f->dig[i] = (uint8_t)d; ++f->cnt[d];
			f->pk[i] = (((k >> 32) & f->himask) << f->cfg.bl | (k & 0xffffffffull)) << f->cfg.bi | i;
		}
	}
	else if (f->phase == 7) { memset(f->cnt, 0, sizeof(f->cnt)); for (i = f->beg; i < f->end; ++i) ++f->cnt[f->dig[i]]; }
	else { const uint64_t im = f->cfg.bi >= 64 ? ~0ull : (1ull << f->cfg.bi) - 1; for (i = f->beg; i < f->end; ++i) f->perm[i] = (uint32_t)(f->pk[i] & im); }
	return 0;
}

static void fill_run(fill_t *proto, size_t n, int phase, int n_threads)
{
	fill_t one, *heap = (fill_t*)malloc(sizeof(fill_t) * RS_MAX_THREADS), *f = heap ? heap : &one; /* (2 KB each; without the block: one thread, its state on the stack) */
	pthread_t th[RS_MAX_THREADS];
	int t;
	if (n_threads > RS_MAX_THREADS) n_threads = RS_MAX_THREADS;
	if (n < (1u << 20) || n_threads < 2 || heap == 0) n_threads = 1;
	for (t = 0; t < n_threads; ++t) {
		f[t] = *proto; f[t].phase = phase;
		f[t].beg = n / n_threads * t, f[t].end = t == n_threads - 1 ? n : n / n_threads * (t + 1);
	}
	for (t = 1; t < n_threads; ++t) pthread_create(&th[t], 0, fill_worker, &f[t]);
	fill_worker(&f[0]);
	for (t = 1; t < n_threads; ++t) pthread_join(th[t], 0);
	if (phase == 0) { proto->mhi = proto->mlo = 0; for (t = 0; t < n_threads; ++t) { if (f[t].mhi > proto->mhi) proto->mhi = f[t].mhi; if (f[t].mlo > proto->mlo) proto->mlo = f[t].mlo; } }
	if (phase == 5) { proto->diff = 0; for (t = 0; t < n_threads; ++t) proto->diff |= f[t].diff; }
	if (phase == 6 || phase == 7) { int k; memset(proto->cnt, 0, sizeof(proto->cnt)); for (t = 0; t < n_threads; ++t) for (k = 0; k < 256; ++k) proto->cnt[k] += f[t].cnt[k]; }
	free(heap);
}

static int bits_of64(uint64_t x) { int b = 0; while (x) ++b, x >>= 1; return b; }

/* the buckets below the top level are independent and memory-latency bound: they scale past the 16 threads the text reader settles on */
static int refsort_threads(void)
{
	const char *s = getenv("MA_THREADS");
	long n = s ? atol(s) : ma_cpu_budget(); /* (what the control group allows, not what the machine has: ingest_mt.c) */
	if (!s && n > 64) n = 64;
	return n < 1 ? 1 : n > RS_MAX_THREADS ? RS_MAX_THREADS : (int)n;
}

int ma_refsort_perm(const uint64_t *keys, size_t n, uint32_t *perm)
{
	const int nt = refsort_threads();
	fill_t f;
	int bh, bl, bi;
	if (n == 0) return 0;
	memset(&f, 0, sizeof(f));
	f.keys = keys; f.perm = perm;
	RS_T0;
	fill_run(&f, n, 0, nt);
	RS_LAP("bounds");
	bh = bits_of64(f.mhi); bl = bits_of64(f.mlo); bi = bits_of64(n - 1);
	if (bl == 0) bl = 1;
	if (bi == 0) bi = 1;
	if (bh + bl + bi <= 64 && !getenv("MA_REFSORT_WIDE")) { /* one word per element */
		f.cfg.bi = bi; f.cfg.bl = bl; f.cfg.lomask = (1ull << bl) - 1;
		f.pk = (uint64_t*)malloc(n * sizeof(uint64_t));
		if (f.pk == 0) return -1;
		fill_run(&f, n, 3, nt);
		RS_LAP("pack");
		packed_sort(f.pk, n, &f.cfg, nt);
		RS_LAP("sort");
		fill_run(&f, n, 4, nt);
		RS_LAP("unpack");
		free(f.pk);
		return 0;
	}
	/* Too wide for one word -- BASELINE configs[4]: 23 id + 14 start + 30 index bits.  Below the top level every element of a bucket has the same top digit,
	 * so those bits need not travel: if the rest fits, the top level's digits are taken from the raw keys and the elements are packed without them. */
	if (nt > 1 && n >= (1u << 17) && !getenv("MA_REFSORT_WIDE")) {
		int shift = 56;
		fill_run(&f, n, 5, nt);
		while (shift > 0 && (f.diff >> shift & 0xff) == 0) shift -= 8;
		if (f.diff != 0 && shift >= 32 && (shift - 32) + bl + bi <= 64) {
			f.shift_top = shift; f.himask = shift > 32 ? (1ull << (shift - 32)) - 1 : 0;
			f.cfg.bi = bi; f.cfg.bl = bl; f.cfg.lomask = (1ull << bl) - 1;
			f.pk = (uint64_t*)malloc(n * sizeof(uint64_t));
			f.dig = (uint8_t*)malloc(n + 16);
			if (f.pk == 0 || f.dig == 0) { free(f.pk); free(f.dig); return -1; }
			memset(f.dig + n, 0, 16);
			fill_run(&f, n, 6, nt);
			RS_LAP("pack (top digit apart)");
			packed_sort_from_top(f.pk, n, &f.cfg, nt, f.cnt, f.dig, shift);
			RS_LAP("sort");
			free(f.dig);
			fill_run(&f, n, 4, nt);
			RS_LAP("unpack");
			free(f.pk);
			return 0;
		}
	}
	f.a = (ma_ki_t*)malloc(n * sizeof(ma_ki_t));
	if (f.a == 0) return -1;
	fill_run(&f, n, 1, nt);
	ma_refsort_ki(f.a, n, nt);
	fill_run(&f, n, 2, nt);
	free(f.a);
	return 0;
}

/* The same sort on elements the CALLER packed (the device does it on the way out, csrc/radix.hip): pk[i] = (hi << bl | lo) << bi | i, with hi = the key's high
 * word, lo its low word (lo < 2^bl), i < 2^bi the input position; sorted in place, so that afterwards pk[j] & (2^bi - 1) is the input position of the j-th
 * record in the reference's order.  shift_top >= 32: keys too wide for that -- hi is stored WITHOUT its bits at and above (shift_top - 32), dig_top[i] (n + 16
 * bytes, the last 16 zero) is the key's digit at shift_top, which is the first level that can move anything (every key is < 2^(shift_top + 8)); the caller
 * guarantees (shift_top - 32) + bl + bi <= 64.  shift_top < 0: the whole key is in the word (bits of hi + bl + bi <= 64). */
int ma_refsort_packed(uint64_t *pk, size_t n, int bl, int bi, int shift_top, const uint8_t *dig_top)
{
	const int nt = refsort_threads();
	rs_cfg_t cfg;
	if (n < 2) return 0;
	if (bl < 1 || bl > 32 || bi < 1 || bi > 32) return -1;
	cfg.bi = bi; cfg.bl = bl; cfg.lomask = (1ull << bl) - 1;
	{
		RS_T0;
		if (shift_top < 0) packed_sort(pk, n, &cfg, nt);
		else {
			fill_t f;
			if (shift_top < 32 || (shift_top & 7) || (shift_top - 32) + bl + bi > 64 || dig_top == 0 || n <= RS_SMALL) return -1; /* (<= 64 records: ksort.h:182 sorts by insertion -- the caller uses ma_refsort_perm) */
			tl_top_apart = 1;
			if (nt > 1 && n >= (1u << 17)) {
				memset(&f, 0, sizeof(f));
				f.dig = (uint8_t*)dig_top;
				fill_run(&f, n, 7, nt); /* the counts of the top level's digits */
				packed_sort_from_top(pk, n, &cfg, nt, f.cnt, dig_top, shift_top);
			} else { /* small: one thread does everything; the top level from its digit array like any other */
				size_t cnt[256], i;
				memset(cnt, 0, sizeof(cnt));
				for (i = 0; i < n; ++i) ++cnt[dig_top[i]];
				packed_sort_from_top(pk, n, &cfg, 1, cnt, dig_top, shift_top);
			}
			tl_top_apart = 0;
		}
		RS_LAP("sort");
	}
	return 0;
}

/* ma_refsort_packed, but the order is only wanted inside the hit groups of the reads with wcum[id + 1] != wcum[id] (wcum: n_ids + 1 counts, wcum[0] = 0; the key's
 * high word is the read id).  Afterwards pk[] holds the reference's order at the positions of THOSE reads' hits; everywhere else it is a permutation of the input that
 * is sorted down to some level only.  Every read's hits still stand in the stretch they occupy in any sorted order if a wanted read shares the stretch's bucket at the
 * levels above the read id -- the caller takes the positions of a wanted read's stretch from a sorted copy of its own. */
int ma_refsort_packed_wanted(uint64_t *pk, size_t n, int bl, int bi, int shift_top, const uint8_t *dig_top, const uint32_t *wcum, uint64_t n_ids)
{
	int rc;
	tl_wcum = wcum; tl_n_ids = n_ids;
	rc = ma_refsort_packed(pk, n, bl, bi, shift_top, dig_top);
	tl_wcum = 0; tl_n_ids = 0;
	return rc;
}
