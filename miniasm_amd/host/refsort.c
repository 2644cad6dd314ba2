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

#define RS_SMALL 64           /* RS_MIN_SIZE ksort.h:132 */
#define TASK_MIN (1u << 11)   /* buckets smaller than this are finished by the thread that made them */

static void ki_insertion(ma_ki_t *a, size_t n) /* ksort.h:142-152: stable insertion sort on the whole key */
{
	size_t i, j;
	for (i = 1; i < n; ++i) {
		if (a[i].key < a[i-1].key) {
			ma_ki_t t = a[i];
			for (j = i; j > 0 && t.key < a[j-1].key; --j) a[j] = a[j-1];
			a[j] = t;
		}
	}
}

typedef struct { ma_ki_t *a; size_t n; int shift; } rs_task_t;

typedef struct {
	pthread_mutex_t mu;
	pthread_cond_t cv;
	rs_task_t *q;
	size_t nq, mq;
	int busy, n_threads;
} rs_pool_t;

static void pool_push(rs_pool_t *p, ma_ki_t *a, size_t n, int shift)
{
	pthread_mutex_lock(&p->mu);
	if (p->nq == p->mq) {
		p->mq = p->mq ? p->mq << 1 : 1024;
		p->q = (rs_task_t*)realloc(p->q, p->mq * sizeof(rs_task_t));
	}
	p->q[p->nq].a = a, p->q[p->nq].n = n, p->q[p->nq].shift = shift;
	++p->nq;
	pthread_cond_signal(&p->cv);
	pthread_mutex_unlock(&p->mu);
}

/* one level of ksort.h:153-179 on a[0..n); pool == NULL: finish everything below on this thread */
static void ki_permute(rs_pool_t *pool, ma_ki_t *a, size_t *tail, int shift);

static void ki_level(rs_pool_t *pool, ma_ki_t *a, size_t n, int shift)
{
	size_t tail[256], i;
	/* A level on which the digit does not vary leaves the range untouched and recurses into the same range (n > 64
	 * here).  One sweep gives the varying bits and, optimistically, the histogram of the current digit. */
	for (;;) {
		uint64_t diff = 0, k0 = a[0].key;
		memset(tail, 0, sizeof(tail));
		for (i = 0; i < n; ++i) diff |= a[i].key ^ k0, ++tail[a[i].key >> shift & 0xff];
		if (diff == 0) return; /* all keys equal: every remaining level is the identity */
		if ((diff >> shift & 0xff) != 0) break;
		while (shift > 0 && (diff >> shift & 0xff) == 0) shift -= 8;
	}
	ki_permute(pool, a, tail, shift);
}

static void *pool_worker(void *arg)
{
	rs_pool_t *p = (rs_pool_t*)arg;
	pthread_mutex_lock(&p->mu);
	for (;;) {
		while (p->nq == 0 && p->busy > 0) pthread_cond_wait(&p->cv, &p->mu);
		if (p->nq == 0) break; /* nothing queued, nobody running: done */
		{
			rs_task_t t = p->q[--p->nq];
			++p->busy;
			pthread_mutex_unlock(&p->mu);
			ki_level(p, t.a, t.n, t.shift);
			pthread_mutex_lock(&p->mu);
			--p->busy;
			if (p->busy == 0 && p->nq == 0) pthread_cond_broadcast(&p->cv);
		}
	}
	pthread_mutex_unlock(&p->mu);
	return 0;
}

/* parallel sweeps over a big range: OR of (key ^ key[0]) and, with shift >= 0, the histogram of one digit */
typedef struct { const ma_ki_t *a; size_t beg, end; int shift; uint64_t diff; size_t cnt[256]; } sweep_t;

static void *sweep_worker(void *arg)
{
	sweep_t *w = (sweep_t*)arg;
	const ma_ki_t *a = w->a;
	const uint64_t k0 = a[0].key;
	uint64_t diff = 0;
	size_t i;
	memset(w->cnt, 0, sizeof(w->cnt));
	if (w->shift >= 0) for (i = w->beg; i < w->end; ++i) diff |= a[i].key ^ k0, ++w->cnt[a[i].key >> w->shift & 0xff];
	else for (i = w->beg; i < w->end; ++i) diff |= a[i].key ^ k0;
	w->diff = diff;
	return 0;
}

static uint64_t sweep_run(const ma_ki_t *a, size_t n, int shift, size_t *cnt, int n_threads)
{
	sweep_t w[64];
	pthread_t th[64];
	uint64_t diff = 0;
	int t, k;
	if (n_threads > 64) n_threads = 64;
	if (n_threads < 1) n_threads = 1;
	for (t = 0; t < n_threads; ++t) w[t].a = a, w[t].shift = shift, w[t].beg = n / n_threads * t, w[t].end = t == n_threads - 1 ? n : n / n_threads * (t + 1);
	for (t = 1; t < n_threads; ++t) pthread_create(&th[t], 0, sweep_worker, &w[t]);
	sweep_worker(&w[0]);
	for (t = 1; t < n_threads; ++t) pthread_join(th[t], 0);
	if (cnt) memset(cnt, 0, 256 * sizeof(size_t));
	for (t = 0; t < n_threads; ++t) {
		diff |= w[t].diff;
		if (cnt) for (k = 0; k < 256; ++k) cnt[k] += w[t].cnt[k];
	}
	return diff;
}

/* the cycle-leader permutation of one level (ksort.h:153-176) given the digit counts; then the buckets below */
static void ki_permute(rs_pool_t *pool, ma_ki_t *a, size_t *tail, int shift)
{
	size_t head[256], start[257];
	int k;
	start[0] = 0;
	for (k = 0; k < 256; ++k) start[k + 1] = start[k] + tail[k], head[k] = start[k], tail[k] = start[k + 1];
	for (k = 0; k < 256;) {
		int dst;
		if (head[k] == tail[k]) { ++k; continue; }
		dst = (int)(a[head[k]].key >> shift & 0xff);
		if (dst == k) { ++head[k]; continue; }
		{
			ma_ki_t carry = a[head[k]];
			do {
				ma_ki_t evicted = a[head[dst]];
				a[head[dst]++] = carry;
				carry = evicted;
				dst = (int)(carry.key >> shift & 0xff);
			} while (dst != k);
			a[head[k]++] = carry;
		}
	}
	if (shift) {
		int next = shift > 8 ? shift - 8 : 0;
		for (k = 0; k < 256; ++k) {
			size_t m = start[k + 1] - start[k];
			if (m > RS_SMALL) {
				if (pool && m >= TASK_MIN) pool_push(pool, a + start[k], m, next);
				else ki_level(pool, a + start[k], m, next);
			} else if (m > 1) ki_insertion(a + start[k], m);
		}
	}
}

void ma_refsort_ki(ma_ki_t *a, size_t n, int n_threads)
{
	if (n <= RS_SMALL) { ki_insertion(a, n); return; } /* ksort.h:182 */
	if (n_threads <= 1 || n < (1u << 17)) { ki_level(0, a, n, 56); return; }
	{
		rs_pool_t p;
		pthread_t *th;
		size_t cnt[256];
		int t, shift = 56;
		uint64_t diff;
		memset(&p, 0, sizeof(p));
		pthread_mutex_init(&p.mu, 0);
		pthread_cond_init(&p.cv, 0);
		if (n_threads > 64) n_threads = 64;
		p.n_threads = n_threads;
		/* the top level: its two sweeps (which bits vary; the digit counts) run on all threads, only the walk itself is sequential */
		diff = sweep_run(a, n, -1, 0, n_threads);
		if (diff == 0) { pthread_mutex_destroy(&p.mu); pthread_cond_destroy(&p.cv); return; } /* all keys equal: every level is the identity */
		while (shift > 0 && (diff >> shift & 0xff) == 0) shift -= 8;
		sweep_run(a, n, shift, cnt, n_threads);
		th = (pthread_t*)malloc(sizeof(pthread_t) * n_threads);
		++p.busy; /* the top-level walk below produces tasks: workers must not leave while it runs */
		for (t = 0; t < n_threads; ++t) pthread_create(&th[t], 0, pool_worker, &p);
		ki_permute(&p, a, cnt, shift);
		pthread_mutex_lock(&p.mu);
		--p.busy;
		pthread_cond_broadcast(&p.cv);
		pthread_mutex_unlock(&p.mu);
		for (t = 0; t < n_threads; ++t) pthread_join(th[t], 0);
		free(th); free(p.q);
		pthread_mutex_destroy(&p.mu);
		pthread_cond_destroy(&p.cv);
	}
}

/* perm[i] = input position of the record the reference's sort leaves at position i */
typedef struct { const uint64_t *keys; ma_ki_t *a; uint32_t *perm; size_t beg, end; int phase; } fill_t;

static void *fill_worker(void *arg)
{
	fill_t *f = (fill_t*)arg;
	size_t i;
	if (f->phase == 0) for (i = f->beg; i < f->end; ++i) f->a[i].key = f->keys[i], f->a[i].idx = (uint32_t)i, f->a[i].pad = 0;
	else for (i = f->beg; i < f->end; ++i) f->perm[i] = f->a[i].idx;
	return 0;
}

static void fill_run(const uint64_t *keys, ma_ki_t *a, uint32_t *perm, size_t n, int phase, int n_threads)
{
	fill_t f[64];
	pthread_t th[64];
	int t;
	if (n_threads > 64) n_threads = 64;
	if (n < (1u << 20) || n_threads < 2) n_threads = 1;
	for (t = 0; t < n_threads; ++t) {
		f[t].keys = keys, f[t].a = a, f[t].perm = perm, f[t].phase = phase;
		f[t].beg = n / n_threads * t, f[t].end = t == n_threads - 1 ? n : n / n_threads * (t + 1);
	}
	if (n_threads == 1) { fill_worker(&f[0]); return; }
	for (t = 0; t < n_threads; ++t) pthread_create(&th[t], 0, fill_worker, &f[t]);
	for (t = 0; t < n_threads; ++t) pthread_join(th[t], 0);
}

int ma_refsort_perm(const uint64_t *keys, size_t n, uint32_t *perm)
{
	ma_ki_t *a;
	const int nt = ma_ingest_threads();
	if (n == 0) return 0;
	a = (ma_ki_t*)malloc(n * sizeof(ma_ki_t));
	if (a == 0) return -1;
	fill_run(keys, a, perm, n, 0, nt);
	ma_refsort_ki(a, n, nt);
	fill_run(keys, a, perm, n, 1, nt);
	free(a);
	return 0;
}
