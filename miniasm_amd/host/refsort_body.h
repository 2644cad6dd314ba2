/* refsort_body.h -- the walk of host/refsort.c for one element layout; included twice (wide: {key, index} records, packed: one 64-bit
 * word holding the squeezed key above the index).  Parameters: RS_T element type, RS_NAME(x) name mangling, RS_ORIG(e, cfg) the
 * original 64-bit key of an element, RS_CMPKEY(e, cfg) something ordered like it (the insertion sort compares nothing else),
 * RS_LEVEL(cfg, shift, sh, m) the position and mask of the reference's 8-bit digit at `shift` inside the element's word RS_WORD(e). */

static void RS_NAME(insertion)(RS_T *a, size_t n, const rs_cfg_t *cfg) /* ksort.h:142-152: stable insertion sort on the whole key */
{
	size_t i, j;
	(void)cfg;
	for (i = 1; i < n; ++i) {
		if (RS_CMPKEY(a[i], cfg) < RS_CMPKEY(a[i-1], cfg)) {
			RS_T t = a[i];
			const uint64_t tk = RS_CMPKEY(t, cfg);
			for (j = i; j > 0 && tk < RS_CMPKEY(a[j-1], cfg); --j) a[j] = a[j-1];
			a[j] = t;
		}
	}
}

static void RS_NAME(level)(rs_pool_t *pool, RS_T *a, size_t n, int shift);

/* restricted sort (rs_pool_t): does the range *e stands in -- the elements that agree with it on the key's bits from P up -- hold a hit of a wanted read? */
static inline int RS_NAME(wanted)(const rs_pool_t *pool, const RS_T *e, int P)
{
	uint64_t key, id, lo, hi;
	if (!pool->wcum) return 1;
	key = RS_ORIG(*e, &pool->cfg);
	if (pool->top_start) { /* the top digit = the top-level bucket the element stands in */
		const size_t at = (size_t)(e - (const RS_T*)pool->top_base);
		int b0 = 0, b1 = 256;
		while (b1 - b0 > 1) { const int mid = (b0 + b1) >> 1; if (pool->top_start[mid] <= at) b0 = mid; else b1 = mid; }
		key |= (uint64_t)b0 << pool->top_shift;
	}
	id = key >> 32;
	if (P <= 32) lo = id, hi = id + 1;
	else { const int w = P - 32; lo = w >= 32 ? 0 : id >> w << w; hi = w >= 32 ? pool->n_ids : lo + (1ull << w); }
	if (lo >= pool->n_ids) return 0;
	if (hi > pool->n_ids) hi = pool->n_ids;
	return pool->wcum[hi] != pool->wcum[lo];
}

/* the buckets a level leaves behind (ksort.h:177-182): radix again when larger than 64, else the stable insertion sort */
static void RS_NAME(dispatch)(rs_pool_t *pool, RS_T *a, const size_t *start, int shift)
{
	const rs_cfg_t *cfg = &pool->cfg;
	int k, next = shift > 8 ? shift - 8 : 0;
	if (!shift) return;
	for (k = 0; k < 256; ++k) {
		size_t cnt = start[k + 1] - start[k];
		if (cnt > 1 && !RS_NAME(wanted)(pool, a + start[k], shift)) continue;
		if (cnt > RS_SMALL) {
			if (pool->n_threads > 1 && cnt >= TASK_MIN) pool_push(pool, a + start[k], cnt, next);
			else RS_NAME(level)(pool, a + start[k], cnt, next);
		} else if (cnt > 1) RS_NAME(insertion)(a + start[k], cnt, cfg);
	}
}

/* The same permutation for the top level of a big input, where the walk is the one thing no other thread can help with.  The walk only ever
 * evicts ORIGINAL elements -- a slot at or behind a bucket's head has not been written yet -- so where it goes next is a function of the
 * input's digits alone: dig[] (one byte per element, written by the parallel sweep that counts the digits) replaces the look into the evicted
 * element, and nd[d] = digit of the element at bucket d's head is kept ready per bucket.  The walk's dependent chain is then ONE load per
 * step (d -> nd[d]); the element moves themselves hang off the bucket heads, not off each other, and overlap.
 * Round 3 tried the walk on the digits ALONE -- a permutation written down with runs of home elements taken at once, the elements moved afterwards
 * by all threads: 0.62 s against 0.32 s for this form on the 50 M-overlap noisy input, 8.0 s against 2.8 s at BASELINE configs[4] on the EPYC of the
 * GPU box (profiles/r03_tiewalk.txt): noisy PAFs have few long runs, and a separate 4-byte permutation plus a gather is more memory traffic than
 * moving 8-byte elements once.  Dropped. */
/* The walk as ONE loop without a data-dependent branch.  ksort.h:160-172 is a token that moves from bucket to bucket: at bucket d it takes the element at
 * d's head (that element's digit says where the token goes next) and leaves the element it carries.  For every bucket but the scanned one k the slot read
 * and the slot written are the same; for k the element carried away at the start of a chain came from the slot the chain's last element will fill, so k
 * reads one slot ahead of where it writes (an element of k that is already home is written back to its own slot).  With a read and a write position per
 * bucket every step is the same few instructions, `while (l != k)` and `if (l != k)` of the reference disappear, and the only branch left is taken when
 * the scanned bucket is full.  Used for the short ranges (level_small), where it saves the set-up of the other form.  For the long walks it was measured
 * against permute_top below on the EPYC 9575F of the GPU box and lost on real inputs (0.34-0.36 s against 0.31 s at the top level of the 50 M-overlap noisy
 * input, 46 against 37 ms on the tie-rich one) although it wins on random digits (tools/probes/walk_probe.c: 2.4 against 3.2 ns per element with 4 buckets,
 * 1.9 against 2.0 with 77): a PAF lists a query's overlaps together, so every other element of a stretch goes to the same bucket, and the literal form's
 * inner loop keeps that bucket's state in flight (profiles/r03_tiewalk.txt). */
typedef struct { size_t r, w; } RS_NAME(rw_t);
static void RS_NAME(permute_uniform)(RS_T *a, const size_t *start, const uint8_t *dig /* n + 1 bytes */, int lo, int hi /* the digits that occur: start[lo .. hi + 1] are set */)
{
	RS_NAME(rw_t) b[256];
	uint8_t nd[256];
	const size_t last = start[hi + 1] - 1;
	size_t left;
	RS_T carry;
	unsigned d;
	int k;
	for (k = lo; k <= hi; ++k) b[k].r = b[k].w = start[k], nd[k] = dig[start[k]];
	for (k = lo; k <= hi && start[k + 1] == start[k]; ++k) {}
	if (k > hi) return;
	left = start[k + 1] - b[k].w; carry = a[b[k].w]; d = nd[k]; b[k].r = b[k].w + 1; nd[k] = dig[b[k].r];
	for (;;) { /* carry = the element taken from the scanned bucket's write slot or evicted on the way; d = its digit */
		const size_t rd = b[d].r, wd = b[d].w;
		const unsigned dn = nd[d];
		const RS_T evicted = a[rd < last ? rd : last]; /* (the scanned bucket reads one slot ahead: at most one past the range, and that element is dropped) */
		a[wd] = carry;
		b[d].r = rd + 1; b[d].w = wd + 1; nd[d] = dig[rd + 1];
		__builtin_prefetch(dig + rd + 129); /* the digit has to be in L1 when this bucket comes round again: on the EPYC of the GPU box 2.6 -> 1.9 ns per element with 77 buckets (tools/probes/walk_probe.c) */
		RS_PREFETCH(&a[rd]);
		carry = evicted;
		left -= d == (unsigned)k;
		d = dn;
		if (left == 0) { /* bucket k is full (what was read beyond its end is dropped): scan the next bucket that has free slots */
			do ++k; while (k <= hi && b[k].w == start[k + 1]);
			if (k > hi) break;
			left = start[k + 1] - b[k].w; carry = a[b[k].w]; d = nd[k]; b[k].r = b[k].w + 1; nd[k] = dig[b[k].r];
		}
	}
}

typedef struct { size_t head; uint32_t nd; uint32_t pad; } RS_NAME(bk_t);
static void RS_NAME(permute_top)(rs_pool_t *pool, RS_T *a, const size_t *cnt, const uint8_t *dig /* n + 1 bytes */, int shift)
{
	RS_NAME(bk_t) b[256];
	size_t start[257];
	/* A bucket the walk has left is final -- it is full, so no element that belongs there is still on its way, and the walk writes to the heads of buckets
	 * that are not full only.  With worker threads around it is handed over THEN (ksort.h:177-182 does not care when: the ranges are disjoint), and the buckets
	 * below the top level are sorted beside the walk instead of behind it (BASELINE configs[4]: top walk 1.68 s, then buckets 1.23 s on 64 threads).  The
	 * buckets too small for a task of their own wait in `later` for this thread. */
	const int early = pool->n_threads > 1 && shift != 0 && !rs_late(), next = shift > 8 ? shift - 8 : 0; /* (MA_REFSORT_LATE=1: the A/B switch) */
	uint8_t later[256];
	int n_later = 0;
	int k;
	start[0] = 0;
	for (k = 0; k < 256; ++k) start[k + 1] = start[k] + cnt[k];
	for (k = 0; k < 256; ++k) b[k].head = start[k], b[k].nd = dig[start[k]], b[k].pad = 0;
	for (k = 0; k < 256;) {
		unsigned d;
		if (b[k].head == start[k + 1]) {
			if (early) {
				const size_t c = start[k + 1] - start[k];
				if (c > 1 && !RS_NAME(wanted)(pool, a + start[k], shift)) {}
				else if (c >= TASK_MIN) pool_push(pool, a + start[k], c, next);
				else if (c > 1) later[n_later++] = (uint8_t)k;
			}
			++k;
			continue;
		}
		d = b[k].nd;
		if (d == (unsigned)k) { /* already home -- and so, in a PAF-ordered input, are most of its neighbours: the whole stretch at once, eight digits per step */
			size_t p = b[k].head + 1;
			const size_t lim = start[k + 1];
			const uint64_t pat = 0x0101010101010101ull * (unsigned)k;
			while (p + 8 <= lim) {
				uint64_t x;
				memcpy(&x, dig + p, 8);
				x ^= pat;
				if (x) { p += (size_t)__builtin_ctzll(x) >> 3; goto stretch_done; }
				p += 8;
			}
			while (p < lim && dig[p] == (unsigned)k) ++p;
		stretch_done:
			b[k].head = p; b[k].nd = dig[p];
			continue;
		}
		{
			RS_T carry = a[b[k].head];
			do {
				const size_t slot = b[d].head;
				const unsigned dn = b[d].nd;
				const RS_T evicted = a[slot];
				b[d].head = slot + 1;
				b[d].nd = dig[slot + 1];
				__builtin_prefetch(dig + slot + 129); /* (2.37 -> 2.17 ns per element with 16 buckets on the GPU box's EPYC, tools/probes/walk_probe.c forms 0 / 6) */
				a[slot] = carry;
				RS_PREFETCH(&a[slot]);
				carry = evicted;
				d = dn;
			} while (d != (unsigned)k);
			a[b[k].head] = carry;
			b[k].nd = dig[++b[k].head];
		}
	}
	if (!early) { RS_NAME(dispatch)(pool, a, start, shift); return; }
	for (k = 0; k < n_later; ++k) {
		const size_t c = start[later[k] + 1] - start[later[k]];
		if (c > RS_SMALL) RS_NAME(level)(pool, a + start[later[k]], c, next);
		else RS_NAME(insertion)(a + start[later[k]], c, &pool->cfg);
	}
}

/* the cycle-leader permutation of one level (ksort.h:153-176) given the digit counts in tail[]; then the buckets below */
static void RS_NAME(permute)(rs_pool_t *pool, RS_T *a, size_t *tail, int shift)
{
	const rs_cfg_t *cfg = &pool->cfg;
	size_t head[256], start[257];
	int k, sh;
	unsigned m;
	RS_LEVEL(cfg, shift, sh, m);
	start[0] = 0;
	for (k = 0; k < 256; ++k) start[k + 1] = start[k] + tail[k], head[k] = start[k], tail[k] = start[k + 1];
	for (k = 0; k < 256;) {
		int dst;
		if (head[k] == tail[k]) { ++k; continue; }
		dst = (int)(RS_WORD(a[head[k]]) >> sh & m);
		if (dst == k) { ++head[k]; continue; }
		{
			RS_T carry = a[head[k]];
			do {
				RS_T evicted = a[head[dst]];
				a[head[dst]++] = carry;
				RS_PREFETCH(&a[head[dst]]);
				carry = evicted;
				dst = (int)(RS_WORD(carry) >> sh & m);
			} while (dst != k);
			a[head[k]++] = carry;
		}
	}
	RS_NAME(dispatch)(pool, a, start, shift);
}

/* A short range -- the hits of one read, the arcs of one vertex: most of the ranges there are, a hundred elements each.  Same procedure, but what a
 * level costs must not be the 256 buckets it could have: one pass says which bits vary (and with that which level is the next that moves anything
 * and which digits can occur there: lo | any subset of the varying bits), and counting, walk and hand-over only look at the digits lo..hi. */
static void RS_NAME(level_small)(rs_pool_t *pool, RS_T *a, size_t n, int shift)
{
	const rs_cfg_t *cfg = &pool->cfg;
	size_t start[257], cnt[256];
	uint8_t dig[RS_SHORT + 16];
	const uint64_t k0 = RS_ORIG(a[0], cfg);
	uint64_t diff = 0;
	size_t i;
	int sh, k, lo, hi, next;
	unsigned m, vb;
	for (i = 1; i < n; ++i) diff |= RS_ORIG(a[i], cfg) ^ k0;
	if (diff == 0) return; /* all keys equal: every remaining level is the identity */
	while (shift > 0 && (diff >> shift & 0xff) == 0) shift -= 8; /* levels on which the digit does not vary leave the range as it is */
	RS_LEVEL(cfg, shift, sh, m);
	vb = (unsigned)(diff >> shift & 0xff) & m;
	lo = (int)((unsigned)(RS_WORD(a[0]) >> sh & m) & ~vb); hi = lo | (int)vb;
	for (k = lo; k <= hi; ++k) cnt[k] = 0;
	for (i = 0; i < n; ++i) { const unsigned dg = (unsigned)(RS_WORD(a[i]) >> sh & m); dig[i] = (uint8_t)dg; ++cnt[dg]; }
	dig[n] = dig[n + 1] = 0;
	start[lo] = 0;
	for (k = lo; k <= hi; ++k) start[k + 1] = start[k] + cnt[k];
	RS_NAME(permute_uniform)(a, start, dig, lo, hi); /* ksort.h:160-172 */
	if (!shift) return;
	next = shift > 8 ? shift - 8 : 0;
	for (k = lo; k <= hi; ++k) { /* ksort.h:177-182 */
		const size_t c = cnt[k];
		if (c > 1 && pool->wcum && !RS_NAME(wanted)(pool, a + start[k], shift)) continue;
		if (c > RS_SMALL) RS_NAME(level_small)(pool, a + start[k], c, next);
		else if (c > 1) RS_NAME(insertion)(a + start[k], c, cfg);
	}
}

/* one level of ksort.h:153-179 on a[0..n) */
static void RS_NAME(level)(rs_pool_t *pool, RS_T *a, size_t n, int shift)
{
	const rs_cfg_t *cfg = &pool->cfg;
	size_t tail[256], i;
	if (n <= RS_SHORT) { RS_NAME(level_small)(pool, a, n, shift); return; }
	/* A level on which the digit does not vary leaves the range untouched and recurses into the same range (n > 64
	 * here).  One sweep gives the varying bits and, optimistically, the histogram of the current digit. */
	uint8_t *dig = n >= RS_DIG_MIN ? rs_dig_get(n) : 0; /* a big range (the 65 536-read buckets below the top level of a 10^9-hit input): the digit walk */
	for (;;) {
		uint64_t diff = 0;
		const uint64_t k0 = RS_ORIG(a[0], cfg);
		int sh;
		unsigned m;
		RS_LEVEL(cfg, shift, sh, m);
		memset(tail, 0, sizeof(tail));
		if (dig) for (i = 0; i < n; ++i) { const unsigned dg = (unsigned)(RS_WORD(a[i]) >> sh & m); diff |= RS_ORIG(a[i], cfg) ^ k0; dig[i] = (uint8_t)dg; ++tail[dg]; }
		else for (i = 0; i < n; ++i) diff |= RS_ORIG(a[i], cfg) ^ k0, ++tail[RS_WORD(a[i]) >> sh & m];
		if (diff == 0) { if (dig) rs_dig_put(dig); return; } /* all keys equal: every remaining level is the identity */
		if ((diff >> shift & 0xff) != 0) break;
		while (shift > 0 && (diff >> shift & 0xff) == 0) shift -= 8;
	}
	if (dig) { memset(dig + n, 0, 16); RS_NAME(permute_top)(pool, a, tail, dig, shift); rs_dig_put(dig); }
	else RS_NAME(permute)(pool, a, tail, shift);
}

/* parallel sweeps over a big range: OR of (key ^ key[0]) and, with shift >= 0, the histogram of one digit */
static void *RS_NAME(sweep_worker)(void *arg)
{
	sweep_t *w = (sweep_t*)arg;
	const RS_T *a = (const RS_T*)w->a;
	const rs_cfg_t *cfg = w->cfg;
	const uint64_t k0 = RS_ORIG(a[0], cfg);
	uint64_t diff = 0;
	size_t i;
	memset(w->cnt, 0, sizeof(w->cnt));
	if (w->shift >= 0) {
		int sh;
		unsigned m;
		RS_LEVEL(cfg, w->shift, sh, m);
		if (w->dig) for (i = w->beg; i < w->end; ++i) { const unsigned dgt = (unsigned)(RS_WORD(a[i]) >> sh & m); w->dig[i] = (uint8_t)dgt; ++w->cnt[dgt]; }
		else for (i = w->beg; i < w->end; ++i) diff |= RS_ORIG(a[i], cfg) ^ k0, ++w->cnt[RS_WORD(a[i]) >> sh & m];
	} else for (i = w->beg; i < w->end; ++i) diff |= RS_ORIG(a[i], cfg) ^ k0;
	w->diff = diff;
	return 0;
}

static void RS_NAME(task)(rs_pool_t *pool, void *a, size_t n, int shift) { RS_NAME(level)(pool, (RS_T*)a, n, shift); }

/* the whole sort of a[0..n) */
/* the top level given its digits (dig[], one byte per element, and their counts), then the buckets below on the worker threads */
static void RS_NAME(sort_from_top)(RS_T *a, size_t n, const rs_cfg_t *cfg, int n_threads, const size_t *cnt, const uint8_t *dig, int shift)
{
	rs_pool_t p;
	pthread_t *th;
	int t;
	(void)n;
	memset(&p, 0, sizeof(p));
	p.cfg = *cfg; p.run = RS_NAME(task); p.elem = sizeof(RS_T);
	size_t tstart[257];
	p.wcum = tl_wcum; p.n_ids = tl_n_ids;
	if (p.wcum && tl_top_apart) { /* (restricted sort: the buckets of this level say what the words leave out) */
		int k;
		tstart[0] = 0;
		for (k = 0; k < 256; ++k) tstart[k + 1] = tstart[k] + cnt[k];
		p.top_start = tstart; p.top_base = a; p.top_shift = shift;
	}
	pthread_mutex_init(&p.mu, 0);
	pthread_cond_init(&p.cv, 0);
	if (n_threads > RS_MAX_THREADS) n_threads = RS_MAX_THREADS;
	p.n_threads = n_threads;
	{
		RS_T0;
		th = (pthread_t*)malloc(sizeof(pthread_t) * n_threads);
		if (rs_timing < 0) rs_timing = getenv("MA_REFSORT_TIMING") != 0;
		++p.busy; /* the top-level walk below produces tasks: workers must not leave while it runs */
		for (t = 0; t < n_threads; ++t) pthread_create(&th[t], 0, pool_worker, &p);
		if (dig) RS_NAME(permute_top)(&p, a, cnt, dig, shift);
		else { size_t tail[256]; memcpy(tail, cnt, sizeof(tail)); RS_NAME(permute)(&p, a, tail, shift); }
		RS_LAP(" top walk");
		p.st_t0 = rs_now();
		pthread_mutex_lock(&p.mu);
		--p.busy;
		pthread_cond_broadcast(&p.cv);
		pthread_mutex_unlock(&p.mu);
		for (t = 0; t < n_threads; ++t) pthread_join(th[t], 0);
		RS_LAP(" buckets");
		if (rs_timing > 0) fprintf(stderr, "[T::refsort]  tasks: %lu (%lu big), %.3f s inside them on %d threads (big ones %.3f s, the last of them done %.3f s after the walk), longest %.3f s\n",
		                           (unsigned long)p.st_n, (unsigned long)p.st_nbig, p.st_busy, n_threads, p.st_big, p.st_last_big, p.st_max);
		free(th); free(p.q[0]); free(p.q[1]);
	}
	pthread_mutex_destroy(&p.mu);
	pthread_cond_destroy(&p.cv);
}

/* the whole sort of a[0..n) */
static void RS_NAME(sort)(RS_T *a, size_t n, const rs_cfg_t *cfg, int n_threads)
{
	rs_pool_t p;
	memset(&p, 0, sizeof(p));
	p.cfg = *cfg; p.n_threads = 1; p.run = RS_NAME(task); p.elem = sizeof(RS_T);
	p.wcum = tl_wcum; p.n_ids = tl_n_ids;
	if (n <= RS_SMALL) { RS_NAME(insertion)(a, n, cfg); return; } /* ksort.h:182 */
	if (n_threads <= 1 || n < (1u << 17)) { RS_NAME(level)(&p, a, n, 56); return; }
	{
		size_t cnt[256];
		int shift = 56;
		uint64_t diff;
		if (n_threads > RS_MAX_THREADS) n_threads = RS_MAX_THREADS;
		/* the top level: its two sweeps (which bits vary; the digit counts) run on all threads, only the walk itself is sequential */
		RS_T0;
		diff = sweep_run(RS_NAME(sweep_worker), a, n, -1, 0, cfg, n_threads, 0);
		if (diff != 0) {
			uint8_t *dig = (uint8_t*)malloc(n + 16);
			while (shift > 0 && (diff >> shift & 0xff) == 0) shift -= 8;
			if (dig) memset(dig + n, 0, 16);
			sweep_run(RS_NAME(sweep_worker), a, n, shift, cnt, cfg, n_threads, dig);
			RS_LAP(" sweeps");
			RS_NAME(sort_from_top)(a, n, cfg, n_threads, cnt, dig, shift);
			free(dig);
		}
	}
}
