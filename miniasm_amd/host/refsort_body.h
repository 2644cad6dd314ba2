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

/* the buckets a level leaves behind (ksort.h:177-182): radix again when larger than 64, else the stable insertion sort */
static void RS_NAME(dispatch)(rs_pool_t *pool, RS_T *a, const size_t *start, int shift)
{
	const rs_cfg_t *cfg = &pool->cfg;
	int k, next = shift > 8 ? shift - 8 : 0;
	if (!shift) return;
	for (k = 0; k < 256; ++k) {
		size_t cnt = start[k + 1] - start[k];
		if (cnt > RS_SMALL) {
			if (pool->n_threads > 1 && cnt >= TASK_MIN) pool_push(pool, a + start[k], cnt, next);
			else RS_NAME(level)(pool, a + start[k], cnt, next);
		} else if (cnt > 1) RS_NAME(insertion)(a + start[k], cnt, cfg);
	}
}

/* The same permutation for the top level of a big input, where the walk is the one thing no other thread can help with -- computed from the
 * DIGITS alone and without moving anything.  The walk only ever evicts ORIGINAL elements (a slot at or behind a bucket's head has not been written
 * yet), so (a) where it goes next is a function of dig[] (one byte per element, written by the parallel sweep that counts the digits), and (b) the
 * element it picks up at a slot is the one that started there: the result can be written down as perm[slot] = original position of the arrival.
 * Runs are taken at once.  At the base bucket k a stretch of elements that are already home stays where it is (perm is the identity there, filled
 * beforehand by all threads).  Elsewhere an arrival at the head of a stretch of m home elements pushes every one of them one slot to the right
 * before the walk can leave with the first stranger: perm[slot + j] = slot + j - 1, written as one run -- or, when long, noted as a segment for the
 * parallel copy afterwards.  PAF files list a query's overlaps together, so half of the records (all of them, for a generator that writes reads in
 * genome order) are home at the top level: what cost 4.3 ns per element as a dependent chain is a byte scan there.  The elements are then moved once,
 * by all threads (rs_apply), instead of one by one inside the chain. */
typedef struct { size_t dst, src, len; } RS_NAME(seg_t);
typedef struct { const RS_T *in; RS_T *out; const uint32_t *perm; const RS_NAME(seg_t) *seg; size_t n, n_seg, beg, end; int phase; } RS_NAME(apply_t);

static inline size_t RS_NAME(run)(const uint8_t *dig, size_t p, size_t lim, unsigned d) /* length of the stretch of digit d that starts at p (not beyond lim) */
{
	const uint64_t pat = 0x0101010101010101ull * d;
	size_t q = p;
	while (q + 8 <= lim) {
		uint64_t x;
		memcpy(&x, dig + q, 8);
		x ^= pat;
		if (x) return q + ((size_t)__builtin_ctzll(x) >> 3) - p;
		q += 8;
	}
	while (q < lim && dig[q] == d) ++q;
	return q - p;
}

static void *RS_NAME(apply_worker)(void *arg)
{
	RS_NAME(apply_t) *w = (RS_NAME(apply_t)*)arg;
	size_t i;
	if (w->phase == 0) for (i = w->beg; i < w->end; ++i) ((uint32_t*)w->perm)[i] = (uint32_t)i;
	else if (w->phase == 1) for (i = w->beg; i < w->end; ++i) w->out[i] = w->in[w->perm[i]];
	else if (w->phase == 2) { /* the shifted stretches: every thread takes its share [beg, end) of the destination range */
		for (i = 0; i < w->n_seg; ++i) {
			const size_t lo = w->seg[i].dst > w->beg ? w->seg[i].dst : w->beg, e0 = w->seg[i].dst + w->seg[i].len, hi = e0 < w->end ? e0 : w->end;
			if (lo < hi) memcpy(w->out + lo, w->in + (w->seg[i].src + (lo - w->seg[i].dst)), (hi - lo) * sizeof(RS_T));
		}
	}
	return 0;
}

static void RS_NAME(apply_run)(RS_NAME(apply_t) *proto, int phase, int n_threads)
{
	RS_NAME(apply_t) w[64];
	pthread_t th[64];
	int t;
	if (n_threads > 64) n_threads = 64;
	if (n_threads < 1) n_threads = 1;
	for (t = 0; t < n_threads; ++t) { w[t] = *proto; w[t].phase = phase; w[t].beg = proto->n / n_threads * t; w[t].end = t == n_threads - 1 ? proto->n : proto->n / n_threads * (t + 1); }
	for (t = 1; t < n_threads; ++t) pthread_create(&th[t], 0, RS_NAME(apply_worker), &w[t]);
	RS_NAME(apply_worker)(&w[0]);
	for (t = 1; t < n_threads; ++t) pthread_join(th[t], 0);
}

#define RS_SEG_MIN 4096 /* shifted stretches at least this long are copied by the threads afterwards instead of being written out inside the walk */
static int RS_NAME(permute_top)(rs_pool_t *pool, RS_T **pa, size_t n, const size_t *cnt, const uint8_t *dig /* n + 8 bytes */, int shift)
{ /* *pa (malloc'ed) is replaced by the permuted copy */
	RS_T *a = *pa;
	size_t start[257], head[256];
	uint32_t *perm = (uint32_t*)ma_big_malloc((n + 1) * sizeof(uint32_t));
	RS_T *out = (RS_T*)ma_big_malloc((n + 1) * sizeof(RS_T));
	RS_NAME(seg_t) *seg = 0;
	size_t n_seg = 0, m_seg = 0;
	RS_NAME(apply_t) ap;
	int k;
	if (perm == 0 || out == 0) { free(perm); free(out); return -1; }
	memset(&ap, 0, sizeof(ap));
	ap.in = a; ap.out = out; ap.perm = perm; ap.n = n;
	RS_NAME(apply_run)(&ap, 0, pool->n_threads); /* perm = identity */
	start[0] = 0;
	for (k = 0; k < 256; ++k) start[k + 1] = start[k] + cnt[k], head[k] = start[k];
	for (k = 0; k < 256; ++k) {
		const size_t end_k = start[k + 1];
		for (;;) {
			size_t carry;
			unsigned l;
			head[k] += RS_NAME(run)(dig, head[k], end_k, (unsigned)k); /* already home */
			if (head[k] == end_k) break;
			carry = head[k]; l = dig[carry];
			do {
				const size_t slot = head[l], m = RS_NAME(run)(dig, slot, start[l + 1], l); /* m elements of bucket l wait at its head: each moves up by one */
				perm[slot] = (uint32_t)carry;
				if (m >= RS_SEG_MIN) {
					if (n_seg == m_seg) { m_seg = m_seg ? m_seg << 1 : 256; seg = (RS_NAME(seg_t)*)realloc(seg, m_seg * sizeof(*seg)); }
					seg[n_seg].dst = slot + 1; seg[n_seg].src = slot; seg[n_seg].len = m; ++n_seg;
				} else { size_t j; for (j = 1; j <= m; ++j) perm[slot + j] = (uint32_t)(slot + j - 1); }
				carry = slot + m; /* the first stranger behind them leaves */
				head[l] = slot + m + 1;
				l = dig[carry];
			} while (l != (unsigned)k);
			perm[head[k]++] = (uint32_t)carry;
		}
	}
	ap.seg = seg; ap.n_seg = n_seg;
	if (rs_timing > 0) fprintf(stderr, "[T::refsort]  top walk: %zu shifted stretches of >= %d elements left to the copy threads\n", n_seg, RS_SEG_MIN);
	RS_NAME(apply_run)(&ap, 1, pool->n_threads);
	if (n_seg) RS_NAME(apply_run)(&ap, 2, pool->n_threads);
	free(perm); free(seg); free(a);
	*pa = out;
	RS_NAME(dispatch)(pool, out, start, shift);
	return 0;
}

/* (round 2's form, kept for one comparison run: MA_REFSORT_MOVES=1) the digit walk that moves the elements as it goes */
typedef struct { size_t head; uint32_t nd; uint32_t pad; } RS_NAME(bk_t);
static void RS_NAME(permute_top_moves)(rs_pool_t *pool, RS_T *a, const size_t *cnt, const uint8_t *dig /* n + 1 bytes */, int shift)
{
	RS_NAME(bk_t) b[256];
	size_t start[257];
	int k;
	start[0] = 0;
	for (k = 0; k < 256; ++k) start[k + 1] = start[k] + cnt[k], b[k].head = start[k], b[k].nd = dig[start[k]], b[k].pad = 0;
	for (k = 0; k < 256;) {
		unsigned d;
		if (b[k].head == start[k + 1]) { ++k; continue; }
		d = b[k].nd;
		if (d == (unsigned)k) { b[k].nd = dig[++b[k].head]; continue; }
		{
			RS_T carry = a[b[k].head];
			do {
				const size_t slot = b[d].head;
				const unsigned dn = b[d].nd;
				const RS_T evicted = a[slot];
				b[d].head = slot + 1;
				b[d].nd = dig[slot + 1];
				a[slot] = carry;
				RS_PREFETCH(&a[slot]);
				carry = evicted;
				d = dn;
			} while (d != (unsigned)k);
			a[b[k].head] = carry;
			b[k].nd = dig[++b[k].head];
		}
	}
	RS_NAME(dispatch)(pool, a, start, shift);
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

/* one level of ksort.h:153-179 on a[0..n) */
static void RS_NAME(level)(rs_pool_t *pool, RS_T *a, size_t n, int shift)
{
	const rs_cfg_t *cfg = &pool->cfg;
	size_t tail[256], i;
	/* A level on which the digit does not vary leaves the range untouched and recurses into the same range (n > 64
	 * here).  One sweep gives the varying bits and, optimistically, the histogram of the current digit. */
	for (;;) {
		uint64_t diff = 0;
		const uint64_t k0 = RS_ORIG(a[0], cfg);
		int sh;
		unsigned m;
		RS_LEVEL(cfg, shift, sh, m);
		memset(tail, 0, sizeof(tail));
		for (i = 0; i < n; ++i) diff |= RS_ORIG(a[i], cfg) ^ k0, ++tail[RS_WORD(a[i]) >> sh & m];
		if (diff == 0) return; /* all keys equal: every remaining level is the identity */
		if ((diff >> shift & 0xff) != 0) break;
		while (shift > 0 && (diff >> shift & 0xff) == 0) shift -= 8;
	}
	RS_NAME(permute)(pool, a, tail, shift);
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
static void RS_NAME(sort)(RS_T **pa, size_t n, const rs_cfg_t *cfg, int n_threads)
{ /* *pa is malloc'ed and may be replaced */
	RS_T *a = *pa;
	rs_pool_t p;
	memset(&p, 0, sizeof(p));
	p.cfg = *cfg; p.n_threads = 1; p.run = RS_NAME(task); p.elem = sizeof(RS_T);
	if (n <= RS_SMALL) { RS_NAME(insertion)(a, n, cfg); return; } /* ksort.h:182 */
	if (n_threads <= 1 || n < (1u << 17)) { RS_NAME(level)(&p, a, n, 56); return; }
	{
		pthread_t *th;
		size_t cnt[256];
		int t, shift = 56;
		uint64_t diff;
		pthread_mutex_init(&p.mu, 0);
		pthread_cond_init(&p.cv, 0);
		if (n_threads > 64) n_threads = 64;
		p.n_threads = n_threads;
		/* the top level: its two sweeps (which bits vary; the digit counts) run on all threads, only the walk itself is sequential */
		RS_T0;
		diff = sweep_run(RS_NAME(sweep_worker), a, n, -1, 0, cfg, n_threads, 0);
		if (diff != 0) {
			uint8_t *dig = (uint8_t*)ma_big_malloc(n + 16);
			while (shift > 0 && (diff >> shift & 0xff) == 0) shift -= 8;
			if (dig) memset(dig + n, 0, 16);
			sweep_run(RS_NAME(sweep_worker), a, n, shift, cnt, cfg, n_threads, dig);
			RS_LAP(" sweeps");
			th = (pthread_t*)malloc(sizeof(pthread_t) * n_threads);
			++p.busy; /* the top-level walk below produces tasks: workers must not leave while it runs */
			for (t = 0; t < n_threads; ++t) pthread_create(&th[t], 0, pool_worker, &p);
			if (dig && getenv("MA_REFSORT_MOVES")) RS_NAME(permute_top_moves)(&p, a, cnt, dig, shift);
			else if (dig == 0 || RS_NAME(permute_top)(&p, pa, n, cnt, dig, shift) != 0) RS_NAME(permute)(&p, a, cnt, shift); /* (no memory for the digit walk: the plain one) */
			free(dig);
			RS_LAP(" top walk");
			pthread_mutex_lock(&p.mu);
			--p.busy;
			pthread_cond_broadcast(&p.cv);
			pthread_mutex_unlock(&p.mu);
			for (t = 0; t < n_threads; ++t) pthread_join(th[t], 0);
			RS_LAP(" buckets");
			free(th); free(p.q);
		}
		pthread_mutex_destroy(&p.mu);
		pthread_cond_destroy(&p.cv);
	}
}
