/* pafgen -- seeded synthetic PAF generator for the miniasm hot path (test/bench tooling, no reference code).
 *
 * Model: R reads laid on a random linear genome of length G; one PAF line per read
 * pair whose genomic intersection is >= min_ovlp bp; coordinates are projected EXACTLY (no indels), so the
 * overlap lengths "len" of two arcs leaving one read end differ whenever read starts and read ends are
 * pairwise distinct -- which the generator enforces.  That keeps the input free of (u,len) arc ties, the
 * only situation in which the reference's unstable radix sort makes its output depend on input order
 * (SURVEY.md section 5.9).
 *
 *   G is found by bisection so that the expected number of emitted lines equals -n.
 *   Read lengths: lognormal (default; containment-heavy, realistic), fixed or uniform (graph-heavy).
 *   Noise: -d dropout fraction, -x false dovetail overlaps (tips/bubbles), -i low-identity fraction.
 *   Line order: grouped by query read, query reads in random order (like an all-vs-all overlapper run on
 *   reads in sequencing order; default) or in genome order (-g; SURVEY's original recipe).
 *   Realistic noise (round 6; what every real overlapper emits, and where the reference's tie order is the COMMON case, not the exception):
 *     -j K   every coordinate of a line (qs, qe, ts, te) and its block length moved independently by up to +-K bp (alignment ends are not exact projections),
 *            so equal sort keys and equal (u,len) arc keys turn up by chance all over the file;
 *     -b F   a fraction F of the pairs is listed in BOTH directions (a line in either read's block, each with its own jitter);
 *     -N P   read names are P followed by the read's number and "/ccs" (default: "r" and the bare number): -N m64011_190830_220126/ gives names of the length real
 *            PacBio / ONT files carry (the device dictionary keys names of up to 8 bytes by their bytes and compares the text of longer ones);
 *     -t     the lines are grouped by TARGET (the roles of the two reads swapped when a line is written): a query's lines are scattered over the file,
 *            as in a PAF sorted by target -- the hit sort cannot take runs of one query's records.
 *
 * Usage: pafgen -r 200000 -n 10000000 -s 1 [-L lognormal|fixed|uniform] [-m mean] [-o out.paf]
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <unistd.h>

typedef struct { uint32_t start, len, name; uint8_t strand; } gread_t;

static uint64_t sm_state;
static inline uint64_t splitmix64(void)
{
	uint64_t z = (sm_state += 0x9E3779B97F4A7C15ULL);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}
static inline double urand(void) { return (splitmix64() >> 11) * (1.0 / 9007199254740992.0); }
static inline uint64_t mix64(uint64_t z)
{
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}

static int cmp_start(const void *a, const void *b)
{
	const gread_t *x = (const gread_t*)a, *y = (const gread_t*)b;
	if (x->start != y->start) return x->start < y->start ? -1 : 1;
	return x->name < y->name ? -1 : x->name > y->name;
}
static int cmp_u64(const void *a, const void *b)
{
	uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
	return x < y ? -1 : x > y;
}

static const char *g_name_prefix; /* -N */
/* fast unsigned -> decimal */
static inline char *put_u32(char *p, uint32_t x)
{
	char tmp[12]; int n = 0;
	do { tmp[n++] = '0' + x % 10; x /= 10; } while (x);
	while (n) *p++ = tmp[--n];
	return p;
}
static inline char *put_name(char *p, uint32_t x)
{
	if (g_name_prefix == 0) { *p++ = 'r'; return put_u32(p, x); }
	{ const char *q = g_name_prefix; while (*q) *p++ = *q++; }
	p = put_u32(p, x);
	*p++ = '/'; *p++ = 'c'; *p++ = 'c'; *p++ = 's';
	return p;
}

typedef struct {
	uint32_t n_reads, len_min, len_max, min_ovlp, quantum;
	uint64_t n_lines, seed;
	int model; /* 0 lognormal, 1 fixed, 2 uniform */
	double mean, sigma, dropout, false_frac, lowid_frac;
	int genome_order;
	uint32_t jitter; double both_frac; int by_target;
	const char *name_prefix;
} opt_t;
static opt_t g_o;

static uint64_t count_pairs(const gread_t *r, uint32_t n, uint32_t min_ovlp)
{ /* reads sorted by start: pairs (i<j) with start_j + min_ovlp <= end_i  (len_j >= min_ovlp always) */
	uint64_t tot = 0;
	uint32_t i, j = 0;
	/* end_i is not monotone, so binary search per read */
	for (i = 0; i < n; ++i) {
		uint32_t lim, lo = i + 1, hi = n;
		(void)j;
		if (r[i].len < min_ovlp) continue;
		lim = r[i].start + r[i].len - min_ovlp; /* start_j <= lim */
		while (lo < hi) { uint32_t mid = lo + ((hi - lo) >> 1); if (r[mid].start <= lim) lo = mid + 1; else hi = mid; }
		tot += lo - (i + 1);
	}
	return tot;
}

static void place(gread_t *r, const double *u, uint32_t n, uint64_t G, uint32_t len_max)
{
	uint32_t i;
	uint64_t span = G > (uint64_t)len_max + 1 ? G - len_max : 1;
	for (i = 0; i < n; ++i) r[i].start = (uint32_t)(u[r[i].name] * (double)span);
	qsort(r, n, sizeof(gread_t), cmp_start);
}

static char *emit(char *p, const gread_t *a, const gread_t *b, uint32_t gs, uint32_t ge, int low_id, uint64_t h)
{ /* one PAF line: query a, target b, genomic intersection [gs,ge) */
	uint32_t qs, qe, ts, te, bl, ml;
	double f;
	if (a->strand == 0) qs = gs - a->start, qe = ge - a->start;
	else qs = a->start + a->len - ge, qe = a->start + a->len - gs;
	if (b->strand == 0) ts = gs - b->start, te = ge - b->start;
	else ts = b->start + b->len - ge, te = b->start + b->len - gs;
	bl = ge - gs;
	if (g_o.jitter) { /* alignment ends are not exact projections: every coordinate moves on its own, inside its read, the spans stay positive */
		const uint32_t K = g_o.jitter, W = 2 * K + 1;
		uint64_t hj = mix64(h ^ 0xA24BAED4963EE407ULL);
		int64_t v;
		v = (int64_t)qs + (int64_t)(hj % W) - K; hj = mix64(hj); if (v < 0) v = 0; if (v > (int64_t)a->len - 2) v = (int64_t)a->len - 2; qs = (uint32_t)v;
		v = (int64_t)qe + (int64_t)(hj % W) - K; hj = mix64(hj); if (v <= (int64_t)qs) v = (int64_t)qs + 1; if (v > (int64_t)a->len) v = a->len; qe = (uint32_t)v;
		v = (int64_t)ts + (int64_t)(hj % W) - K; hj = mix64(hj); if (v < 0) v = 0; if (v > (int64_t)b->len - 2) v = (int64_t)b->len - 2; ts = (uint32_t)v;
		v = (int64_t)te + (int64_t)(hj % W) - K; hj = mix64(hj); if (v <= (int64_t)ts) v = (int64_t)ts + 1; if (v > (int64_t)b->len) v = b->len; te = (uint32_t)v;
		bl = (qe - qs > te - ts ? qe - qs : te - ts) + (uint32_t)(hj % (K + 1));
	}
	f = (h >> 11) * (1.0 / 9007199254740992.0);
	f = low_id ? 0.01 + 0.08 * f : 0.08 + 0.22 * f;
	ml = (uint32_t)(bl * f);
	p = put_name(p, a->name); *p++ = '\t';
	p = put_u32(p, a->len); *p++ = '\t'; p = put_u32(p, qs); *p++ = '\t'; p = put_u32(p, qe); *p++ = '\t';
	*p++ = a->strand == b->strand ? '+' : '-'; *p++ = '\t';
	p = put_name(p, b->name); *p++ = '\t';
	p = put_u32(p, b->len); *p++ = '\t'; p = put_u32(p, ts); *p++ = '\t'; p = put_u32(p, te); *p++ = '\t';
	p = put_u32(p, ml); *p++ = '\t'; p = put_u32(p, bl); *p++ = '\t';
	*p++ = '2'; *p++ = '5'; *p++ = '5'; *p++ = '\n';
	return p;
}

static char *emit_false(char *p, const gread_t *a, const gread_t *b, uint64_t h)
{ /* false dovetail: suffix of a (x bp) onto prefix (or, reversed, suffix) of b */
	uint32_t x = 2500 + (uint32_t)(h % 2501), m = a->len < b->len ? a->len : b->len;
	uint32_t qs, qe, ts, te, ml;
	int rev = (h >> 40) & 1;
	if (x + 500 > m) return p;
	qs = a->len - x, qe = a->len;
	if (!rev) ts = 0, te = x; else ts = b->len - x, te = b->len;
	ml = (uint32_t)(x * (0.08 + 0.22 * ((h >> 11 & 0xfffff) / 1048576.0)));
	p = put_name(p, a->name); *p++ = '\t';
	p = put_u32(p, a->len); *p++ = '\t'; p = put_u32(p, qs); *p++ = '\t'; p = put_u32(p, qe); *p++ = '\t';
	*p++ = rev ? '-' : '+'; *p++ = '\t';
	p = put_name(p, b->name); *p++ = '\t';
	p = put_u32(p, b->len); *p++ = '\t'; p = put_u32(p, ts); *p++ = '\t'; p = put_u32(p, te); *p++ = '\t';
	p = put_u32(p, ml); *p++ = '\t'; p = put_u32(p, x); *p++ = '\t';
	*p++ = '2'; *p++ = '5'; *p++ = '5'; *p++ = '\n';
	return p;
}

/* a hash of the PAIR (the same from either side): dropout and -b decide per pair */
static uint64_t pair_hash(uint32_t x, uint32_t y, const opt_t *o)
{
	const uint32_t lo = x < y ? x : y, hi = x < y ? y : x;
	return mix64(((uint64_t)lo << 32 | hi) ^ (o->seed * 0xC2B2AE3D27D4EB4FULL));
}
static int both_ways(uint32_t x, uint32_t y, const opt_t *o)
{
	return o->both_frac > 0 && (pair_hash(x, y, o) >> 40 & 0xffff) < (uint64_t)(o->both_frac * 65536.0);
}

int main(int argc, char *argv[])
{
	opt_t o;
	int c;
	uint32_t i, k, *order, *rank;
	uint64_t G, lo, hi, tot = 0, want;
	gread_t *r;
	double *u;
	const char *fn_out = 0;
	FILE *fp;
	char *buf, *p;
	size_t bufcap = 1u << 22;

	memset(&o, 0, sizeof(o));
	o.n_reads = 2000; o.n_lines = 50000; o.seed = 1; o.model = 0; o.mean = 8000.; o.sigma = .5;
	o.len_min = 2500; o.len_max = 60000; o.min_ovlp = 2000;
	while ((c = getopt(argc, argv, "r:n:s:L:m:S:d:x:i:o:gl:M:O:q:j:b:tN:")) >= 0) {
		if (c == 'r') o.n_reads = atol(optarg);
		else if (c == 'n') o.n_lines = atoll(optarg);
		else if (c == 's') o.seed = atoll(optarg);
		else if (c == 'L') o.model = strcmp(optarg, "fixed") == 0 ? 1 : strcmp(optarg, "uniform") == 0 ? 2 : 0;
		else if (c == 'm') o.mean = atof(optarg);
		else if (c == 'S') o.sigma = atof(optarg);
		else if (c == 'd') o.dropout = atof(optarg);
		else if (c == 'x') o.false_frac = atof(optarg);
		else if (c == 'i') o.lowid_frac = atof(optarg);
		else if (c == 'o') fn_out = optarg;
		else if (c == 'g') o.genome_order = 1;
		else if (c == 'l') o.len_min = atol(optarg);
		else if (c == 'M') o.len_max = atol(optarg);
		else if (c == 'O') o.min_ovlp = atol(optarg);
		else if (c == 'q') o.quantum = atol(optarg); /* coordinates on a grid: many equal sort keys (tie-order tests) */
		else if (c == 'j') o.jitter = atol(optarg);
		else if (c == 'b') o.both_frac = atof(optarg);
		else if (c == 't') o.by_target = 1;
		else if (c == 'N') o.name_prefix = optarg;
	}
	g_o = o;
	g_name_prefix = o.name_prefix && strlen(o.name_prefix) < 200 ? o.name_prefix : 0;
	if (o.n_reads < 2) { fprintf(stderr, "pafgen: need >= 2 reads\n"); return 1; }
	sm_state = o.seed * 0x2545F4914F6CDD1DULL + 12345;

	r = (gread_t*)calloc(o.n_reads, sizeof(gread_t));
	u = (double*)calloc(o.n_reads, sizeof(double));
	for (i = 0; i < o.n_reads; ++i) {
		double len;
		if (o.model == 1) len = o.mean;
		else if (o.model == 2) len = o.mean * (0.875 + 0.25 * urand()); /* U[7000,9000] at mean 8000 */
		else { /* lognormal(mu = ln(mean) - sigma^2/2, sigma) */
			double u1 = urand(), u2 = urand();
			double z = sqrt(-2.0 * log(u1 > 1e-300 ? u1 : 1e-300)) * cos(6.283185307179586 * u2);
			len = exp(log(o.mean) - .5 * o.sigma * o.sigma + o.sigma * z);
		}
		if (len < o.len_min) len = o.len_min;
		if (len > o.len_max) len = o.len_max;
		r[i].len = (uint32_t)len;
		r[i].name = i;
		r[i].strand = splitmix64() & 1;
		u[i] = urand();
	}
	/* bisection on G so that pairs*(1-dropout) ~= n_lines (before false lines) */
	want = (uint64_t)(o.n_lines / (1.0 + o.false_frac) / (1.0 - o.dropout > .01 ? 1.0 - o.dropout : .01));
	lo = o.len_max + 2; hi = (uint64_t)o.n_reads * o.len_max + o.len_max + 2;
	if (hi > 0xF0000000ULL) hi = 0xF0000000ULL;
	for (k = 0; k < 48 && lo + 1 < hi; ++k) {
		G = lo + ((hi - lo) >> 1);
		place(r, u, o.n_reads, G, o.len_max);
		tot = count_pairs(r, o.n_reads, o.min_ovlp);
		if (tot > want) lo = G; else hi = G;
	}
	G = hi;
	place(r, u, o.n_reads, G, o.len_max);
	if (o.quantum > 1) { /* tie-rich variant: starts and lengths on a grid, no de-duplication */
		for (i = 0; i < o.n_reads; ++i) {
			r[i].start = r[i].start / o.quantum * o.quantum;
			r[i].len = r[i].len / o.quantum * o.quantum;
			if (r[i].len < o.quantum) r[i].len = o.quantum;
		}
	}
	/* make starts pairwise distinct, then ends pairwise distinct (shrink a read by 1 bp on collision) */
	for (i = 1; i < o.n_reads && o.quantum <= 1; ++i)
		if (r[i].start <= r[i-1].start) r[i].start = r[i-1].start + 1;
	if (o.quantum <= 1) {
		uint64_t *e = (uint64_t*)malloc(sizeof(uint64_t) * o.n_reads);
		int changed = 1, iter = 0;
		while (changed && iter++ < 64) {
			changed = 0;
			for (i = 0; i < o.n_reads; ++i) e[i] = (uint64_t)(r[i].start + r[i].len) << 32 | i;
			qsort(e, o.n_reads, 8, cmp_u64);
			for (i = 1; i < o.n_reads; ++i)
				if (e[i] >> 32 == e[i-1] >> 32) { gread_t *q = &r[(uint32_t)e[i]]; if (q->len > o.min_ovlp + 1) --q->len, changed = 1; }
		}
		free(e);
	}
	tot = count_pairs(r, o.n_reads, o.min_ovlp);
	fprintf(stderr, "[pafgen] reads=%u genome=%llu bp pairs=%llu depth=%.1f\n", o.n_reads, (unsigned long long)G,
			(unsigned long long)tot, (double)o.n_reads * o.mean / (double)G);

	/* file order of the query reads */
	order = (uint32_t*)malloc(4 * o.n_reads);
	rank = (uint32_t*)malloc(4 * o.n_reads);
	for (i = 0; i < o.n_reads; ++i) order[i] = i;
	if (!o.genome_order)
		for (i = o.n_reads - 1; i > 0; --i) { uint32_t j = (uint32_t)(splitmix64() % (i + 1)), t = order[i]; order[i] = order[j]; order[j] = t; }
	for (i = 0; i < o.n_reads; ++i) rank[order[i]] = i;

	fp = fn_out && strcmp(fn_out, "-") ? fopen(fn_out, "w") : stdout;
	if (!fp) { fprintf(stderr, "pafgen: cannot write %s\n", fn_out); return 1; }
	buf = (char*)malloc(bufcap + 4096);
	p = buf;
	tot = 0;
	for (k = 0; k < o.n_reads; ++k) {
		uint32_t a = order[k], j;
		const gread_t *ra = &r[a];
		uint32_t a_end = ra->start + ra->len;
		/* partners to the left (start_j < start_a): need end_j >= start_a + min_ovlp */
		for (j = a; j-- > 0;) {
			const gread_t *rb = &r[j];
			uint32_t b_end = rb->start + rb->len, ge;
			uint64_t h;
			if (ra->start - rb->start > o.len_max) break;
			if (rank[j] < k && !both_ways(ra->name, rb->name, &o)) continue; /* pair already emitted with j as the query (-b: some pairs are listed from both sides) */
			ge = a_end < b_end ? a_end : b_end;
			if (ge < ra->start + o.min_ovlp) continue;
			h = mix64(((uint64_t)ra->name << 32 | rb->name) ^ (o.seed * 0x9E3779B97F4A7C15ULL));
			if (o.dropout > 0 && ((o.both_frac > 0 ? pair_hash(ra->name, rb->name, &o) >> 8 : h) & 0xffffff) < (uint64_t)(o.dropout * 16777216.0)) continue; /* (-b: per pair, the same from either side) */
			p = o.by_target ? emit(p, rb, ra, ra->start, ge, o.lowid_frac > 0 && (h >> 24 & 0xffff) < (uint64_t)(o.lowid_frac * 65536.0), mix64(h))
			                : emit(p, ra, rb, ra->start, ge, o.lowid_frac > 0 && (h >> 24 & 0xffff) < (uint64_t)(o.lowid_frac * 65536.0), mix64(h));
			++tot;
			if ((size_t)(p - buf) > bufcap) fwrite(buf, 1, p - buf, fp), p = buf;
		}
		for (j = a + 1; j < o.n_reads; ++j) {
			const gread_t *rb = &r[j];
			uint32_t b_end = rb->start + rb->len, ge;
			uint64_t h;
			if (rb->start + o.min_ovlp > a_end) break;
			if (rank[j] < k && !both_ways(ra->name, rb->name, &o)) continue;
			ge = a_end < b_end ? a_end : b_end;
			h = mix64(((uint64_t)ra->name << 32 | rb->name) ^ (o.seed * 0x9E3779B97F4A7C15ULL));
			if (o.dropout > 0 && ((o.both_frac > 0 ? pair_hash(ra->name, rb->name, &o) >> 8 : h) & 0xffffff) < (uint64_t)(o.dropout * 16777216.0)) continue; /* (-b: per pair, the same from either side) */
			p = o.by_target ? emit(p, rb, ra, rb->start, ge, o.lowid_frac > 0 && (h >> 24 & 0xffff) < (uint64_t)(o.lowid_frac * 65536.0), mix64(h))
			                : emit(p, ra, rb, rb->start, ge, o.lowid_frac > 0 && (h >> 24 & 0xffff) < (uint64_t)(o.lowid_frac * 65536.0), mix64(h));
			++tot;
			if ((size_t)(p - buf) > bufcap) fwrite(buf, 1, p - buf, fp), p = buf;
		}
		if (o.false_frac > 0) { /* a few false dovetails per query, to random partners */
			double expect = o.false_frac * (double)want / o.n_reads;
			uint32_t nf = (uint32_t)expect;
			uint64_t h = mix64((uint64_t)ra->name * 0xD6E8FEB86659FD93ULL + o.seed);
			if ((h & 0xffffff) < (uint64_t)((expect - nf) * 16777216.0)) ++nf;
			for (j = 0; j < nf; ++j) {
				uint32_t b;
				char *q0 = p;
				h = mix64(h + j + 1);
				b = (uint32_t)(h % o.n_reads);
				if (b == a) continue;
				p = o.by_target ? emit_false(p, &r[b], ra, mix64(h)) : emit_false(p, ra, &r[b], mix64(h));
				if (p != q0) ++tot;
			}
			if ((size_t)(p - buf) > bufcap) fwrite(buf, 1, p - buf, fp), p = buf;
		}
	}
	fwrite(buf, 1, p - buf, fp);
	if (fp != stdout) fclose(fp);
	fprintf(stderr, "[pafgen] wrote %llu lines\n", (unsigned long long)tot);
	free(buf); free(order); free(rank); free(r); free(u);
	return 0;
}
