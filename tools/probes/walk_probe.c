/* walk_probe.c -- the dependent chain of host/refsort_body.h's top-level walk, alone: ns per element for a few forms of the loop, bucket counts
 * and page sizes, on the host CPU of the box (gcc -O2 -o walk_probe walk_probe.c; tools/gpu_round.sh walkprobe).
 * forms: 0 = ksort.h's two loops with the next digit kept per bucket (permute_top), 1 = the same without moving elements (the chain alone),
 *        3 = one loop, read + write position per bucket (permute_uniform), 5 = 3 + software prefetch of the digit stream, 6 = 0 + that prefetch. */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <time.h>
#include <sys/mman.h>
static double now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
typedef struct { size_t head; uint32_t nd; uint32_t pad; } bk_t;
typedef struct { size_t r, w; } rw_t;
typedef uint64_t T;
static uint64_t rng = 88172645463325252ull;
static inline uint64_t xr(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; }
static void *big(size_t bytes, int thp)
{
	if (thp) {
		size_t b = (bytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
		void *p = mmap(0, b, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
		if (p != MAP_FAILED) { madvise(p, b, MADV_HUGEPAGE); return p; }
	}
	return malloc(bytes);
}
int main(int argc, char **argv)
{
	size_t n = argc > 1 ? (size_t)atol(argv[1]) : 100000000, i, steps = 0, cnt[256] = { 0 }, start[257];
	int nb = argc > 2 ? atoi(argv[2]) : 16, form = argc > 3 ? atoi(argv[3]) : 0, thp = argc > 4 ? atoi(argv[4]) : 0, k;
	double homef = 0.07, t0, t1, tf0;
	T *a; uint8_t *dig; bk_t b[256];
	tf0 = now();
	a = (T*)big(n * 8, thp); dig = (uint8_t*)big(n + 64, thp);
	for (i = 0; i < n; ++i) dig[i] = (xr() % 1000) < homef * 1000 ? (uint8_t)((double)i / n * nb) : (uint8_t)(xr() % nb);
	for (i = 0; i < n; ++i) ++cnt[dig[i]];
	memset(dig + n, 0, 64);
	for (i = 0; i < n; ++i) a[i] = ((uint64_t)dig[i] << 56) | i;
	start[0] = 0;
	for (k = 0; k < 256; ++k) { start[k + 1] = start[k] + cnt[k]; b[k].head = start[k]; b[k].nd = dig[start[k]]; }
	t0 = now();
	if (form == 0 || form == 1 || form == 6) {
		for (k = 0; k < 256;) {
			unsigned d;
			if (b[k].head == start[k + 1]) { ++k; continue; }
			d = b[k].nd;
			if (d == (unsigned)k) { ++b[k].head; b[k].nd = dig[b[k].head]; continue; }
			{
				T carry = a[b[k].head];
				do {
					const size_t slot = b[d].head; const unsigned dn = b[d].nd;
					b[d].head = slot + 1; b[d].nd = dig[slot + 1];
					if (form == 6) __builtin_prefetch(dig + slot + 129, 0, 3);
					if (form != 1) { const T ev = a[slot]; a[slot] = carry; __builtin_prefetch((char*)&a[slot] + 256, 1, 3); carry = ev; }
					d = dn; ++steps;
				} while (d != (unsigned)k);
				if (form != 1) a[b[k].head] = carry;
				b[k].nd = dig[++b[k].head];
			}
		}
	} else {
		rw_t q[256]; uint8_t nd[256]; size_t left; T carry; unsigned d; const size_t last = n - 1;
		for (k = 0; k < 256; ++k) q[k].r = q[k].w = start[k], nd[k] = dig[start[k]];
		for (k = 0; k < 256 && start[k + 1] == start[k]; ++k) {}
		left = start[k + 1] - q[k].w; carry = a[q[k].w]; d = nd[k]; q[k].r = q[k].w + 1; nd[k] = dig[q[k].r];
		for (;;) {
			const size_t rd = q[d].r, wd = q[d].w; const unsigned dn = nd[d]; const T ev = a[rd < last ? rd : last];
			a[wd] = carry; q[d].r = rd + 1; q[d].w = wd + 1; nd[d] = dig[rd + 1];
			if (form == 5) __builtin_prefetch(dig + rd + 129, 0, 3);
			__builtin_prefetch((char*)&a[rd] + 256, 1, 3);
			carry = ev; left -= d == (unsigned)k; ++steps; d = dn;
			if (left == 0) {
				do ++k; while (k < 256 && q[k].w == start[k + 1]);
				if (k == 256) break;
				left = start[k + 1] - q[k].w; carry = a[q[k].w]; d = nd[k]; q[k].r = q[k].w + 1; nd[k] = dig[q[k].r];
			}
		}
	}
	t1 = now();
	printf("form %d  buckets %3d  thp %d  n %zu: fill %.2f s, walk %.3f s = %.2f ns/element (%zu chain steps, check %llx)\n", form, nb, thp, n, t0 - tf0, t1 - t0, (t1 - t0) / n * 1e9, steps,
	       (unsigned long long)(a[n / 2] ^ a[n / 3]));
	return 0;
}
