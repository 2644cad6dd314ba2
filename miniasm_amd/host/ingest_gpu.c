/* ingest_gpu.c -- the text part of ma_hit_read (reference hit.c:70-101, paf.c, sdict.c) on the device.
 *
 * The host only moves bytes: a plain file goes from the page cache straight into pinned staging slots and on to
 * HBM (mahip_paf_load_fd); gzip / stdin input is inflated into memory first (zlib, as the reference does through
 * gzread) and uploaded.  Lines, columns, numbers, the span/match filter, the name dictionary with the reference's
 * first-appearance ids and the (mirrored) hit records are all produced by csrc/paf.hip; what comes back is the
 * dictionary (names + first-seen lengths, R entries) and, only for the per-symbol ABI, the records.
 * The -R pre-filter (ma_hit_no_cont, hit.c:38-68) rides in the same parse: the exclusion is a flag per name.
 * MA_HOST_PARSE=1 forces the host reader (ingest_mt.c / paf_reader.c); both are pinned to the same records and ids.
 */
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include "ma_host.h"

#define GPU(call) do { if ((call) != 0) ma_gpu_fail(__func__); } while (0)
/* the text stage may simply not fit (text + 61 B of columns per line + the records): report -2 and let the caller fall
 * back to the host reader, which streams the file and only needs the records on the device */
#define GPU_SOFT(call) do { if ((call) != 0) { fprintf(stderr, "[W::%s] device-side parse not possible (%s); using the host reader\n", __func__, mahip_strerror()); mahip_paf_release(c); return -2; } } while (0)

static char *slurp_gz(gzFile fp, size_t *len)
{
	size_t n = 0, m = 64u << 20;
	char *buf = (char*)malloc(m);
	for (;;) {
		int got;
		if (m - n < (16u << 20)) { m += m >> 1; buf = (char*)realloc(buf, m); }
		got = gzread(fp, buf + n, (unsigned)((m - n) < (1u << 30) ? (m - n) : (1u << 30)));
		if (got <= 0) break;
		n += (size_t)got;
	}
	*len = n;
	return buf;
}

int ma_gpu_parse_enabled(void)
{
	const char *s = getenv("MA_HOST_PARSE");
	return !(s && atoi(s) != 0);
}

/* parse the text already loaded into the context (mahip_paf_load_*): records stay on the device, the dictionary is
 * rebuilt in d (which must be empty or a previous result of this function); release = free the text afterwards */
int ma_hit_ingest_loaded(mahip_ctx_t *c, int min_span, int min_match, sdict_t *d, size_t *n_hits, int bi_dir, int release)
{
	return ma_hit_ingest_loaded_excl(c, min_span, min_match, d, n_hits, bi_dir, release, 0, 0, 0.f);
}

/* no_cont: -R, the reference's Step 0 (hit.c:38-68) folded into the same parse; prints its log line first */
int ma_hit_ingest_loaded_excl(mahip_ctx_t *c, int min_span, int min_match, sdict_t *d, size_t *n_hits, int bi_dir, int release, int no_cont, int max_hang, float int_frac)
{
	const int timing = getenv("MA_PIPE_TIMING") != 0;
	double t1 = sys_realtime(), t2, t3;
	mahip_paf_info_t info;
	size_t tot_len = 0;
	GPU(mahip_set_shard(c, 0, 0xffffffffu));
	GPU_SOFT(mahip_paf_parse_excl(c, min_span, min_match, bi_dir, no_cont, max_hang, int_frac, &info));
	if (no_cont) {
		if (ma_verbose >= 3) fprintf(MA_LOG, "[M::%s::%s] dropped %d contained reads\n", "ma_hit_no_cont", sys_timestamp(), info.n_excl);
		fprintf(MA_LOG, "[M::%s] ===> Step 1: reading read mappings <===\n", "main");
	}
	t2 = sys_realtime();
	/* the dictionary: the names in one block (the dictionary's arena) and the sd_seq_t records the device wrote for that block -- two copies, no
	 * per-name work on the host */
	{
		char *names = 0;
		sd_seq_t *seq = 0;
		uint64_t tl = 0;
		if (!ma_sd_recycle(d, info.name_bytes, info.n_seq, &names, &seq)) { /* (the same dictionary filled again keeps its blocks) */
			names = (char*)ma_big_alloc(info.name_bytes ? info.name_bytes : 1);
			seq = (sd_seq_t*)ma_big_alloc(((size_t)info.n_seq + 1) * sizeof(sd_seq_t));
		}
		GPU(mahip_paf_seqs(c, names, seq, &tl));
		ma_sd_adopt(d, names, info.name_bytes, info.n_seq, seq);
		tot_len = (size_t)tl;
	}
	if (release) GPU(mahip_paf_release(c));
	t3 = sys_realtime();
	if (ma_verbose >= 3)
		fprintf(MA_LOG, "[M::%s::%s] read %ld hits; stored %ld hits and %d sequences (%ld bp)\n", "ma_hit_read", sys_timestamp(), (long)info.n_records, (long)info.n_hits, d->n_seq, (long)tot_len);
	if (timing) fprintf(stderr, "[T::ingest_gpu] parse %.3f  dictionary%s %.3f s (%lu lines)\n", t2 - t1, release ? "+release" : "", t3 - t2, (unsigned long)info.n_lines);
	*n_hits = (size_t)info.n_hits;
	return 0;
}

/* file -> HBM; 0 ok, -1 = could not open */
int ma_paf_load_file(mahip_ctx_t *c, const char *fn)
{
	int fd = -1, is_plain = 0;
	struct stat st;
	if (fn && strcmp(fn, "-") != 0) {
		unsigned char magic[2] = { 0, 0 };
		fd = open(fn, O_RDONLY);
		if (fd < 0) return -1;
		if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) {
			ssize_t r = pread(fd, magic, 2, 0);
			is_plain = !(r == 2 && magic[0] == 0x1f && magic[1] == 0x8b);
		}
	}
	if (is_plain) {
		if (mahip_paf_load_fd(c, fd, (size_t)st.st_size) != 0) { close(fd); fprintf(stderr, "[W::%s] device-side parse not possible (%s); using the host reader\n", __func__, mahip_strerror()); mahip_paf_release(c); return -2; }
		close(fd);
	} else {
		gzFile fp = fd >= 0 ? gzdopen(fd, "r") : gzdopen(fileno(stdin), "r");
		size_t len = 0;
		char *buf;
		if (fp == 0) { if (fd >= 0) close(fd); return -1; }
		gzbuffer(fp, 1u << 20);
		buf = slurp_gz(fp, &len);
		gzclose(fp);
		if (mahip_paf_load_mem(c, buf, len) != 0) { free(buf); fprintf(stderr, "[W::%s] device-side parse not possible (%s); using the host reader\n", __func__, mahip_strerror()); mahip_paf_release(c); return -2; }
		free(buf);
	}
	return 0;
}

/* returns 0 and leaves the unsorted records in the context (as after mahip_hits_upload); -1 = could not open;
 * -2 = the device-side stage could not run (memory): nothing was consumed, the caller may use the host reader */
int ma_hit_ingest_gpu(mahip_ctx_t *c, const char *fn, int min_span, int min_match, sdict_t *d, size_t *n_hits, int bi_dir)
{
	return ma_hit_ingest_gpu_excl(c, fn, min_span, min_match, d, n_hits, bi_dir, 0, 0, 0.f);
}

int ma_hit_ingest_gpu_excl(mahip_ctx_t *c, const char *fn, int min_span, int min_match, sdict_t *d, size_t *n_hits, int bi_dir, int no_cont, int max_hang, float int_frac)
{
	const int timing = getenv("MA_PIPE_TIMING") != 0;
	double t0 = sys_realtime();
	{
		int rc = ma_paf_load_file(c, fn);
		if (rc != 0) return rc; /* -1 cannot open, -2 does not fit */
	}
	if (timing) fprintf(stderr, "[T::ingest_gpu] load %.3f s\n", sys_realtime() - t0);
	return ma_hit_ingest_loaded_excl(c, min_span, min_match, d, n_hits, bi_dir, 1, no_cont, max_hang, int_frac);
}
