/* paf.c -- PAF text reader (reference paf.h:30-32, paf.c:9-67; line semantics of kseq.h:101-150).
 *
 * Behaviour kept from the reference: plain / gzip / stdin ("-") input through zlib; a record is a line
 * split on TABs; numeric columns go through strtol (so signs, leading blanks and junk behave the same)
 * and are truncated to uint32 (ml to 31 bits); rev = first char of column 5 is '-'; a trailing CR is
 * dropped when the line is longer than one char; lines with fewer than 10 columns are skipped silently;
 * with exactly 10 columns `bl` is left untouched (the caller's record keeps its previous value).
 * The buffering is our own (1 MiB chunks, lines parsed in place, no per-line copy unless a line straddles
 * two chunks).
 */
#include <zlib.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "miniasm_amd.h"
#include "ma_host.h"

#define PAF_CHUNK (1u << 20)

typedef struct {
	gzFile fp;
	char *buf;          /* chunk + 1 byte for a terminator */
	size_t beg, end;    /* unread bytes are buf[beg,end) */
	int eof;
} paf_stream_t;

paf_file_t *paf_open(const char *fn)
{
	gzFile fp = fn && strcmp(fn, "-") ? gzopen(fn, "r") : gzdopen(fileno(stdin), "r");
	paf_stream_t *s;
	paf_file_t *pf;
	if (fp == 0) return 0;
	gzbuffer(fp, 1u << 18);
	s = (paf_stream_t*)calloc(1, sizeof(paf_stream_t));
	s->fp = fp;
	s->buf = (char*)malloc(PAF_CHUNK + 1);
	pf = (paf_file_t*)calloc(1, sizeof(paf_file_t));
	pf->fp = s;
	return pf;
}

int paf_close(paf_file_t *pf)
{
	paf_stream_t *s;
	if (pf == 0) return 0;
	s = (paf_stream_t*)pf->fp;
	gzclose(s->fp);
	free(s->buf); free(s);
	free(pf->buf.s);
	free(pf);
	return 0;
}

static inline uint32_t field_u32(char *q)
{ /* strtol semantics; all-digit short fields take the fast path */
	uint32_t x = 0;
	const char *p = q;
	int n = 0;
	while ((unsigned)(*p - '0') < 10u && n < 9) x = x * 10 + (uint32_t)(*p - '0'), ++p, ++n;
	if (*p == 0 && n > 0) return x;
	return (uint32_t)strtol(q, 0, 10);
}

/* parse one NUL-terminated line of length l in place (reference paf.c:34-56); <0 if fewer than 10 columns */
int ma_paf_parse_line(int l, char *s, paf_rec_t *pr)
{
	char *q = s;
	int i, t = 0;
	for (i = 0; i <= l; ++i) {
		if (i < l && s[i] != '\t') continue;
		s[i] = 0;
		switch (t) {
		case 0: pr->qn = q; break;
		case 1: pr->ql = field_u32(q); break;
		case 2: pr->qs = field_u32(q); break;
		case 3: pr->qe = field_u32(q); break;
		case 4: pr->rev = (*q == '-'); break;
		case 5: pr->tn = q; break;
		case 6: pr->tl = field_u32(q); break;
		case 7: pr->ts = field_u32(q); break;
		case 8: pr->te = field_u32(q); break;
		case 9: pr->ml = field_u32(q) & 0x7fffffffu; break;
		case 10: pr->bl = field_u32(q); break;
		default: break;
		}
		++t;
		q = i < l ? &s[i + 1] : 0;
	}
	return t < 10 ? -1 : 0;
}

/* the reference's own name for it (paf.c:34; non-static there, so an object written against paf.o may bind it) */
int paf_parse(int l, char *s, paf_rec_t *pr) { return ma_paf_parse_line(l, s, pr); }

/* next raw line: returns its length (>=0) and a pointer valid until the next call, or -1 at end of input */
static int next_line(paf_file_t *pf, char **line)
{
	paf_stream_t *s = (paf_stream_t*)pf->fp;
	size_t carried = 0;
	if (s->beg >= s->end && s->eof) return -1;
	for (;;) {
		char *nl;
		if (s->beg >= s->end) {
			int got;
			if (s->eof) break;
			got = gzread(s->fp, s->buf, PAF_CHUNK);
			if (got < 0) got = 0;
			s->beg = 0; s->end = (size_t)got;
			if ((unsigned)got < PAF_CHUNK) s->eof = 1;
			if (got == 0) break;
		}
		nl = (char*)memchr(s->buf + s->beg, '\n', s->end - s->beg);
		if (nl && carried == 0) { /* whole line inside the chunk: hand it out in place */
			size_t l = (size_t)(nl - (s->buf + s->beg));
			*line = s->buf + s->beg;
			s->beg += l + 1;
			if (l > 1 && (*line)[l - 1] == '\r') --l;
			(*line)[l] = 0;
			return (int)l;
		} else { /* straddles chunks (or is the unterminated tail): accumulate in pf->buf */
			size_t piece = (nl ? (size_t)(nl - (s->buf + s->beg)) : s->end - s->beg);
			if (pf->buf.m < carried + piece + 1) {
				pf->buf.m = (carried + piece + 1) * 2;
				pf->buf.s = (char*)realloc(pf->buf.s, pf->buf.m);
			}
			memcpy(pf->buf.s + carried, s->buf + s->beg, piece);
			carried += piece;
			s->beg += piece + (nl ? 1 : 0);
			if (nl) break;
			if (carried == 0 && s->eof && s->beg >= s->end) { /* nothing left at all */ }
		}
	}
	if (pf->buf.s == 0) { pf->buf.m = 1; pf->buf.s = (char*)calloc(1, 1); }
	if (carried > 1 && pf->buf.s[carried - 1] == '\r') --carried;
	pf->buf.s[carried] = 0;
	pf->buf.l = carried;
	*line = pf->buf.s;
	return (int)carried;
}

int paf_read(paf_file_t *pf, paf_rec_t *r)
{
	for (;;) {
		char *line;
		int l = next_line(pf, &line);
		if (l < 0) return l;
		if (ma_paf_parse_line(l, line, r) >= 0) return 0;
	}
}
