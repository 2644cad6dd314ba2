// paf.hip -- PAF text -> hit records + read-name dictionary on the device.
//
// Reproduces, bit for bit, what the reference's reader and ma_hit_read produce before the sort
// (paf.c:34-67 paf_parse/paf_read, kseq.h line semantics, sdict.c:27-45 sd_put, hit.c:70-101):
//   * a record is a line split on TABs; a trailing CR is dropped when the line is longer than one char;
//     lines with fewer than 10 columns are skipped; with exactly 10 columns `bl` keeps the value of the last
//     line that had an 11th column (0 before any);
//   * numeric columns follow strtol (leading blanks, sign, junk after the digits, saturation at LONG_MAX/MIN)
//     and are truncated to 32 bits (ml, bl to 31); rev = first char of column 5 is '-';
//   * a line is stored when both spans >= min_span and ml >= min_match (hit.c:85); its query name, then its target
//     name, enter the dictionary: ids are dense in order of FIRST APPEARANCE, the first length seen wins;
//   * every stored line yields the hit and, with bi_dir and qid != tid, the mirrored hit right after it.
// The sequential parts of the reference become data-parallel as follows: line starts = positions after '\n'
// (count, scan, scatter); first-appearance ids = hash-table insert with an atomic MIN of the occurrence number
// (2*line + column) per distinct name, then a sort of the distinct names by that minimum; the stale `bl` = a
// compaction of the lines that have one plus a prefix count; record slots = prefix sum of 1-or-2 per line.
// All HBM-bound byte/integer work: text is read once (through LDS tiles), per-line columns are 61 B.
#include "mahip_internal.hpp"
#include <time.h>

#define PAF_TILE 4096u            // bytes per block in the newline passes (256 threads x 16 B)
#define PAF_LDS_BYTES 49152u      // text of 256 consecutive lines is staged in LDS when it fits
#define PAF_PROBE_LIMIT 2048u
#define PAF_EMPTY 0xffffffffffffffffull

// counter slots used by this file (aliases into ctx->ctr)
#define PC_LINES CT_TOTAL
#define PC_VALID CT_LIVE
#define PC_PASS CT_REMAIN
#define PC_NOBL CT_OVF
#define PC_MAXQS CT_MAXQS
#define PC_OVERFLOW CT_OVF2
#define PC_HITS CT_NRED
#define PC_DISTINCT CT_NMULTI

struct PafBufs {
	DevBuf text, lstart, tile;
	DevBuf glast, gmax, tfirst; // tile parser: last newline per granule / per group of granules, first line start per tile
	DevBuf flags, num[8], tnoff, qlen, tlen, hq, ht, qslot, tslot;
	DevBuf tab, tmin, info, slot_id, blv, scal, excl;
	DevBuf name_off, name_len, name_pos, seq_len, names;
	size_t nbytes = 0, name_bytes = 0;
	uint32_t n_seq = 0;
	bool loaded = false;
};

struct PafCols {
	uint8_t *flags;            // bit0 valid (>= 10 columns), bit1 stored, bit2 has column 11, bit3 rev
	uint32_t *ql, *qs, *qe, *tl, *ts, *te, *ml, *bl;
	uint32_t *tnoff, *qlen, *tlen; // target-name offset inside the line, name lengths (up to the first NUL)
	uint64_t *hq, *ht;
	uint32_t *qslot, *tslot;   // hash-table slot of the two names, later their ids
};

static PafBufs *paf_of(mahip_ctx *c)
{
	if (!c->paf) c->paf = new PafBufs();
	return (PafBufs*)c->paf;
}

void paf_free(mahip_ctx *c)
{
	PafBufs *b = (PafBufs*)c->paf;
	if (!b) return;
	DevBuf *all[] = { &b->text, &b->lstart, &b->tile, &b->glast, &b->gmax, &b->tfirst, &b->flags, &b->tnoff, &b->qlen, &b->tlen, &b->hq, &b->ht, &b->qslot, &b->tslot, &b->tab, &b->tmin, &b->info,
		&b->slot_id, &b->blv, &b->scal, &b->excl, &b->name_off, &b->name_len, &b->name_pos, &b->seq_len, &b->names };
	for (DevBuf *d : all) dev_free(c, *d);
	for (int k = 0; k < 8; ++k) dev_free(c, b->num[k]);
	delete b;
	c->paf = nullptr;
}

// ------------------------------------------------------------------------------------------------ line starts

__device__ __forceinline__ uint32_t nl_mask(uint32_t v) // 0x80 in every byte of v that equals '\n'
{
	v ^= 0x0A0A0A0Au;
	uint32_t t = (v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
	return ~(t | v | 0x7F7F7F7Fu);
}

__device__ __forceinline__ uint4 load16(const unsigned char *__restrict__ text, size_t off, size_t n)
{
	if (off + 16 <= n) return *(const uint4*)(text + off);
	uint32_t w[4] = { 0, 0, 0, 0 };
	for (int k = 0; k < 16; ++k) if (off + k < n) w[k >> 2] |= (uint32_t)text[off + k] << (8 * (k & 3));
	return make_uint4(w[0], w[1], w[2], w[3]);
}

__global__ __launch_bounds__(256) void k_paf_nl_count(const unsigned char *__restrict__ text, size_t n, uint32_t *__restrict__ tile_cnt)
{
	__shared__ uint32_t s_w[4];
	size_t off = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
	uint32_t cnt = 0;
	if (off < n) {
		uint4 v = load16(text, off, n);
		cnt = __popc(nl_mask(v.x)) + __popc(nl_mask(v.y)) + __popc(nl_mask(v.z)) + __popc(nl_mask(v.w));
	}
	cnt = wv_sum_u32(cnt);
	if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = cnt;
	__syncthreads();
	if (threadIdx.x == 0) tile_cnt[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// lstart[k+1] = byte after the k-th newline; lstart[0] = 0; an unterminated last line gets the sentinel n+1
__global__ __launch_bounds__(256) void k_paf_nl_pos(const unsigned char *__restrict__ text, size_t n, const uint32_t *__restrict__ tile_off,
                                                     const uint32_t *__restrict__ d_total, uint64_t *__restrict__ lstart, unsigned long long *__restrict__ ctr)
{
	__shared__ uint32_t s_w[4];
	size_t off = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
	uint32_t m[4] = { 0, 0, 0, 0 }, cnt = 0, tot;
	if (off < n) {
		uint4 v = load16(text, off, n);
		m[0] = nl_mask(v.x); m[1] = nl_mask(v.y); m[2] = nl_mask(v.z); m[3] = nl_mask(v.w);
		cnt = __popc(m[0]) + __popc(m[1]) + __popc(m[2]) + __popc(m[3]);
	}
	uint32_t k = tile_off[blockIdx.x] + block_excl_scan_256(cnt, s_w, &tot);
	for (int w = 0; w < 4; ++w)
		for (uint32_t x = m[w]; x; x &= x - 1) {
			int byte = (__ffs(x) - 1) >> 3;
			lstart[++k] = off + (size_t)(w * 4 + byte) + 1;
		}
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		uint32_t nl = *d_total;
		int open = n > 0 && text[n - 1] != '\n';
		lstart[0] = 0;
		if (open) lstart[(size_t)nl + 1] = (uint64_t)n + 1;
		ctr[PC_LINES] = (unsigned long long)nl + (unsigned long long)open;
	}
}

// ------------------------------------------------------------------------------------------------ per-line parse

#define FNV_OFF 0xcbf29ce484222325ull
#define FNV_PRIME 0x100000001b3ull

// strtol(base 10) of the bytes [beg, end) of a column, truncated to 32 bits (paf.c:41-52 through the (uint32_t) casts)
template <typename PTR>
__device__ __forceinline__ uint32_t paf_num(PTR &p, uint32_t beg, uint32_t end)
{
	uint32_t pos = beg, nd = 0;
	unsigned ch = 0;
	for (; pos < end; ++pos) { ch = p[pos]; if (!(ch == ' ' || (ch >= 0x0bu && ch <= 0x0du))) break; } // leading blanks (TAB / LF cannot occur inside a column)
	bool neg = false, ovf = false;
	if (pos < end && (ch == '+' || ch == '-')) { neg = ch == '-'; ++pos; }
	uint64_t acc = 0;
	for (; pos < end; ++pos, ++nd) {
		const unsigned d = (unsigned)p[pos] - '0';
		if (d >= 10u) break; // junk (or a NUL) ends the number
		if (nd < 18) acc = acc * 10 + d; // 18 digits cannot overflow 63 bits
		else {
			const uint64_t lim = neg ? 0x8000000000000000ull : 0x7fffffffffffffffull;
			if (ovf || acc > (lim - d) / 10) ovf = true; else acc = acc * 10 + d;
		}
	}
	return ovf ? (neg ? 0u : 0xffffffffu) : (uint32_t)(neg ? (uint64_t)0 - acc : acc); // LONG_MIN -> 0, LONG_MAX -> 0xffffffff
}

// ---- name keys.  A name of 1..8 bytes IS its key (the bytes, zero-padded: names are C strings, so no byte of them is 0 and the padded word is unique);
// any other name (empty, or longer) gets a 64-bit hash of its length and its 8-byte words.  A file whose names are all short therefore has an EXACT
// dictionary without a single text comparison (k_dict_insert_short); as soon as one name is long the table is keyed by the mixed key and every probe
// compares the text (k_dict_insert).  key_mix spreads either kind over the table.
#define KEY_SEED 0x9e3779b97f4a7c15ull
#define KEY_LENMUL 0xff51afd7ed558ccdull
#define KEY_MUL 0xd6e8feb86659fd93ull
__host__ __device__ __forceinline__ uint64_t key_mix(uint64_t z) { z ^= z >> 33; z *= 0xff51afd7ed558ccdull; z ^= z >> 33; z *= 0xc4ceb9fe1a85ec53ull; z ^= z >> 33; return z; }
__host__ __device__ __forceinline__ uint64_t key_step(uint64_t h, uint64_t w) { h = (h ^ w) * KEY_MUL; return h ^ (h >> 29); }
__host__ __device__ __forceinline__ bool key_is_short(uint32_t len) { return len - 1u < 8u; }

// key of a name column up to its first NUL (the reference handles names as C strings); *len = its bytes.  Byte-wise form (any PTR): the slow paths
template <typename PTR>
__device__ __forceinline__ uint64_t paf_name(PTR &p, uint32_t beg, uint32_t end, uint32_t *len)
{
	uint32_t pos = beg;
	for (; pos < end; ++pos) if (p[pos] == 0) break;
	const uint32_t l = pos - beg;
	*len = l;
	uint64_t h = KEY_SEED ^ ((uint64_t)l * KEY_LENMUL), w = 0;
	for (uint32_t k = 0; k < l; ++k) {
		w |= (uint64_t)p[beg + k] << (8 * (k & 7));
		if ((k & 7) == 7 || k + 1 == l) { if (l <= 8) return w; h = key_step(h, w); w = 0; }
	}
	return h;
}

// the column starts of the block's 256 lines in LDS, column-major: lane i's k-th entry sits at [k][i], so a wave's accesses never share a bank
struct FsCols {
	uint32_t *b;
	__device__ __forceinline__ uint32_t &operator[](uint32_t k) const { return b[k * 256u]; }
};
// column starts: fs[k] = position after the k-th TAB of the line (k < 12); returns the number of TABs
template <typename PTR>
__device__ __forceinline__ uint32_t paf_tabs(PTR &p, uint32_t l, FsCols fs)
{
	uint32_t t = 0;
	fs[0] = 0;
	for (uint32_t pos = 0; pos < l; ++pos)
		if (p[pos] == '\t') { ++t; if (t < 12) fs[t] = pos + 1; }
	return t;
}

struct PafLine { uint32_t valid, hasbl, rev, ql, qs, qe, tl, ts, te, ml, bl, tnoff, qlen, tlen; uint64_t hq, ht; };

// One line: first the column starts (one pass over the bytes, starts kept in LDS), then each column with straight-line
// code -- every lane is in the same routine at the same time, only the trip counts differ.
template <typename PTR>
__device__ __forceinline__ void paf_line(PTR p, uint32_t l, FsCols fs /* LDS, 12 entries */, PafLine &o)
{
	if (l > 1 && p[l - 1] == '\r') --l;
	uint32_t t = paf_tabs(p, l, fs);
	++t; // columns
	if (t < 12) fs[t] = l + 1; // end sentinel: column k is [fs[k], fs[k+1] - 1)
	o.valid = t >= 10; o.hasbl = t >= 11;
	o.rev = 0; o.ql = o.qs = o.qe = o.tl = o.ts = o.te = o.ml = o.bl = o.tnoff = o.qlen = o.tlen = 0; o.hq = o.ht = 0;
	if (!o.valid) return; // the reference fills the columns it finds and drops the record: nothing of it is ever used
	o.hq = paf_name(p, fs[0], fs[1] - 1, &o.qlen);
	o.ql = paf_num(p, fs[1], fs[2] - 1);
	o.qs = paf_num(p, fs[2], fs[3] - 1);
	o.qe = paf_num(p, fs[3], fs[4] - 1);
	o.rev = fs[4] < fs[5] - 1 && p[fs[4]] == '-';
	o.tnoff = fs[5];
	o.ht = paf_name(p, fs[5], fs[6] - 1, &o.tlen);
	o.tl = paf_num(p, fs[6], fs[7] - 1);
	o.ts = paf_num(p, fs[7], fs[8] - 1);
	o.te = paf_num(p, fs[8], fs[9] - 1);
	o.ml = paf_num(p, fs[9], fs[10] - 1) & 0x7fffffffu;
	if (o.hasbl) o.bl = paf_num(p, fs[10], fs[11] - 1);
}

__global__ __launch_bounds__(256) void k_paf_parse(const unsigned char *__restrict__ text, size_t n, const uint64_t *__restrict__ lstart, uint32_t L,
                                                    int min_span, int min_match, PafCols o, uint32_t *__restrict__ f_hasbl, unsigned long long *__restrict__ ctr, uint32_t lds_bytes)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char s_text[]; // lds_bytes (+ slack); read and written 16 / 8 bytes at a time: sized by the host from the mean line length, so that several blocks fit a CU
	__shared__ uint32_t s_fs[256 * 12];
	const uint32_t i0 = blockIdx.x * 256u, i1 = i0 + 256u < L ? i0 + 256u : L;
	const uint64_t b0 = lstart[i0], e1 = lstart[i1] - 1; // bytes of these lines: [b0, e1)
	const uint64_t a0 = b0 & ~(uint64_t)15;
	const bool in_lds = e1 - a0 <= lds_bytes;
	if (in_lds) {
		for (uint64_t x = (uint64_t)threadIdx.x * 16; a0 + x < e1; x += 256 * 16) *(uint4*)(s_text + x) = load16(text, a0 + x, n);
		__syncthreads();
	}
	const uint32_t i = i0 + threadIdx.x;
	uint32_t valid = 0, pass = 0, nobl = 0;
	uint64_t mq = 0;
	if (i < i1) {
		const uint64_t ls = lstart[i];
		const uint32_t l = (uint32_t)(lstart[i + 1] - 1 - ls);
		PafLine r;
		const FsCols fs = { s_fs + threadIdx.x };
		if (in_lds) paf_line((const unsigned char*)(s_text + (ls - a0)), l, fs, r); // LDS byte reads
		else paf_line(text + ls, l, fs, r);                                              // oversized lines: straight from global memory
		valid = r.valid;
		pass = valid && !(r.qe - r.qs < (uint32_t)min_span || r.te - r.ts < (uint32_t)min_span || (int)r.ml < min_match); // hit.c:85
		nobl = valid && !r.hasbl;
		o.flags[i] = (uint8_t)(valid | pass << 1 | r.hasbl << 2 | r.rev << 3);
		f_hasbl[i] = r.hasbl;
		o.ql[i] = r.ql; o.qs[i] = r.qs; o.qe[i] = r.qe; o.tl[i] = r.tl; o.ts[i] = r.ts; o.te[i] = r.te; o.ml[i] = r.ml; o.bl[i] = r.bl;
		o.tnoff[i] = r.tnoff; o.qlen[i] = r.qlen; o.tlen[i] = r.tlen; o.hq[i] = r.hq; o.ht[i] = r.ht;
		if (pass) mq = r.qs > r.ts ? r.qs : r.ts;
	}
	blk_add_u64(&ctr[PC_VALID], valid);
	blk_add_u64(&ctr[PC_PASS], pass);
	blk_add_u64(&ctr[PC_NOBL], nobl);
	blk_max_u64(&ctr[PC_MAXQS], mq);
}

// the `bl` a 10-column line inherits: value of the nearest earlier line with an 11th column (paf.c leaves the field alone)
__global__ __launch_bounds__(256) void k_paf_bl_compact(const uint32_t *__restrict__ f_hasbl, const uint32_t *__restrict__ pos, const uint32_t *__restrict__ bl, uint32_t L, uint32_t *__restrict__ blv)
{
	uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < L && f_hasbl[i]) blv[pos[i]] = bl[i];
}
__global__ __launch_bounds__(256) void k_paf_bl_fill(const uint32_t *__restrict__ f_hasbl, const uint32_t *__restrict__ pos, const uint32_t *__restrict__ blv, uint32_t L, uint32_t *__restrict__ bl)
{
	uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < L && !f_hasbl[i]) bl[i] = pos[i] ? blv[pos[i] - 1] : 0u;
}

// ------------------------------------------------------------------------------------------------ tile parser (round 6)
// The lane-per-line parser above walks every byte of its line twice in divergent loops: 2 100 scalar + 1 340 vector instructions per 64 lines, bound by the
// CU's ONE scalar unit (profiles/r04_sq_counters_cfg4.txt: 12.6 ms per 100 M lines).  This one has no per-byte loop at all:
//   * the text is cut into tiles of K KiB (K from the mean line length: about 240 lines a tile); a tile owns the lines that END in it and keeps the up to
//     960 bytes in front of it in LDS as well, where its first line may start;
//   * while the 16-byte pieces travel from HBM to LDS their TABs and newlines are found by SWAR compares and leave as one BIT per byte (two 2-KiB arrays);
//   * newline ranks (a block scan of popcounts) give every line of the tile a lane; the lane finds its 11 column ends with count-trailing-zeros on a
//     64-bit window of the TAB bits and converts each number from the 8 bytes at its start: XOR '0', shift the digits to the top, check "all <= 9" and fold
//     them with three multiply-adds per half -- straight-line code, every lane in step;
//   * anything the straight line does not cover -- a sign, blanks, junk behind the digits, more than 8 digits, an empty column, a NUL in a name, a name of more
//     than 64 bytes, a line that starts more than 960 bytes in front of its tile -- marks the line ODD, and k_paf_parse_odd runs the byte-wise routine above
//     on exactly those lines (the reference's strtol semantics live there, and only there).
// Line i is the line that ends with the i-th newline; a text without a final newline gets a virtual one at position n.
#define PAF_GRAN 1024u        // granule of the newline census: one wave, 64 x 16 bytes
#define PAF_OVER 960u         // bytes in front of a tile that are staged with it
#define PAF_BGRP 12           // log2 of the granules per group of the "last newline" look-up
#define PF_ODD 0x80u          // flags: the line waits for k_paf_parse_odd
#define PF_QCONT 0x10u        // flags: stored line whose query name (short or long) equals that of the stored line in front of it
#define PC_LONG CT_NSHORT     // stored lines with a name that is not 1..8 bytes
#define PC_ODD CT_NASYMM

__device__ __forceinline__ uint32_t eq_mask(uint32_t v, uint32_t c4) { v ^= c4; const uint32_t t = (v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu; return ~(t | v | 0x7F7F7F7Fu); } // 0x80 in every byte of v that equals c
// the flags of two masks as one byte: newline flags -> bits 0..3, TAB flags -> bits 4..7 (one multiply gathers both: the partial products land on 32 distinct bits)
__device__ __forceinline__ uint32_t nib8(uint32_t m_nl, uint32_t m_tab) { return (((m_nl >> 7) | (m_tab >> 3)) * 0x204081u) >> 21 & 0xFFu; }
// TAB bits (low half) and newline bits (high half) of a 16-byte piece
__device__ __forceinline__ uint32_t masks16(const uint4 v)
{
	const uint32_t a = nib8(eq_mask(v.x, 0x0A0A0A0Au), eq_mask(v.x, 0x09090909u)), b = nib8(eq_mask(v.y, 0x0A0A0A0Au), eq_mask(v.y, 0x09090909u));
	const uint32_t c = nib8(eq_mask(v.z, 0x0A0A0A0Au), eq_mask(v.z, 0x09090909u)), d = nib8(eq_mask(v.w, 0x0A0A0A0Au), eq_mask(v.w, 0x09090909u));
	const uint32_t nl = (a & 15u) | (b & 15u) << 4 | (c & 15u) << 8 | (d & 15u) << 12, tb = a >> 4 | (b >> 4) << 4 | (c >> 4) << 8 | (d >> 4) << 12;
	return tb | nl << 16;
}

// granule g = bytes [1024 g, 1024 g + 1024), one wave: cnt[g] = its newlines, last[g] = offset of the byte behind its last newline (0: it has none).
// (The text ends with a newline -- the host appends one to a text that does not -- and is followed by zeros up to the next 16 bytes.)
#define PAF_GC_PER_WAVE 8u
__global__ __launch_bounds__(256) void k_paf_gran_count(const unsigned char *__restrict__ text, size_t n, uint32_t n_gran, uint32_t *__restrict__ cnt, uint32_t *__restrict__ last)
{ // a wave takes PAF_GC_PER_WAVE consecutive granules and has all their loads in flight at once (one granule per wave: 5.9 M waves of one load each, 1.4 ms for 6 GB)
	const uint32_t g0 = (blockIdx.x * 4u + (threadIdx.x >> 6)) * PAF_GC_PER_WAVE, lane = threadIdx.x & 63;
	if (g0 >= n_gran) return; // the whole wave
	uint4 v[PAF_GC_PER_WAVE];
#pragma unroll
	for (uint32_t k = 0; k < PAF_GC_PER_WAVE; ++k) {
		const size_t off = (size_t)(g0 + k) * PAF_GRAN + lane * 16u;
		v[k] = off < n ? *(const uint4*)(text + off) : make_uint4(0, 0, 0, 0);
	}
#pragma unroll
	for (uint32_t k = 0; k < PAF_GC_PER_WAVE; ++k) {
		const uint32_t g = g0 + k;
		if (g >= n_gran) break; // the whole wave
		const uint32_t mx = eq_mask(v[k].x, 0x0A0A0A0Au), my = eq_mask(v[k].y, 0x0A0A0A0Au), mz = eq_mask(v[k].z, 0x0A0A0A0Au), mw = eq_mask(v[k].w, 0x0A0A0A0Au);
		const uint32_t c = wv_sum_u32((uint32_t)(__popc(mx) + __popc(my) + __popc(mz) + __popc(mw)));
		const unsigned long long has = __ballot((mx | my | mz | mw) != 0);
		if (lane == 0) { cnt[g] = c; if (!has) last[g] = 0; }
		if (has && (int)lane == 63 - __clzll((long long)has)) { // the lane with the granule's last newline: byte index of its highest flag + 1
			const uint32_t w = mw ? 3u : mz ? 2u : my ? 1u : 0u, m = mw ? mw : mz ? mz : my ? my : mx;
			last[g] = lane * 16u + w * 4u + ((31u - (uint32_t)__clz((int)m)) >> 3) + 1u;
		}
	}
}
// bmax[b] = byte behind the last newline of granules [4096 b, 4096 b + 4096) (0: none), one wave per group
__global__ __launch_bounds__(64) void k_paf_gran_bmax(const uint32_t *__restrict__ last, uint32_t n_gran, unsigned long long *__restrict__ bmax)
{
	const uint32_t b = blockIdx.x, lane = threadIdx.x;
	const long g_lo = (long)b << PAF_BGRP;
	long g_hi = g_lo + (1l << PAF_BGRP); if (g_hi > (long)n_gran) g_hi = (long)n_gran;
	unsigned long long r = 0;
	for (long top = g_hi; top > g_lo && r == 0; top -= 64) { // 64 granules per step, from the back
		const long g = top - 1 - (long)lane;
		const uint32_t l = g >= g_lo ? last[g] : 0u;
		const unsigned long long has = __ballot(l != 0);
		if (has) { const int src = __ffsll((long long)has) - 1; const uint32_t lv = wv_bcast(l, src); r = (unsigned long long)(top - 1 - src) * PAF_GRAN + lv; }
	}
	if (lane == 0) bmax[b] = r;
}
// first[t] = start of the first line that ends in tile t = byte behind the last newline in front of the tile (0: the text starts it)
__global__ __launch_bounds__(256) void k_paf_tile_first(const uint32_t *__restrict__ last, const unsigned long long *__restrict__ bmax, uint32_t K, uint32_t n_tiles, unsigned long long *__restrict__ first)
{
	const uint32_t t = blockIdx.x * 256u + threadIdx.x;
	if (t >= n_tiles) return;
	long g = (long)t * (long)K - 1;
	const long g_lo = g >= 0 ? (g >> PAF_BGRP) << PAF_BGRP : 0;
	unsigned long long r = 0;
	for (; g >= g_lo; --g) { const uint32_t l = last[g]; if (l) { r = (unsigned long long)g * PAF_GRAN + l; break; } }
	if (r == 0) for (long b = (g_lo >> PAF_BGRP) - 1; b >= 0; --b) if (bmax[b]) { r = bmax[b]; break; }
	first[t] = r;
}

struct TileArgs {
	const unsigned char *text; size_t n; // (ends with a newline, zeros behind it)
	uint32_t K, n_gran, n_tiles, L;
	const uint32_t *goff;            // [n_gran + 1] newlines in front of every granule
	const unsigned long long *first; // [n_tiles]
	int min_span, min_match;
};

// the 8 bytes at LDS offset pos (any alignment) as lo | hi << 32
__device__ __forceinline__ uint64_t lds_get8(const uint32_t *__restrict__ w, uint32_t pos)
{
	const uint32_t i = pos >> 2, sh = (pos & 3u) * 8u;
	const uint32_t w0 = w[i], w1 = w[i + 1], w2 = w[i + 2];
	const uint32_t lo = (uint32_t)(((uint64_t)w1 << 32 | w0) >> sh), hi = (uint32_t)(((uint64_t)w2 << 32 | w1) >> sh);
	return (uint64_t)hi << 32 | lo;
}
// distance from pos to the first TAB at or behind it: exact below 33 (two words of bits) / below 64 (three), at least that otherwise
__device__ __forceinline__ uint32_t tab_dist2(const uint32_t *__restrict__ tb, uint32_t pos)
{
	const uint32_t i = pos >> 5, sh = pos & 31u;
	const uint64_t x = ((uint64_t)tb[i + 1] << 32 | tb[i]) >> sh;
	return x ? (uint32_t)__builtin_ctzll(x) : 64u - sh; // (none among the 64 - sh >= 33 bits the two words hold)
}
__device__ __forceinline__ uint32_t tab_dist3(const uint32_t *__restrict__ tb, uint32_t pos)
{
	const uint32_t i = pos >> 5, sh = pos & 31u;
	uint64_t x = ((uint64_t)tb[i + 1] << 32 | tb[i]) >> sh;
	if (sh) x |= (uint64_t)tb[i + 2] << (64u - sh);
	return x ? (uint32_t)__builtin_ctzll(x) : 64u;
}
// a column of 1..8 digits at v (its first byte lowest) -> its value; *bad collects 0x80 flags of bytes that are not digits
__device__ __forceinline__ uint32_t num8(uint64_t v, uint32_t len, uint32_t *bad)
{
	v ^= 0x3030303030303030ull;            // '0'..'9' -> 0..9
	v <<= 8u * (8u - len);                 // the column's last digit into the top byte; what stood behind the column falls out, leading zeros come in
	uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
	*bad |= (lo + 0x76767676u) | lo | (hi + 0x76767676u) | hi; // a byte > 9 sets its top bit (a carry can only leave a byte that has set it already)
	lo = (lo & 0x00ff00ffu) * 10u + ((lo >> 8) & 0x00ff00ffu); // d0 d1 | d2 d3
	lo = (lo & 0xffffu) * 100u + (lo >> 16);
	hi = (hi & 0x00ff00ffu) * 10u + ((hi >> 8) & 0x00ff00ffu);
	hi = (hi & 0xffffu) * 100u + (hi >> 16);
	return lo * 10000u + hi;
}
// key of the name [pos, pos + len) in LDS (paf_name's function, on words); *nul: a NUL among its bytes
__device__ __forceinline__ uint64_t lds_name_key(const uint32_t *__restrict__ w, uint32_t pos, uint32_t len, uint32_t *nul)
{
	if (len <= 8) {
		const uint64_t m = len == 8 ? ~0ull : (1ull << (8u * len)) - 1ull;
		const uint64_t v = lds_get8(w, pos), f = v | ~m;
		*nul |= ((f - 0x0101010101010101ull) & ~f & 0x8080808080808080ull) != 0;
		return len ? (v & m) : KEY_SEED;
	}
	uint64_t h = KEY_SEED ^ ((uint64_t)len * KEY_LENMUL);
	for (uint32_t k = 0; k < len; k += 8) {
		const uint32_t rem = len - k;
		const uint64_t m = rem >= 8 ? ~0ull : (1ull << (8u * rem)) - 1ull;
		const uint64_t v = lds_get8(w, pos + k), f = v | ~m;
		*nul |= ((f - 0x0101010101010101ull) & ~f & 0x8080808080808080ull) != 0;
		h = key_step(h, v & m);
	}
	return h;
}

#ifndef PAF_TILE_WAVES
#define PAF_TILE_WAVES 1 // __launch_bounds__' second argument (waves per SIMD the register allocation has to leave room for): an experiment handle
#endif
template <int CH> // a block stages up to CH * 16 KiB of text: CH * 64 bytes per thread in the newline ranking
__global__ __launch_bounds__(256, PAF_TILE_WAVES) void k_paf_parse_tile(const TileArgs a, PafCols o, uint64_t *__restrict__ lstart, unsigned long long *__restrict__ ctr)
{
	constexpr uint32_t REG = (uint32_t)CH * 16384u;
	extern __shared__ __attribute__((aligned(16))) unsigned char s_text[]; // the text: REG + 32; behind it:
	uint16_t *s_tb = (uint16_t*)(s_text + REG + 32);        // REG / 16 + 8: TAB bits, one u16 per 16-byte piece
	uint16_t *s_nb = s_tb + REG / 16 + 8;                   // the same for newlines
	uint32_t *s_lend = (uint32_t*)(s_nb + REG / 16 + 8);    // [0] end of the line in front of the batch, [1 + j] newline of the batch's j-th line
	__shared__ uint32_t s_w[4];
	const uint32_t *tw = (const uint32_t*)s_text, *tb = (const uint32_t*)s_tb;
	const unsigned lane = threadIdx.x & 63;
	uint32_t c_valid = 0, c_pass = 0, c_nobl = 0, c_long = 0, c_odd = 0;
	uint64_t c_mq = 0;
	// A block works through tiles blockIdx.x, + gridDim.x, ...; the pieces of the NEXT tile are on their way from HBM (in registers) while the lines of this one are
	// parsed: the trip to memory is paid once per block, not once per tile.
	constexpr int NP = 4 * CH; // 16-byte pieces per thread
	struct Geom { uint32_t line0, tot, len; uint64_t first, lo; bool long_first; };
	auto geom = [&](uint32_t t) {
		Geom g;
		const uint32_t g0 = t * a.K, g1 = g0 + a.K < a.n_gran ? g0 + a.K : a.n_gran;
		g.line0 = a.goff[g0]; g.tot = a.goff[g1] - g.line0; // the lines that end in this tile
		const uint64_t tb0 = (uint64_t)g0 * PAF_GRAN, te0 = (uint64_t)g1 * PAF_GRAN;
		g.first = a.first[t];
		uint64_t lo = tb0 > PAF_OVER ? tb0 - PAF_OVER : 0;
		g.long_first = g.first < lo; // the first line starts further in front than the block keeps: its bytes are not all here
		if (!g.long_first) lo = g.first;
		g.lo = lo & ~(uint64_t)63;
		g.len = g.tot ? (uint32_t)(te0 - g.lo) : 0u; // <= REG, a multiple of 64; a tile in which no line ends is not loaded at all
		return g;
	};
	uint4 pv[NP];
	auto fetch = [&](const Geom &g) {
#pragma unroll
		for (int q = 0; q < NP; ++q) {
			const uint32_t p = threadIdx.x + 256u * (uint32_t)q;
			const uint64_t off = g.lo + (uint64_t)p * 16u;
			pv[q] = p * 16u < g.len && off < a.n ? *(const uint4*)(a.text + off) : make_uint4(0, 0, 0, 0);
		}
	};
	uint32_t t = blockIdx.x;
	Geom cur = {};
	if (t < a.n_tiles) { cur = geom(t); fetch(cur); }
	for (; t < a.n_tiles; t += gridDim.x) {
		const uint32_t line0 = cur.line0, tot = cur.tot, len = cur.len;
		const uint64_t first = cur.first, lo = cur.lo;
		const bool long_first = cur.long_first;
#pragma unroll
		for (int q = 0; q < NP; ++q) {
			const uint32_t p = threadIdx.x + 256u * (uint32_t)q;
			if (p * 16u >= len) break;
			const uint64_t off = lo + (uint64_t)p * 16u;
			uint32_t m = masks16(pv[q]);
			if (off < first) m &= off + 16 <= first ? 0u : (0xffffu << (uint32_t)(first - off) & 0xffffu) * 0x10001u; // bytes of lines that ended in front of this tile
			*(uint4*)(s_text + p * 16u) = pv[q];
			s_tb[p] = (uint16_t)m; s_nb[p] = (uint16_t)(m >> 16);
		}
		if (t + gridDim.x < a.n_tiles) { cur = geom(t + gridDim.x); fetch(cur); } // in flight until the next turn of the loop
		if (tot == 0) continue; // the whole block (nothing was written, nothing is read)
		for (uint32_t p = len / 16u + threadIdx.x; p < REG / 16u; p += 256u) s_nb[p] = 0; // (the ranking below reads the whole array)
		__syncthreads();
		// ---- ranks of the newlines: thread x looks after bytes [64 CH x, 64 CH (x + 1))
		unsigned long long nlm[CH];
		uint32_t cnt = 0;
#pragma unroll
		for (int r = 0; r < CH; ++r) { nlm[r] = *(const unsigned long long*)(s_nb + 4u * (threadIdx.x * CH + r)); cnt += (uint32_t)__popcll(nlm[r]); }
		uint32_t tot_here;
		const uint32_t rank0 = block_excl_scan_256(cnt, s_w, &tot_here);
		if (threadIdx.x == 0) s_lend[0] = long_first ? 0u : (uint32_t)(first - lo) - 1u; // (a long first line never uses it)
		for (uint32_t base = 0; base < tot; base += 256u) {
			{
				uint32_t r = rank0;
#pragma unroll
				for (int q = 0; q < CH; ++q)
					for (unsigned long long m = nlm[q]; m; m &= m - 1, ++r)
						if (r - base < 256u) s_lend[1u + r - base] = (threadIdx.x * CH + q) * 64u + (uint32_t)__builtin_ctzll(m);
			}
			__syncthreads();
			const uint32_t r = base + threadIdx.x;
			const bool act = r < tot;
			const uint32_t i = line0 + r;
			uint32_t fl = 0, mine = 0, x_ps = 0, x_qlen = 0; // (x_*: where this lane's query name stands in LDS and how long it is, for the lane above)
			uint64_t hq = 0, ht = 0;
			if (act) {
				const uint32_t pe = s_lend[1u + threadIdx.x], ps = s_lend[threadIdx.x] + 1u;
				const bool lf = long_first && r == 0;
				lstart[i] = lf ? first : lo + ps;
				if (i + 1 == a.L) lstart[i + 1] = lo + pe + 1;
				bool odd = lf;
				if (!lf) {
					uint32_t e1 = pe;
					if (e1 - ps > 1 && s_text[e1 - 1] == '\r') --e1;
					// ---- column ends (all eleven first, then what they hold: converting each column as soon as its end is known -- 22 registers fewer on paper -- made hipcc
					// interleave less: 99 registers and 4.12 ms instead of 96 and 3.85; capping the registers for six waves a SIMD spills: 5.9 ms)
					uint32_t fb[11], fe[11], ncol = 0, pos = ps;
					bool more = true, amb = false;
#pragma unroll
					for (int k = 0; k < 11; ++k) {
						const bool name = k == 0 || k == 5;
						const uint32_t d = name ? tab_dist3(tb, pos) : tab_dist2(tb, pos);
						uint32_t end = pos + d;
						const bool lastc = end >= e1; // no TAB before the line ends: the last column
						amb |= more && !lastc && d >= (name ? 64u : 33u); // the window ended before the column did
						end = lastc ? e1 : end;
						fb[k] = pos; fe[k] = end;
						ncol += more;
						more = more && !lastc;
						pos = more ? end + 1u : e1;
					}
					const uint32_t valid = ncol >= 10, hasbl = ncol >= 11;
					odd = amb;
					uint32_t ql = 0, qs = 0, qe = 0, tl = 0, ts = 0, te = 0, ml = 0, bl = 0, rev = 0, tnoff = 0, qlen = 0, tlen = 0, pass = 0;
					if (valid && !amb) {
						uint32_t bad = 0, lbad = 0, nul = 0;
#define PAF_NUM(k) ({ const uint32_t l_ = fe[k] - fb[k]; lbad |= (l_ - 1u) > 7u; num8(lds_get8(tw, fb[k]), (l_ - 1u) > 7u ? 8u : l_, &bad); })
						ql = PAF_NUM(1); qs = PAF_NUM(2); qe = PAF_NUM(3);
						tl = PAF_NUM(6); ts = PAF_NUM(7); te = PAF_NUM(8);
						ml = PAF_NUM(9) & 0x7fffffffu;
						if (hasbl) bl = PAF_NUM(10);
#undef PAF_NUM
						rev = fe[4] > fb[4] && s_text[fb[4]] == '-';
						tnoff = fb[5] - ps;
						qlen = fe[0] - fb[0]; tlen = fe[5] - fb[5];
						if (qlen > 64 || tlen > 64) lbad = 1;
						else { hq = lds_name_key(tw, fb[0], qlen, &nul); ht = lds_name_key(tw, fb[5], tlen, &nul); }
						odd = (bad & 0x80808080u) != 0 || lbad || nul;
						pass = !(qe - qs < (uint32_t)a.min_span || te - ts < (uint32_t)a.min_span || (int)ml < a.min_match); // hit.c:85
					}
					if (!odd) {
						fl = valid | pass << 1 | hasbl << 2 | rev << 3;
						c_valid += valid; c_pass += pass; c_nobl += valid && !hasbl;
						if (pass) {
							const uint32_t lng = !key_is_short(qlen) || !key_is_short(tlen);
							c_long += lng;
							mine = 1; x_ps = ps; x_qlen = qlen;
							const uint64_t mq = qs > ts ? qs : ts;
							c_mq = mq > c_mq ? mq : c_mq;
						}
						o.ql[i] = ql; o.qs[i] = qs; o.qe[i] = qe; o.tl[i] = tl; o.ts[i] = ts; o.te[i] = te; o.ml[i] = ml; o.bl[i] = bl;
						o.tnoff[i] = tnoff; o.qlen[i] = qlen; o.tlen[i] = tlen; o.hq[i] = hq; o.ht[i] = ht;
					}
				}
				if (odd) { fl = PF_ODD; ++c_odd; }
			}
			// does the line continue the run of one query name?  (the lane below holds the line in front; lane 0 of a wave starts a run)
			const uint32_t p_mine = wv_prev_lane_u32(mine, 0u, lane), p_lo = wv_prev_lane_u32((uint32_t)hq, 0u, lane), p_hi = wv_prev_lane_u32((uint32_t)(hq >> 32), 0u, lane);
			const uint32_t p_len = wv_prev_lane_u32(x_qlen, 0u, lane), p_ps = wv_prev_lane_u32(x_ps, 0u, lane);
			bool cont = mine && p_mine && p_len == x_qlen && p_lo == (uint32_t)hq && p_hi == (uint32_t)(hq >> 32);
			if (cont && !key_is_short(x_qlen)) // a hashed key: the bytes decide (both names are in LDS: the one place where comparing them costs nothing)
				for (uint32_t k = 0; k < x_qlen && cont; k += 8) {
					const uint32_t rem = x_qlen - k;
					const uint64_t m = rem >= 8 ? ~0ull : (1ull << (8u * rem)) - 1ull;
					cont = ((lds_get8(tw, x_ps + k) ^ lds_get8(tw, p_ps + k)) & m) == 0;
				}
			if (cont) fl |= PF_QCONT;
			if (act) o.flags[i] = (uint8_t)fl;
			__syncthreads();
			if (base + 256u < tot) { if (threadIdx.x == 0) s_lend[0] = s_lend[256]; __syncthreads(); }
		}
	}
	blk_add_u64(&ctr[PC_VALID], c_valid);
	blk_add_u64(&ctr[PC_PASS], c_pass);
	blk_add_u64(&ctr[PC_NOBL], c_nobl);
	blk_add_u64(&ctr[PC_LONG], c_long);
	blk_add_u64(&ctr[PC_ODD], c_odd);
	blk_max_u64(&ctr[PC_MAXQS], c_mq);
}

// the lines the tile parser left (flags == PF_ODD): the byte-wise routine, straight from global memory
__global__ __launch_bounds__(256) void k_paf_parse_odd(const unsigned char *__restrict__ text, const uint64_t *__restrict__ lstart, uint32_t L, int min_span, int min_match, PafCols o,
                                                        unsigned long long *__restrict__ ctr)
{
	__shared__ uint32_t s_fs[256 * 12];
	uint32_t c_valid = 0, c_pass = 0, c_nobl = 0, c_long = 0;
	uint64_t c_mq = 0;
	for (uint32_t base = blockIdx.x * 256u; base < L; base += gridDim.x * 256u) {
		const uint32_t i = base + threadIdx.x;
		if (i >= L || o.flags[i] != PF_ODD) continue;
		const uint64_t ls = lstart[i];
		const uint32_t l = (uint32_t)(lstart[i + 1] - 1 - ls);
		PafLine r;
		const FsCols fs = { s_fs + threadIdx.x };
		paf_line(text + ls, l, fs, r);
		const uint32_t pass = r.valid && !(r.qe - r.qs < (uint32_t)min_span || r.te - r.ts < (uint32_t)min_span || (int)r.ml < min_match); // hit.c:85
		c_valid += r.valid; c_pass += pass; c_nobl += r.valid && !r.hasbl;
		o.flags[i] = (uint8_t)(r.valid | pass << 1 | r.hasbl << 2 | r.rev << 3);
		o.ql[i] = r.ql; o.qs[i] = r.qs; o.qe[i] = r.qe; o.tl[i] = r.tl; o.ts[i] = r.ts; o.te[i] = r.te; o.ml[i] = r.ml; o.bl[i] = r.bl;
		o.tnoff[i] = r.tnoff; o.qlen[i] = r.qlen; o.tlen[i] = r.tlen; o.hq[i] = r.hq; o.ht[i] = r.ht;
		if (pass) {
			c_long += !key_is_short(r.qlen) || !key_is_short(r.tlen);
			const uint64_t mq = r.qs > r.ts ? r.qs : r.ts;
			c_mq = mq > c_mq ? mq : c_mq;
		}
	}
	blk_add_u64(&ctr[PC_VALID], c_valid);
	blk_add_u64(&ctr[PC_PASS], c_pass);
	blk_add_u64(&ctr[PC_NOBL], c_nobl);
	blk_add_u64(&ctr[PC_LONG], c_long);
	blk_max_u64(&ctr[PC_MAXQS], c_mq);
}
__global__ __launch_bounds__(256) void k_paf_hasbl(const uint8_t *__restrict__ flags, uint32_t L, uint32_t *__restrict__ f_hasbl)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < L) f_hasbl[i] = flags[i] >> 2 & 1u;
}

// ------------------------------------------------------------------------------------------------ name dictionary

__device__ __forceinline__ bool name_eq(const unsigned char *__restrict__ a, const unsigned char *__restrict__ b, uint32_t len)
{ // 8 bytes at a time (global loads need no alignment), the tail byte by byte
	uint32_t k = 0;
	for (; k + 8 <= len; k += 8) {
		unsigned long long x, y;
		__builtin_memcpy(&x, a + k, 8); __builtin_memcpy(&y, b + k, 8);
		if (x != y) return false;
	}
	for (; k < len; ++k) if (a[k] != b[k]) return false;
	return true;
}

// (Round 3 tried 32-byte slots that carry the first 16 bytes of the name, so that the sector that answers a probe also settles the comparison: with
// agent-scope accesses to keep the eight L2s honest and a table four times the bytes it was SLOWER -- 17.0 against 11.5 ms per 100 M lines
// (profiles/r03_experiments.txt) -- and was taken out again.)
// One thread per stored line: query name, then target name.  A slot is two words: w0 = tag(32) | an occurrence (2*line + column) of the slot's name -- the
// claimant's at first, the SMALLEST seen so far in the end = the name's first appearance in the file --, info = text offset << 24 | length of the slot's name,
// written by the claimant right after its CAS: a prober that finds it compares the bytes after ONE dependent fetch; one that does not see it yet (the store is
// not ordered with the CAS, a stale line) goes the long way through the occurrence (line start, column offset, length: three more random fetches) -- both ways
// read the same bytes.
#define PAF_INFO_LEN_BITS 24
// the slot of one name occurrence (probe, insert if new); 0xffffffff: the probe sequence ran out
// (Round 6: the slot's two words sit side by side -- one 16-byte fetch answers "whose slot" and "where is its text" --, and the low half of the first word is the SMALLEST
// occurrence seen so far, kept by a 64-bit atomicMin (every writer of a slot carries the slot's tag in the high half): the separate info and tmin arrays, two of a
// probe's five random fetches, are gone.  tmin[] is written once, behind the pass (k_dict_exact_tmin), for the code that ranks the names.)
struct __attribute__((aligned(16))) XSlot { unsigned long long w0, info; };
__device__ __forceinline__ uint32_t dict_probe(const unsigned char *__restrict__ text, const uint64_t *__restrict__ lstart, const PafCols &o,
                                               XSlot *__restrict__ tab, uint32_t mask,
                                               uint64_t h, uint32_t len, uint32_t occ, uint64_t noff, uint32_t *fresh)
{
	const unsigned char *nm = text + noff;
	const uint64_t hm = key_mix(h); // (a short name's key is its bytes: mixed before it picks a slot)
	const uint32_t tag = (uint32_t)(hm >> 32);
	uint32_t s = (uint32_t)hm & mask;
	for (uint32_t probe = 0; probe < PAF_PROBE_LIMIT; ++probe, s = (s + 1) & mask) {
		// (two 64-bit words, not four 32-bit ones: a word another thread is claiming must be seen whole -- as the halves of a uint4 the CPU test build read it in two pieces, and a
		// torn tag sent the probe on to a second slot for the same name: one run of the command line in a thousand)
		const ulonglong2 raw = *(const ulonglong2*)&tab[s];
		unsigned long long e = raw.x, inf = raw.y;
		if (e == PAF_EMPTY) {
			e = atomicCAS(&tab[s].w0, PAF_EMPTY, (unsigned long long)tag << 32 | occ);
			if (e == PAF_EMPTY) {
				if (len < (1u << PAF_INFO_LEN_BITS) && noff < (1ull << (64 - PAF_INFO_LEN_BITS))) tab[s].info = noff << PAF_INFO_LEN_BITS | len;
				++*fresh;
				return s;
			}
			inf = PAF_EMPTY; // (somebody else's slot by now: its second word was not in the fetch)
		}
		if ((uint32_t)(e >> 32) == tag) {
			bool same;
			if (inf != PAF_EMPTY) same = (uint32_t)(inf & ((1u << PAF_INFO_LEN_BITS) - 1)) == len && name_eq(nm, text + (inf >> PAF_INFO_LEN_BITS), len);
			else { // the long way, through an occurrence of the slot's name (any will do: the low half only ever moves to another occurrence of the same name)
				const uint32_t r = (uint32_t)e, rl = r >> 1;
				const uint32_t rlen = (r & 1) ? o.tlen[rl] : o.qlen[rl];
				same = rlen == len && name_eq(nm, text + lstart[rl] + ((r & 1) ? o.tnoff[rl] : 0u), len);
			}
			if (same) {
				if ((uint32_t)e > occ) atomicMin(&tab[s].w0, (unsigned long long)tag << 32 | occ);
				return s;
			}
		}
	}
	return 0xffffffffu;
}
// what the rest of the dictionary code reads: tmin[slot] = first appearance of the slot's name (~0: free slot)
__global__ __launch_bounds__(256) void k_dict_exact_tmin(const XSlot *__restrict__ tab, uint32_t cap, uint32_t *__restrict__ tmin)
{
	const uint32_t s = blockIdx.x * 256u + threadIdx.x;
	if (s < cap) { const unsigned long long e = tab[s].w0; tmin[s] = e == PAF_EMPTY ? 0xffffffffu : (uint32_t)e; }
}

// A PAF file lists a query's overlaps together (the reference's own all-vs-all pipeline writes them so; so does every overlapper that works query by
// query): the QUERY name of a line is, 49 times in 50 at BASELINE coverage, the previous line's.  A wave holds 64 consecutive lines; a lane whose query
// name equals its left neighbour's (same hash, same length, same bytes) does not probe but takes the slot of the nearest lane to its left that did (the
// head of its run: its occurrence number is the run's smallest, so tmin is right as well).  Round 2 probed once per name occurrence: 200 M probes and
// 68 GB of fetches for 100 M lines; the query column now costs one probe per run and wave.  Target names change from line to line and probe as before.
__global__ __launch_bounds__(256) void k_dict_insert(const unsigned char *__restrict__ text, const uint64_t *__restrict__ lstart, uint32_t L, PafCols o,
                                                      XSlot *__restrict__ tab, uint32_t mask, unsigned long long *__restrict__ ctr, int runs_flagged)
{ // runs_flagged: the tile parser compared every line's query name with the line in front (PF_QCONT, bytes in LDS); else (round 5's parser) it is done here, on the text
	uint32_t fail = 0, fresh = 0;
	const unsigned lane = threadIdx.x & 63;
	for (uint32_t base = blockIdx.x * 256u; base < L; base += gridDim.x * 256u) { // wave-uniform: the lanes talk to each other below
		const uint32_t i = base + threadIdx.x;
		const uint32_t fl = i < L ? o.flags[i] : 0u;
		const bool stored = (fl & 2u) != 0;
		const uint64_t ls = stored ? lstart[i] : 0;
		const uint64_t hq = stored ? o.hq[i] : 0;
		const uint32_t qlen = stored ? o.qlen[i] : 0;
		// does this line continue its left neighbour's run of one query name?
		const uint64_t hq_l = (uint64_t)__shfl_up((uint32_t)(hq >> 32), 1, 64) << 32 | __shfl_up((uint32_t)hq, 1, 64);
		const uint32_t qlen_l = __shfl_up(qlen, 1, 64);
		const uint64_t ls_l = (uint64_t)__shfl_up((uint32_t)(ls >> 32), 1, 64) << 32 | __shfl_up((uint32_t)ls, 1, 64);
		const int stored_l = __shfl_up((int)stored, 1, 64);
		const bool cont = stored && lane > 0 && stored_l && (runs_flagged ? (fl & PF_QCONT) != 0 : hq_l == hq && qlen_l == qlen && name_eq(text + ls, text + ls_l, qlen));
		uint32_t qslot = 0xffffffffu;
		if (stored && !cont) qslot = dict_probe(text, lstart, o, tab, mask, hq, qlen, i * 2u, ls, &fresh);
		const unsigned long long heads = __ballot(stored && !cont);
		{ // a continuing lane: the slot of the nearest head to its left (there is one: lane 0 never continues)
			const unsigned long long left = heads & ((1ull << lane) - 1ull);
			const int src = left ? 63 - __builtin_clzll(left) : (int)lane;
			const uint32_t got = __shfl(qslot, src, 64);
			if (cont) qslot = got;
		}
		if (stored) {
			const uint32_t tslot = dict_probe(text, lstart, o, tab, mask, o.ht[i], o.tlen[i], i * 2u + 1u, ls + o.tnoff[i], &fresh);
			if (qslot == 0xffffffffu || tslot == 0xffffffffu) fail = 1;
			o.qslot[i] = qslot; o.tslot[i] = tslot;
		}
	}
	blk_add_u64(&ctr[PC_OVERFLOW], fail);
	blk_add_u64(&ctr[PC_DISTINCT], fresh);
}

// ---- the dictionary of a file whose names are all 1..8 bytes (PC_LONG == 0): the key IS the name, so a slot that holds the key holds the name -- no text is
// compared and none is fetched.  Slot = 16 bytes { key, ~(smallest occurrence), - }: all zero = free, claimed by a 64-bit CAS on the key; the occurrence word is
// kept by atomicMax of the complement (so that zeroed memory is "none yet") and a probe touches it only when its own occurrence is smaller than what the same
// 16-byte fetch showed.  One random fetch per name occurrence (round 5's probe: five -- slot word, info word, the first inserter's text, tmin, own text).
// Run heads come from the parser (PF_QCONT), which compared the keys while it had them.
struct __attribute__((aligned(16))) DSlot { unsigned long long key; uint32_t inv_occ, pad; };
__device__ __forceinline__ uint32_t dict_probe_short(DSlot *__restrict__ tab, uint32_t mask, uint64_t key, uint32_t occ, uint32_t *fresh)
{
	uint32_t s = (uint32_t)key_mix(key) & mask;
	for (uint32_t probe = 0; probe < PAF_PROBE_LIMIT; ++probe, s = (s + 1) & mask) {
		const ulonglong2 e = *(const ulonglong2*)&tab[s]; // (64-bit words: a key being claimed is seen whole or not at all, see dict_probe)
		unsigned long long k = e.x;
		uint32_t seen = (uint32_t)e.y;
		if (k == 0) {
			k = atomicCAS(&tab[s].key, 0ull, (unsigned long long)key);
			if (k == 0) { ++*fresh; k = key; }
			seen = 0;
		}
		if (k == key) {
			if (seen < ~occ) atomicMax(&tab[s].inv_occ, ~occ);
			return s;
		}
	}
	return 0xffffffffu;
}
__global__ __launch_bounds__(256) void k_dict_insert_short(PafCols o, uint32_t L, DSlot *__restrict__ tab, uint32_t mask, unsigned long long *__restrict__ ctr)
{
	uint32_t fail = 0, fresh = 0;
	const unsigned lane = threadIdx.x & 63;
	for (uint32_t base = blockIdx.x * 256u; base < L; base += gridDim.x * 256u) { // wave-uniform
		const uint32_t i = base + threadIdx.x;
		const uint32_t fl = i < L ? o.flags[i] : 0u;
		const bool stored = (fl & 2u) != 0;
		const bool head = stored && (!(fl & PF_QCONT) || lane == 0);
		uint32_t qslot = 0xffffffffu;
		if (head) qslot = dict_probe_short(tab, mask, o.hq[i], i * 2u, &fresh);
		const unsigned long long heads = __ballot(head);
		{ // a continuing lane: the slot of the nearest head to its left (lane 0 is one whenever it is stored)
			const unsigned long long left = heads & ((1ull << lane) - 1ull);
			const int src = left ? 63 - __builtin_clzll(left) : (int)lane;
			const uint32_t got = __shfl(qslot, src, 64);
			if (stored && !head) qslot = got;
		}
		if (stored) {
			const uint32_t tslot = dict_probe_short(tab, mask, o.ht[i], i * 2u + 1u, &fresh);
			if (qslot == 0xffffffffu || tslot == 0xffffffffu) fail = 1;
			o.qslot[i] = qslot; o.tslot[i] = tslot;
		}
	}
	blk_add_u64(&ctr[PC_OVERFLOW], fail);
	blk_add_u64(&ctr[PC_DISTINCT], fresh);
}
// what the rest of the dictionary code reads: tmin[slot] = first appearance of the slot's name (~0: free slot)
__global__ __launch_bounds__(256) void k_dict_short_tmin(const DSlot *__restrict__ tab, uint32_t cap, uint32_t *__restrict__ tmin)
{
	const uint32_t s = blockIdx.x * 256u + threadIdx.x;
	if (s < cap) tmin[s] = tab[s].key ? ~tab[s].inv_occ : 0xffffffffu;
}

// ---- -R (ma_hit_no_cont, hit.c:38-68) on the parsed columns: reads that are clearly contained are excluded BEFORE ids are given out
// (hit.c:86), so the exclusion is a property of NAMES: a line's verdict flags the name's table slot, lines that touch a flagged name are
// dropped, and first appearances are taken over the lines that are left.
__global__ __launch_bounds__(256) void k_paf_nocont(PafCols o, uint32_t L, int max_hang, float int_frac, uint8_t *__restrict__ excl)
{
	for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < L; i += gridDim.x * 256u) {
		if (!(o.flags[i] & 2)) continue;
		const int v = mc_no_cont(o.ql[i], o.qs[i], o.qe[i], o.tl[i], o.ts[i], o.te[i], o.flags[i] >> 3 & 1, max_hang, int_frac);
		if (v == 1) excl[o.tslot[i]] = 1;
		else if (v == 2) excl[o.qslot[i]] = 1;
	}
}
__global__ __launch_bounds__(256) void k_paf_refilter(PafCols o, uint32_t L, const uint8_t *__restrict__ excl, uint32_t *__restrict__ tmin, unsigned long long *__restrict__ ctr)
{
	uint32_t pass = 0;
	for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < L; i += gridDim.x * 256u) {
		if (!(o.flags[i] & 2)) continue;
		const uint32_t qs = o.qslot[i], ts = o.tslot[i];
		if (excl[qs] || excl[ts]) { o.flags[i] &= (uint8_t)~2u; continue; }
		atomicMin(&tmin[qs], i * 2u); atomicMin(&tmin[ts], i * 2u + 1u);
		++pass;
	}
	blk_add_u64(&ctr[PC_PASS], pass);
}
__global__ __launch_bounds__(256) void k_excl_count(const uint8_t *__restrict__ excl, uint32_t cap, unsigned long long *__restrict__ ctr)
{
	uint32_t n = 0;
	for (uint32_t s = blockIdx.x * 256u + threadIdx.x; s < cap; s += gridDim.x * 256u) n += excl[s];
	blk_add_u64(&ctr[PC_VALID], n);
}

// a name gets an id if some STORED line carries it (with -R a name may sit in the table without such a line); tmin is ~0 in free slots as well
__global__ __launch_bounds__(256) void k_dict_flag(const uint32_t *__restrict__ tmin, uint32_t cap, uint32_t *__restrict__ keep)
{
	uint32_t s = blockIdx.x * 256u + threadIdx.x;
	if (s < cap) keep[s] = tmin[s] != 0xffffffffu;
}
__global__ __launch_bounds__(256) void k_dict_collect(const uint32_t *__restrict__ keep, const uint32_t *__restrict__ pos, const uint32_t *__restrict__ tmin, uint32_t cap,
                                                       uint64_t *__restrict__ key, uint32_t *__restrict__ val)
{
	uint32_t s = blockIdx.x * 256u + threadIdx.x;
	if (s < cap && keep[s]) { key[pos[s]] = tmin[s]; val[pos[s]] = s; }
}
// names sorted by first appearance: rank = id (sdict.c:27-45); first-seen length, where the name's bytes are
__global__ __launch_bounds__(256) void k_dict_assign(const uint64_t *__restrict__ key, const uint32_t *__restrict__ val, uint32_t R, const uint64_t *__restrict__ lstart, PafCols o,
                                                      uint32_t *__restrict__ slot_id, uint32_t *__restrict__ seq_len, uint64_t *__restrict__ name_off, uint32_t *__restrict__ name_len,
                                                      uint32_t *__restrict__ keep)
{
	uint32_t j = blockIdx.x * 256u + threadIdx.x;
	if (j >= R) return;
	const uint32_t occ = (uint32_t)key[j], line = occ >> 1, col = occ & 1;
	slot_id[val[j]] = j;
	seq_len[j] = col ? o.tl[line] : o.ql[line];
	name_off[j] = lstart[line] + (col ? o.tnoff[line] : 0u);
	const uint32_t len = col ? o.tlen[line] : o.qlen[line];
	name_len[j] = len;
	keep[j] = len + 1;
}
__global__ __launch_bounds__(256) void k_dict_names(const unsigned char *__restrict__ text, const uint64_t *__restrict__ name_off, const uint32_t *__restrict__ name_len,
                                                     const uint32_t *__restrict__ name_pos, uint32_t R, char *__restrict__ out)
{
	uint32_t j = blockIdx.x * 256u + threadIdx.x;
	if (j >= R) return;
	const unsigned char *s = text + name_off[j];
	char *d = out + name_pos[j];
	const uint32_t len = name_len[j];
	for (uint32_t k = 0; k < len; ++k) d[k] = (char)s[k];
	d[len] = 0;
}

// ------------------------------------------------------------------------------------------------ the ranks' name tables -> one dictionary
// Sharded ingest (host/ingest_sharded.c, SURVEY 8e "ingest routing, option B"): every rank parses its own byte range of the text and holds the distinct names
// of that range with the GLOBAL occurrence number (2 x line in the whole file + column) of their first appearance there.  The tables of all ranks are gathered
// (rows + name bytes), every rank inserts all of them into one table -- minimum occurrence per name, first-seen length riding in the low word -- and sorts the
// distinct names by that minimum: the same ids on every rank, the ids the reference's sequential reader hands out (sdict.c:27-45).
struct NameRow { uint32_t occ, seq_len, name_len, name_pos; }; // one distinct name of one rank's range; name_pos: offset in that rank's block of name bytes

__device__ __forceinline__ uint64_t fnv_bytes(const unsigned char *__restrict__ p, uint32_t len)
{
	uint64_t h = FNV_OFF;
	for (uint32_t k = 0; k < len; ++k) h = (h ^ p[k]) * FNV_PRIME;
	return h;
}
// row t = rank (t / stride_rows), index (t % stride_rows); its bytes: blobs + rank * stride_bytes + name_pos
__global__ __launch_bounds__(256) void k_dict_merge(const NameRow *__restrict__ rows, const unsigned char *__restrict__ blobs, const uint32_t *__restrict__ n_rows /* per rank */, int world,
                                                     uint32_t stride_rows, size_t stride_bytes, unsigned long long *__restrict__ tab, unsigned long long *__restrict__ gkey, uint32_t mask,
                                                     uint32_t *__restrict__ slot_of, unsigned long long *__restrict__ ctr)
{
	uint32_t fail = 0;
	for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < (size_t)world * stride_rows; t += (size_t)gridDim.x * 256) {
		const uint32_t rk = (uint32_t)(t / stride_rows), j = (uint32_t)(t % stride_rows);
		if (j >= n_rows[rk]) continue;
		const NameRow r = rows[t];
		const unsigned char *nm = blobs + (size_t)rk * stride_bytes + r.name_pos;
		const uint64_t h = fnv_bytes(nm, r.name_len);
		const uint32_t tag = (uint32_t)(h >> 32);
		uint32_t s = (uint32_t)h & mask, got = 0xffffffffu;
		for (uint32_t probe = 0; probe < PAF_PROBE_LIMIT; ++probe, s = (s + 1) & mask) {
			unsigned long long e = tab[s];
			if (e == PAF_EMPTY) {
				e = atomicCAS(&tab[s], PAF_EMPTY, (unsigned long long)tag << 32 | (uint32_t)t);
				if (e == PAF_EMPTY) { got = s; break; }
			}
			if ((uint32_t)(e >> 32) == tag) { // the claimant's row is part of the gathered (read-only) input: comparable at once
				const uint32_t t2 = (uint32_t)e, rk2 = t2 / stride_rows;
				const NameRow r2 = rows[t2];
				if (r2.name_len == r.name_len && name_eq(nm, blobs + (size_t)rk2 * stride_bytes + r2.name_pos, r.name_len)) { got = s; break; }
			}
		}
		if (got == 0xffffffffu) { fail = 1; continue; }
		atomicMin(&gkey[got], (unsigned long long)r.occ << 32 | r.seq_len); // first appearance in the whole file, and the length seen there
		slot_of[t] = got;
	}
	blk_add_u64(&ctr[PC_OVERFLOW], fail);
}
__global__ __launch_bounds__(256) void k_merge_flag(const unsigned long long *__restrict__ tab, uint32_t cap, uint32_t *__restrict__ keep)
{
	uint32_t s = blockIdx.x * 256u + threadIdx.x;
	if (s < cap) keep[s] = tab[s] != PAF_EMPTY;
}
__global__ __launch_bounds__(256) void k_merge_collect(const uint32_t *__restrict__ keep, const uint32_t *__restrict__ pos, const unsigned long long *__restrict__ gkey, uint32_t cap,
                                                        uint64_t *__restrict__ key, uint32_t *__restrict__ val)
{
	uint32_t s = blockIdx.x * 256u + threadIdx.x;
	if (s < cap && keep[s]) { key[pos[s]] = gkey[s] >> 32; val[pos[s]] = s; }
}
// id i = the i-th name by first appearance: its table slot, length, where its bytes are, how long they are
__global__ __launch_bounds__(256) void k_merge_assign(const uint32_t *__restrict__ val, uint32_t R, const unsigned long long *__restrict__ tab, const unsigned long long *__restrict__ gkey,
                                                       const NameRow *__restrict__ rows, uint32_t stride_rows, size_t stride_bytes, uint32_t *__restrict__ gid_of_slot,
                                                       uint32_t *__restrict__ seq_len, uint64_t *__restrict__ name_off, uint32_t *__restrict__ name_len, uint32_t *__restrict__ keep)
{
	uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= R) return;
	const uint32_t s = val[i], t = (uint32_t)tab[s];
	const NameRow r = rows[t];
	gid_of_slot[s] = i;
	seq_len[i] = (uint32_t)gkey[s];
	name_off[i] = (uint64_t)(t / stride_rows) * stride_bytes + r.name_pos;
	name_len[i] = r.name_len;
	keep[i] = r.name_len + 1;
}
// this rank's table slot -> global id, through the slot's local id and the row it became
__global__ __launch_bounds__(256) void k_merge_map(const uint32_t *__restrict__ keep_local /* slot in use */, const uint32_t *__restrict__ slot_id /* slot -> local id */, uint32_t cap,
                                                    const uint32_t *__restrict__ slot_of, uint32_t row0, const uint32_t *__restrict__ gid_of_slot, uint32_t *__restrict__ out)
{
	uint32_t s = blockIdx.x * 256u + threadIdx.x;
	if (s < cap && keep_local[s]) out[s] = gid_of_slot[slot_of[row0 + slot_id[s]]];
}
__global__ __launch_bounds__(256) void k_name_rows(const uint64_t *__restrict__ key /* sorted local first appearances */, uint32_t R, uint32_t occ_base, const uint32_t *__restrict__ seq_len,
                                                    const uint32_t *__restrict__ name_len, const uint32_t *__restrict__ name_pos, NameRow *__restrict__ out)
{
	uint32_t j = blockIdx.x * 256u + threadIdx.x;
	if (j < R) { NameRow r; r.occ = (uint32_t)key[j] + occ_base; r.seq_len = seq_len[j]; r.name_len = name_len[j]; r.name_pos = name_pos[j]; out[j] = r; }
}
__global__ __launch_bounds__(256) void k_bl_fill_from(const uint32_t *__restrict__ f_hasbl, const uint32_t *__restrict__ pos, const uint32_t *__restrict__ blv, uint32_t L, uint32_t *__restrict__ bl, uint32_t before)
{ // k_paf_bl_fill for a text range that does not start the file: lines in front of the range's first 11-column line inherit `before`, the previous ranges' last bl
	uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < L && !f_hasbl[i]) bl[i] = pos[i] ? blv[pos[i] - 1] : before;
}

// ------------------------------------------------------------------------------------------------ records

__global__ __launch_bounds__(256) void k_paf_ids(PafCols o, const uint32_t *__restrict__ slot_id, uint32_t L, int bi_dir, uint32_t *__restrict__ keep)
{
	uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= L) return;
	uint32_t cnt = 0;
	if (o.flags[i] & 2) {
		const uint32_t qid = slot_id[o.qslot[i]], tid = slot_id[o.tslot[i]];
		o.qslot[i] = qid; o.tslot[i] = tid;
		cnt = 1 + (bi_dir && qid != tid); // hit.c:87-98
	}
	keep[i] = cnt;
}

__global__ __launch_bounds__(256) void k_paf_emit(PafCols o, const uint32_t *__restrict__ keep, const uint32_t *__restrict__ pos, uint32_t L, uint4 *__restrict__ rec)
{
	uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= L || keep[i] == 0) return;
	const uint32_t qid = o.qslot[i], tid = o.tslot[i];
	const uint32_t mlrev = o.ml[i] | (uint32_t)(o.flags[i] >> 3 & 1) << 31, bl = o.bl[i] & 0x7fffffffu;
	const size_t k = (size_t)pos[i] * 2;
	rec[k] = make_uint4(o.qs[i], qid, o.qe[i], tid);       // qns = qid<<32 | qs ; qe ; tn
	rec[k + 1] = make_uint4(o.ts[i], o.te[i], mlrev, bl);   // ts ; te ; ml|rev ; bl|del=0
	if (keep[i] == 2) {
		rec[k + 2] = make_uint4(o.ts[i], tid, o.te[i], qid);
		rec[k + 3] = make_uint4(o.qs[i], o.qe[i], mlrev, bl);
	}
}

// ids + record slots + records in ONE pass (until round 5: k_paf_ids, a scan, k_paf_emit -- the columns read twice, ids and counts written and read back):
// a tile of 1 024 lines looks up its ids, counts its records, learns where they go from the tiles in front of it (chained look-back, scan.hip) and writes them.
// A lane has a LINE (row r of the tile = lines 64 r .. 64 r + 63, wave w takes rows w, w + 4, ...): a line's 32 or 64 bytes of records leave from one lane, so the
// lanes of a store instruction fill neighbouring 64-byte stretches (four lines per lane, as the first version had it, put them 256 bytes apart: 4.7 ms against
// the 4.1 of the three launches it replaced).
#ifndef EM_ROWS
#define EM_ROWS 16u // rows per tile: a tile draws ONE ticket from one word (12 - 17 ns each, serial): 1 row 5.57 ms, 2: 4.59, 4: 4.09, 8: 3.96 per 100 M lines
#endif
#define EM_TILE (256u * EM_ROWS)
__global__ __launch_bounds__(256) void k_paf_emit_chain(PafCols o, const uint32_t *__restrict__ slot_id, uint32_t L, int bi_dir, uint4 *__restrict__ rec, uint32_t *__restrict__ d_total,
                                                         unsigned long long *state, uint32_t *ticket, uint32_t ticket_base, uint32_t epoch)
{
	__shared__ uint32_t s_row[4 * EM_ROWS + 1];
	__shared__ uint32_t s_tile, s_prefix;
	if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u) - ticket_base;
	__syncthreads();
	const uint32_t tile = s_tile;
	const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const size_t tbase = (size_t)tile * EM_TILE;
	uint32_t fl[EM_ROWS], qid[EM_ROWS], tid[EM_ROWS], cnt[EM_ROWS], ex[EM_ROWS];
#pragma unroll
	for (unsigned k = 0; k < EM_ROWS; ++k) {
		const size_t i = tbase + (size_t)(k * 4u + wave) * 64u + lane;
		const bool in = i < L; const size_t j = in ? i : 0;
		fl[k] = in ? o.flags[j] : 0u; qid[k] = o.qslot[j]; tid[k] = o.tslot[j];
	}
#pragma unroll
	for (unsigned k = 0; k < EM_ROWS; ++k) {
		cnt[k] = 0;
		if (fl[k] & 2u) {
#ifndef EXP_EMIT_NO_LOOKUP // (experiment: what the pass costs without its random fetches -- the ids are wrong then)
			qid[k] = slot_id[qid[k]]; tid[k] = slot_id[tid[k]];
#endif
			cnt[k] = 1u + (bi_dir && qid[k] != tid[k]); // hit.c:87-98
		}
		const uint32_t incl = (uint32_t)wv_scan_incl_i32((int)cnt[k], lane);
		ex[k] = incl - cnt[k];
		if (lane == 63) s_row[k * 4u + wave] = incl;
	}
	__syncthreads();
	if (threadIdx.x == 0) { // the rows' totals -> their offsets in the tile, the tile's total
		uint32_t run = 0;
		for (unsigned r = 0; r < 4 * EM_ROWS; ++r) { const uint32_t v = s_row[r]; s_row[r] = run; run += v; }
		s_row[4 * EM_ROWS] = run;
		SC_PUBLISH(&state[tile], sc_pack(epoch, tile == 0 ? SC_INCL : SC_AGG, run));
		if (tile == 0) s_prefix = 0;
	}
	__syncthreads();
	if (tile > 0 && threadIdx.x < 64) {
		const uint32_t tot = s_row[4 * EM_ROWS];
		const uint32_t prefix = sc_look_back(state, tile, epoch, threadIdx.x);
		if (threadIdx.x == 0) { s_prefix = prefix; SC_PUBLISH(&state[tile], sc_pack(epoch, SC_INCL, prefix + tot)); }
	}
	__syncthreads();
	const uint32_t p0 = s_prefix;
#pragma unroll
	for (unsigned k = 0; k < EM_ROWS; ++k) { // (the six number columns are fetched here, a row ahead of its stores: asking for all of them up front bought nothing and costs 6 registers a row)
		if (!cnt[k]) continue;
		const size_t i = tbase + (size_t)(k * 4u + wave) * 64u + lane;
		const uint32_t qs = o.qs[i], qe = o.qe[i], ts = o.ts[i], te = o.te[i];
		const uint32_t mlrev = o.ml[i] | (fl[k] >> 3 & 1u) << 31, b31 = o.bl[i] & 0x7fffffffu;
		uint4 *r = rec + (size_t)(p0 + s_row[k * 4u + wave] + ex[k]) * 2;
		r[0] = make_uint4(qs, qid[k], qe, tid[k]);   // qns = qid<<32 | qs ; qe ; tn
		r[1] = make_uint4(ts, te, mlrev, b31);       // ts ; te ; ml|rev ; bl|del=0
		if (cnt[k] == 2) {
			r[2] = make_uint4(ts, tid[k], te, qid[k]);
			r[3] = make_uint4(qs, qe, mlrev, b31);
		}
	}
	if (threadIdx.x == 0 && tbase < L && tbase + EM_TILE >= L) *d_total = p0 + s_row[4 * EM_ROWS]; // the last tile: its end is the total
}

// ------------------------------------------------------------------------------------------------ host entries

static int paf_reserve_text(mahip_ctx *c, size_t nbytes)
{
	PafBufs *b = paf_of(c);
	{ // a cap on the text the device stage accepts (tests use it to exercise the caller's fall-back to the host reader)
		const char *e = getenv("MA_PAF_MAX_BYTES");
		if (e && nbytes > (size_t)atoll(e)) { mahip_set_error("text of %zu bytes exceeds MA_PAF_MAX_BYTES", nbytes); return -1; }
	}
	const bool timing = getenv("MA_PIPE_TIMING") != nullptr;
	struct timespec ts0, ts1;
	if (timing) clock_gettime(CLOCK_MONOTONIC, &ts0);
	CHK(dev_reserve(c, b->text, nbytes + 64));
	if (timing) {
		clock_gettime(CLOCK_MONOTONIC, &ts1);
		fprintf(stderr, "[T::paf] hipMalloc of the text buffer: %.3f s\n", (double)(ts1.tv_sec - ts0.tv_sec) + 1e-9 * (double)(ts1.tv_nsec - ts0.tv_nsec));
	}
	b->nbytes = nbytes;
	b->loaded = false;
	return 0;
}

extern "C" int mahip_paf_load_mem(mahip_ctx_t *c, const void *text, size_t nbytes)
{
	HIPCHK(hipSetDevice(c->dev));
	CHK(paf_reserve_text(c, nbytes));
	CHK(xfer_copy(c, paf_of(c)->text.p, (void*)text, nbytes, 1));
	paf_of(c)->loaded = true;
	return 0;
}

extern "C" int mahip_paf_load_fd(mahip_ctx_t *c, int fd, size_t nbytes)
{
	HIPCHK(hipSetDevice(c->dev));
	CHK(paf_reserve_text(c, nbytes));
	CHK(xfer_from_fd(c, paf_of(c)->text.p, fd, nbytes));
	paf_of(c)->loaded = true;
	return 0;
}

extern "C" int mahip_paf_load_fd_range(mahip_ctx_t *c, int fd, size_t off, size_t nbytes)
{ // bytes [off, off + nbytes) of an open plain file: a rank's own range of the text (cut at line starts by the caller)
	HIPCHK(hipSetDevice(c->dev));
	CHK(paf_reserve_text(c, nbytes));
	if (nbytes) CHK(xfer_from_fd_at(c, paf_of(c)->text.p, fd, off, nbytes));
	paf_of(c)->loaded = true;
	return 0;
}

static uint32_t pow2_at_least(uint64_t x) { uint64_t p = 1; while (p < x) p <<= 1; return p > 0x80000000ull ? 0x80000000u : (uint32_t)p; }
static int bits_of(uint64_t x) { int b = 0; while (x) ++b, x >>= 1; return b; }

extern "C" int mahip_paf_parse(mahip_ctx_t *c, int min_span, int min_match, int bi_dir, mahip_paf_info_t *info)
{
	return mahip_paf_parse_excl(c, min_span, min_match, bi_dir, 0, 0, 0.f, info);
}

// sharded: the text in the context is THIS RANK'S byte range of the file (cut at line starts, ranges in rank order); the context has a communicator.  The ranks
// exchange what the sequential semantics need across range borders -- line counts (occurrence numbers count lines of the whole file), the last `bl` a range
// leaves behind, the distinct names of every range with their first appearances -- and every rank ends with the records of ITS lines carrying the ids the
// reference would give (the same dictionary on every rank).  info: n_lines / n_records / n_stored_lines are totals over the ranks, n_hits is this rank's.
static int paf_parse_impl(mahip_ctx_t *c, int min_span, int min_match, int bi_dir, int no_cont, int max_hang, float int_frac, mahip_paf_info_t *info, bool sharded)
{
	HIPCHK(hipSetDevice(c->dev));
	PafBufs *b = paf_of(c);
	const int W = sharded ? mahip_comm_world(c) : 1, me = sharded ? mahip_comm_rank(c) : 0;
	if (sharded && no_cont) { mahip_set_error("mahip_paf_parse_sharded: the -R pre-filter needs the whole text on one rank"); return -1; }
	if (sharded && W > 32) { mahip_set_error("mahip_paf_parse_sharded: at most 32 ranks"); return -1; }
	uint64_t line_base = 0, lines_total = 0, valid_total = 0, pass_total = 0;
	if (!b->loaded) { mahip_set_error("mahip_paf_parse: no text loaded"); return -1; }
	const size_t n = b->nbytes;
	const unsigned char *text = P<unsigned char>(b->text);
	unsigned long long *ctr = P<unsigned long long>(c->ctr);
	memset(info, 0, sizeof(*info));
	b->n_seq = 0; b->name_bytes = 0;
	// MA_PIPE_TIMING=2: wall-clock laps of this function's parts on stderr (each lap waits for the stream: a diagnostic, it changes what it measures by the waits)
	const bool laps = []{ const char *e = getenv("MA_PIPE_TIMING"); return e && atoi(e) >= 2; }();
	struct timespec lap_t0;
	if (laps) clock_gettime(CLOCK_MONOTONIC, &lap_t0);
	auto lap = [&](const char *what) {
		if (!laps) return;
		(void)hipStreamSynchronize(c->st);
		struct timespec t1;
		clock_gettime(CLOCK_MONOTONIC, &t1);
		fprintf(stderr, "[T::paf_parse] %-28s %9.3f ms\n", what, ((double)(t1.tv_sec - lap_t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - lap_t0.tv_nsec)) * 1e3);
		lap_t0 = t1;
	};

	// ---- line starts
	const bool old_path = []{ const char *e = getenv("MA_PAF_OLD"); return e && atoi(e) != 0; }(); // round 5's kernels (lane-per-line parser, text-comparing dictionary, ids / scan / emit): the A/B switch
	CHK(dev_reserve(c, b->scal, 64));
	CHK(ctr_zero(c));
	uint32_t L = 0;
	int open_line = 0;
	uint32_t n_gran = 0, tile_k = 1, n_tiles = 0;
	size_t n_eff = n; // the tile parser's text: n + the appended newline
	if (n) { // same decision as the kernels': L = newlines + an unterminated tail
		unsigned char last = 0;
		HIPCHK(hipMemcpyAsync(&last, text + n - 1, 1, hipMemcpyDeviceToHost, c->st));
		HIPCHK(hipStreamSynchronize(c->st));
		open_line = last != '\n';
	}
	if (n && old_path) {
		const size_t n_tiles4k = (n + PAF_TILE - 1) / PAF_TILE;
		if (n_tiles4k > 0x7fffffffull) { mahip_set_error("mahip_paf_parse: text too large"); return -1; }
		CHK(dev_reserve(c, b->tile, (n_tiles4k + 8) * 4));
		{
			ProfScope ps(c, "k_paf_nl_count", (double)n);
			hipLaunchKernelGGL(k_paf_nl_count, dim3((unsigned)n_tiles4k), dim3(256), 0, c->st, text, n, P<uint32_t>(b->tile));
		}
		CHK(scan_exclusive_u32(c, P<uint32_t>(b->tile), P<uint32_t>(b->tile), n_tiles4k, P<uint32_t>(b->scal)));
		uint32_t n_nl = 0;
		HIPCHK(hipMemcpyAsync(&n_nl, b->scal.p, 4, hipMemcpyDeviceToHost, c->st));
		HIPCHK(hipStreamSynchronize(c->st));
		if ((uint64_t)n_nl + 1 >= 0x7fffffffull) { mahip_set_error("mahip_paf_parse: more than 2^31 lines"); return -1; }
		CHK(dev_reserve(c, b->lstart, ((size_t)n_nl + 4) * 8));
		{
			ProfScope ps(c, "k_paf_nl_pos", (double)n + 8.0 * (double)n_nl);
			hipLaunchKernelGGL(k_paf_nl_pos, dim3((unsigned)n_tiles4k), dim3(256), 0, c->st, text, n, (const uint32_t*)P<uint32_t>(b->tile), (const uint32_t*)P<uint32_t>(b->scal), P<uint64_t>(b->lstart), ctr);
		}
		L = n_nl + (uint32_t)open_line;
	} else if (n) { // newline census per KiB; the tile parser writes the line starts itself
		// every line ends with a newline: a text that does not gets one (the buffer has 64 spare bytes; the line's sentinel n + 1 is what k_paf_nl_pos gives it),
		// and zeros follow, so that the kernels read whole 16-byte pieces
		HIPCHK(hipMemsetAsync((void*)(text + n), 0, 64, c->st));
		if (open_line) HIPCHK(hipMemsetAsync((void*)(text + n), '\n', 1, c->st));
		n_eff = n + (size_t)open_line;
		const size_t ng = (n_eff + PAF_GRAN - 1) / PAF_GRAN;
		if (ng > 0x7ffffff0ull) { mahip_set_error("mahip_paf_parse: text too large"); return -1; }
		n_gran = (uint32_t)ng;
		CHK(dev_reserve(c, b->tile, ((size_t)n_gran + 8) * 4)); CHK(dev_reserve(c, b->glast, ((size_t)n_gran + 8) * 4));
		{
			ProfScope ps(c, "k_paf_nl_count", (double)n);
			hipLaunchKernelGGL(k_paf_gran_count, dim3((n_gran + 4 * PAF_GC_PER_WAVE - 1) / (4 * PAF_GC_PER_WAVE)), dim3(256), 0, c->st, text, n_eff, n_gran, P<uint32_t>(b->tile), P<uint32_t>(b->glast));
		}
		CHK(scan_exclusive_u32(c, P<uint32_t>(b->tile), P<uint32_t>(b->tile), n_gran, P<uint32_t>(b->tile) + n_gran)); // goff[n_gran] = all newlines
		uint64_t n_nl = 0;
		{
			uint32_t t32 = 0;
			HIPCHK(hipMemcpyAsync(&t32, P<uint32_t>(b->tile) + n_gran, 4, hipMemcpyDeviceToHost, c->st));
			HIPCHK(hipStreamSynchronize(c->st));
			n_nl = t32;
		}
		if (n_nl + 1 >= 0x7fffffffull) { mahip_set_error("mahip_paf_parse: more than 2^31 lines"); return -1; }
		L = (uint32_t)n_nl; // (the virtual newline is counted)
		CHK(dev_reserve(c, b->lstart, ((size_t)L + 4) * 8));
		if (L) {
			// tile = K granules with about 240 lines (a lane per line, 256 lanes); at most what a block stages (15 KiB + 1 KiB in front with 16 KiB of LDS text, 31 + 1 with 32)
			double k = 240.0 * ((double)n / (double)L) / (double)PAF_GRAN;
			if (const char *e = getenv("MA_PAF_TILE_K")) k = atof(e);
			tile_k = k < 1.0 ? 1u : k > 31.0 ? 31u : (uint32_t)k;
			n_tiles = (n_gran + tile_k - 1) / tile_k;
			const uint32_t n_grp = (n_gran >> PAF_BGRP) + 1;
			CHK(dev_reserve(c, b->gmax, ((size_t)n_grp + 1) * 8)); CHK(dev_reserve(c, b->tfirst, ((size_t)n_tiles + 1) * 8));
			hipLaunchKernelGGL(k_paf_gran_bmax, dim3(n_grp), dim3(64), 0, c->st, (const uint32_t*)P<uint32_t>(b->glast), n_gran, P<unsigned long long>(b->gmax));
			hipLaunchKernelGGL(k_paf_tile_first, dim3(grid_for(n_tiles, 256)), dim3(256), 0, c->st, (const uint32_t*)P<uint32_t>(b->glast), (const unsigned long long*)P<unsigned long long>(b->gmax), tile_k, n_tiles,
			                   P<unsigned long long>(b->tfirst));
		}
	}
	lap("line census");
	PafCols o;
	{
		const size_t Lr = (size_t)L + 4;
		CHK(dev_reserve(c, b->flags, Lr));
		for (int k = 0; k < 8; ++k) CHK(dev_reserve(c, b->num[k], Lr * 4));
		CHK(dev_reserve(c, b->tnoff, Lr * 4)); CHK(dev_reserve(c, b->qlen, Lr * 4)); CHK(dev_reserve(c, b->tlen, Lr * 4));
		CHK(dev_reserve(c, b->hq, Lr * 8)); CHK(dev_reserve(c, b->ht, Lr * 8));
		CHK(dev_reserve(c, b->qslot, Lr * 4)); CHK(dev_reserve(c, b->tslot, Lr * 4));
		o.flags = P<uint8_t>(b->flags);
		o.ql = P<uint32_t>(b->num[0]); o.qs = P<uint32_t>(b->num[1]); o.qe = P<uint32_t>(b->num[2]); o.tl = P<uint32_t>(b->num[3]);
		o.ts = P<uint32_t>(b->num[4]); o.te = P<uint32_t>(b->num[5]); o.ml = P<uint32_t>(b->num[6]); o.bl = P<uint32_t>(b->num[7]);
		o.tnoff = P<uint32_t>(b->tnoff); o.qlen = P<uint32_t>(b->qlen); o.tlen = P<uint32_t>(b->tlen);
		o.hq = P<uint64_t>(b->hq); o.ht = P<uint64_t>(b->ht); o.qslot = P<uint32_t>(b->qslot); o.tslot = P<uint32_t>(b->tslot);
	}
	size_t n_valid = 0, n_pass = 0, n_nobl = 0, n_long = 0;
	uint32_t max_qs = 0;
	if (L) {
		CHK(dev_reserve(c, c->keep, ((size_t)L + 16) * 4)); CHK(dev_reserve(c, c->pos, ((size_t)L + 16) * 4));
		lap("columns reserved");
		if (old_path) {
			ProfScope ps(c, "k_paf_parse", (double)n + 61.0 * (double)L);
			// LDS tile: 1.5 x the mean text of 256 lines, in 4 KiB steps (blocks whose lines are longer read global memory)
			uint32_t lds_bytes = (uint32_t)((double)n / (double)L * 256.0 * 1.5);
			lds_bytes = (lds_bytes + 4095u) & ~4095u;
			if (lds_bytes < 8192u) lds_bytes = 8192u;
			if (lds_bytes > PAF_LDS_BYTES) lds_bytes = PAF_LDS_BYTES;
			hipLaunchKernelGGL(k_paf_parse, dim3(grid_for(L, 256)), dim3(256), lds_bytes + 32, c->st, text, n, (const uint64_t*)P<uint64_t>(b->lstart), L, min_span, min_match, o, P<uint32_t>(c->keep), ctr, lds_bytes);
		} else {
			ProfScope ps(c, "k_paf_parse", (double)n + 69.0 * (double)L);
			TileArgs ta;
			ta.text = text; ta.n = n_eff; ta.K = tile_k; ta.n_gran = n_gran; ta.n_tiles = n_tiles; ta.L = L;
			ta.goff = (const uint32_t*)P<uint32_t>(b->tile); ta.first = (const unsigned long long*)P<unsigned long long>(b->tfirst);
			ta.min_span = min_span; ta.min_match = min_match;
			const int ch = tile_k <= 15 ? 1 : 2;
			const uint32_t reg = (uint32_t)ch * 16384u, lds = reg + 32 + 2 * (reg / 16 + 8) * 2 + 260 * 4;
			uint32_t per_cu = (160u * 1024u) / (lds + 64u); if (per_cu > 8) per_cu = 8; // blocks a CU holds: the grid is what fits the chip, a block works through tiles (one set of counter atomics per block)
			const unsigned grid = grid_for(n_tiles, 1, 256u * per_cu);
			if (ch == 1) hipLaunchKernelGGL(k_paf_parse_tile<1>, dim3(grid), dim3(256), lds, c->st, ta, o, P<uint64_t>(b->lstart), ctr);
			else hipLaunchKernelGGL(k_paf_parse_tile<2>, dim3(grid), dim3(256), lds, c->st, ta, o, P<uint64_t>(b->lstart), ctr);
		}
		CHK(ctr_fetch(c));
		if (!old_path && c->h_ctr[PC_ODD]) { // lines the straight-line parser does not cover: the byte-wise routine on them (the counters add up)
			ProfScope ps(c, "k_paf_parse_odd", 0.0);
			hipLaunchKernelGGL(k_paf_parse_odd, dim3(grid_for(L, 256, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, text, (const uint64_t*)P<uint64_t>(b->lstart), L, min_span, min_match, o, ctr);
			CHK(ctr_fetch(c));
		}
		n_valid = (size_t)c->h_ctr[PC_VALID]; n_pass = (size_t)c->h_ctr[PC_PASS]; n_nobl = (size_t)c->h_ctr[PC_NOBL];
		n_long = old_path ? n_pass : (size_t)c->h_ctr[PC_LONG];
		max_qs = (uint32_t)c->h_ctr[PC_MAXQS];
		if (n_nobl && !sharded) { // stale bl: rare (PAF writers emit 12+ columns)
			CHK(dev_reserve(c, b->blv, ((size_t)L + 4) * 4));
			if (!old_path) hipLaunchKernelGGL(k_paf_hasbl, dim3(grid_for(L, 256)), dim3(256), 0, c->st, (const uint8_t*)o.flags, L, P<uint32_t>(c->keep));
			CHK(scan_exclusive_u32(c, P<uint32_t>(c->keep), P<uint32_t>(c->pos), L, nullptr));
			hipLaunchKernelGGL(k_paf_bl_compact, dim3(grid_for(L, 256)), dim3(256), 0, c->st, (const uint32_t*)P<uint32_t>(c->keep), (const uint32_t*)P<uint32_t>(c->pos), (const uint32_t*)o.bl, L, P<uint32_t>(b->blv));
			hipLaunchKernelGGL(k_paf_bl_fill, dim3(grid_for(L, 256)), dim3(256), 0, c->st, (const uint32_t*)P<uint32_t>(c->keep), (const uint32_t*)P<uint32_t>(c->pos), (const uint32_t*)P<uint32_t>(b->blv), L, o.bl);
		}
	}
	lap("fields");
	uint64_t nobl_total = n_nobl;
	if (sharded) { // what the ranges have to know of each other before names and records can be numbered
		uint64_t mine[5] = { L, n_valid, n_pass, n_nobl, max_qs }, all[5 * 32];
		CHK(mahip_comm_all_gather_u64(c, mine, 5, all));
		nobl_total = 0;
		for (int r = 0; r < W; ++r) {
			if (r < me) line_base += all[5 * r];
			lines_total += all[5 * r]; valid_total += all[5 * r + 1]; pass_total += all[5 * r + 2]; nobl_total += all[5 * r + 3];
			if (all[5 * r + 4] > max_qs) max_qs = (uint32_t)all[5 * r + 4];
		}
		if (lines_total + 1 >= 0x7fffffffull) { mahip_set_error("mahip_paf_parse_sharded: more than 2^31 lines"); return -1; }
		if (nobl_total) { // somebody has a 10-column line: every range says what `bl` it leaves behind (the bl of its last 11-column line), and a line in front of
			// a range's first 11-column line inherits from the nearest range before it that has one (paf.c:54: the field is simply not written)
			uint64_t two[2] = { 0, 0 }, every[2 * 32];
			uint32_t n_has = 0, last_bl = 0;
			if (L) {
				CHK(dev_reserve(c, b->blv, ((size_t)L + 4) * 4));
				if (!old_path) hipLaunchKernelGGL(k_paf_hasbl, dim3(grid_for(L, 256)), dim3(256), 0, c->st, (const uint8_t*)o.flags, L, P<uint32_t>(c->keep));
				CHK(scan_exclusive_u32(c, P<uint32_t>(c->keep), P<uint32_t>(c->pos), L, P<uint32_t>(b->scal)));
				hipLaunchKernelGGL(k_paf_bl_compact, dim3(grid_for(L, 256)), dim3(256), 0, c->st, (const uint32_t*)P<uint32_t>(c->keep), (const uint32_t*)P<uint32_t>(c->pos), (const uint32_t*)o.bl, L, P<uint32_t>(b->blv));
				HIPCHK(hipMemcpyAsync(&n_has, b->scal.p, 4, hipMemcpyDeviceToHost, c->st));
				HIPCHK(hipStreamSynchronize(c->st));
				if (n_has) { HIPCHK(hipMemcpyAsync(&last_bl, P<uint32_t>(b->blv) + (n_has - 1), 4, hipMemcpyDeviceToHost, c->st)); HIPCHK(hipStreamSynchronize(c->st)); }
			}
			two[0] = n_has ? 1 : 0; two[1] = last_bl;
			CHK(mahip_comm_all_gather_u64(c, two, 2, every));
			uint32_t before = 0;
			for (int r = 0; r < me; ++r) if (every[2 * r]) before = (uint32_t)every[2 * r + 1];
			if (L && n_nobl) hipLaunchKernelGGL(k_bl_fill_from, dim3(grid_for(L, 256)), dim3(256), 0, c->st, (const uint32_t*)P<uint32_t>(c->keep), (const uint32_t*)P<uint32_t>(c->pos), (const uint32_t*)P<uint32_t>(b->blv), L, o.bl, before);
		}
	}

	// ---- dictionary: distinct names, ids in order of first appearance
	uint32_t R = 0, cap_used = 0;
	int gen_local = 0;
	if (n_pass) {
		// The number of distinct names is not known before the pass (<= 2 per stored line; in overlap files a read has tens of lines).  Start with a
		// table sized for 16 lines per name -- 8x smaller than the safe size, it stays in the last-level cache -- count the names as they go in, and
		// repeat the pass with a table for 4x that many only if the load factor came out above 1/2 (or a probe sequence ran out)
		const uint32_t cap_max = pow2_at_least(4 * (uint64_t)n_pass + 65536); // load <= 1/2 whatever the file holds
		uint32_t cap = pow2_at_least(n_pass / 16 + 65536);
		if (const char *e = getenv("MA_DICT_CAP_LOG2")) { int l2 = atoi(e); if (l2 >= 4 && l2 <= 31) cap = 1u << l2; } // tests: force the growth path
		const bool short_names = !old_path && n_long == 0 && !getenv("MA_DICT_EXACT_TEXT"); // every name is its own key: no text compared (k_dict_insert_short)
		for (int attempt = 0;; ++attempt) {
			CHK(dev_reserve(c, b->tab, (size_t)cap * 16)); CHK(dev_reserve(c, b->tmin, (size_t)cap * 4)); CHK(dev_reserve(c, b->slot_id, (size_t)cap * 4));
			CHK(ctr_zero(c));
			if (short_names) {
				HIPCHK(hipMemsetAsync(b->tab.p, 0, (size_t)cap * 16, c->st));
				ProfScope ps(c, "k_dict_insert", 2.0 * 40.0 * (double)n_pass);
				hipLaunchKernelGGL(k_dict_insert_short, dim3(grid_for(L, 256, 8192)), dim3(256), 0, c->st, o, L, (DSlot*)b->tab.p, cap - 1, ctr);
			} else {
				HIPCHK(hipMemsetAsync(b->tab.p, 0xff, (size_t)cap * 16, c->st));
				ProfScope ps(c, "k_dict_insert", 2.0 * 40.0 * (double)n_pass);
				hipLaunchKernelGGL(k_dict_insert, dim3(grid_for(L, 256, 8192)), dim3(256), 0, c->st, text, (const uint64_t*)P<uint64_t>(b->lstart), L, o, (XSlot*)b->tab.p, cap - 1, ctr, old_path ? 0 : 1);
			}
			CHK(ctr_fetch(c));
			const uint64_t distinct = c->h_ctr[PC_DISTINCT];
			if (c->h_ctr[PC_OVERFLOW] == 0 && 2 * distinct <= cap) break;
			if (attempt >= 3 || cap >= cap_max) { if (c->h_ctr[PC_OVERFLOW] == 0) break; mahip_set_error("mahip_paf_parse: name table overflow"); return -1; }
			uint32_t want = pow2_at_least(4 * distinct + 65536);
			if (want <= cap) want = cap < 0x10000000u ? cap << 3 : cap_max; // a probe sequence ran out: the count is incomplete
			cap = want < cap_max ? want : cap_max;
		}
		if (short_names) hipLaunchKernelGGL(k_dict_short_tmin, dim3(grid_for(cap, 256)), dim3(256), 0, c->st, (const DSlot*)b->tab.p, cap, P<uint32_t>(b->tmin));
		else hipLaunchKernelGGL(k_dict_exact_tmin, dim3(grid_for(cap, 256)), dim3(256), 0, c->st, (const XSlot*)b->tab.p, cap, P<uint32_t>(b->tmin));
		if (no_cont) { // hit.c:38-68 + hit.c:86
			CHK(dev_reserve(c, b->excl, (size_t)cap + 16));
			HIPCHK(hipMemsetAsync(b->excl.p, 0, cap, c->st));
			hipLaunchKernelGGL(k_paf_nocont, dim3(grid_for(L, 256, 8192)), dim3(256), 0, c->st, o, L, max_hang, int_frac, P<uint8_t>(b->excl));
			HIPCHK(hipMemsetAsync(b->tmin.p, 0xff, (size_t)cap * 4, c->st));
			CHK(ctr_zero(c));
			hipLaunchKernelGGL(k_paf_refilter, dim3(grid_for(L, 256, 8192)), dim3(256), 0, c->st, o, L, (const uint8_t*)P<uint8_t>(b->excl), P<uint32_t>(b->tmin), ctr);
			hipLaunchKernelGGL(k_excl_count, dim3(grid_for(cap, 256, 2048)), dim3(256), 0, c->st, (const uint8_t*)P<uint8_t>(b->excl), cap, ctr);
			CHK(ctr_fetch(c));
			n_pass = (size_t)c->h_ctr[PC_PASS];
			info->n_excl = (uint32_t)c->h_ctr[PC_VALID];
		}
		CHK(dev_reserve(c, c->keep, ((size_t)cap + 16) * 4)); CHK(dev_reserve(c, c->pos, ((size_t)cap + 16) * 4));
		hipLaunchKernelGGL(k_dict_flag, dim3(grid_for(cap, 256)), dim3(256), 0, c->st, (const uint32_t*)P<uint32_t>(b->tmin), cap, P<uint32_t>(c->keep));
		CHK(scan_exclusive_u32(c, P<uint32_t>(c->keep), P<uint32_t>(c->pos), cap, P<uint32_t>(b->scal)));
		HIPCHK(hipMemcpyAsync(&R, b->scal.p, 4, hipMemcpyDeviceToHost, c->st));
		HIPCHK(hipStreamSynchronize(c->st));
		for (int k = 0; k < 2; ++k) { CHK(dev_reserve(c, c->key[k], ((size_t)R + 1) * 8)); CHK(dev_reserve(c, c->val[k], ((size_t)R + 1) * 4)); }
		hipLaunchKernelGGL(k_dict_collect, dim3(grid_for(cap, 256)), dim3(256), 0, c->st, (const uint32_t*)P<uint32_t>(c->keep), (const uint32_t*)P<uint32_t>(c->pos), (const uint32_t*)P<uint32_t>(b->tmin), cap,
		                   P<uint64_t>(c->key[0]), P<uint32_t>(c->val[0]));
		int gen = 0;
		cap_used = cap;
		CHK(radix_sort_pairs(c, R, 0, bits_of(2ull * L), 0, 0, &gen));
		gen_local = gen;
		CHK(dev_reserve(c, b->seq_len, ((size_t)R + 4) * 4)); CHK(dev_reserve(c, b->name_off, ((size_t)R + 4) * 8));
		CHK(dev_reserve(c, b->name_len, ((size_t)R + 4) * 4)); CHK(dev_reserve(c, b->name_pos, ((size_t)R + 4) * 4));
		hipLaunchKernelGGL(k_dict_assign, dim3(grid_for(R, 256)), dim3(256), 0, c->st, (const uint64_t*)P<uint64_t>(c->key[gen]), (const uint32_t*)P<uint32_t>(c->val[gen]), R,
		                   (const uint64_t*)P<uint64_t>(b->lstart), o, P<uint32_t>(b->slot_id), P<uint32_t>(b->seq_len), P<uint64_t>(b->name_off), P<uint32_t>(b->name_len), P<uint32_t>(c->keep));
		uint32_t nb = 0;
		CHK(scan_exclusive_u32(c, P<uint32_t>(c->keep), P<uint32_t>(b->name_pos), R, P<uint32_t>(b->scal)));
		HIPCHK(hipMemcpyAsync(&nb, b->scal.p, 4, hipMemcpyDeviceToHost, c->st));
		HIPCHK(hipStreamSynchronize(c->st));
		CHK(dev_reserve(c, b->names, (size_t)nb + 16));
		hipLaunchKernelGGL(k_dict_names, dim3(grid_for(R, 256)), dim3(256), 0, c->st, text, (const uint64_t*)P<uint64_t>(b->name_off), (const uint32_t*)P<uint32_t>(b->name_len),
		                   (const uint32_t*)P<uint32_t>(b->name_pos), R, P<char>(b->names));
		b->name_bytes = nb;
	}
	b->n_seq = R;
	lap("dictionary");
	const uint32_t *slot_to_id = P<uint32_t>(b->slot_id); // table slot -> id, for the records
	if (sharded) { // ---- the ranks' name tables -> one dictionary (kernels above)
		const uint32_t R_loc = R;
		const size_t nb_loc = b->name_bytes;
		uint64_t mine[2] = { R_loc, nb_loc }, all[2 * 32];
		CHK(mahip_comm_all_gather_u64(c, mine, 2, all));
		uint32_t stride_rows = 1, h_rows[32];
		size_t stride_bytes = 16, sum_rows = 0;
		for (int r = 0; r < W; ++r) { h_rows[r] = (uint32_t)all[2 * r]; sum_rows += all[2 * r]; if (all[2 * r] > stride_rows) stride_rows = (uint32_t)all[2 * r]; if (all[2 * r + 1] > stride_bytes) stride_bytes = (size_t)all[2 * r + 1]; }
		stride_bytes = (stride_bytes + 15) & ~(size_t)15;
		if ((uint64_t)stride_rows * (uint64_t)W >= 0xffffffffull) { mahip_set_error("mahip_paf_parse_sharded: too many names"); return -1; }
		// my rows and name bytes into exchange buffers, gathered with the stride of the largest range
		DevBuf rows_all, blobs_all, aux;
		void *xr = nullptr, *xb = nullptr;
		CHK(mahip_xbuf(c, 0, (size_t)stride_rows * sizeof(NameRow) + stride_bytes, &xr));
		xb = (char*)xr + (size_t)stride_rows * sizeof(NameRow);
		if (R_loc) {
			hipLaunchKernelGGL(k_name_rows, dim3(grid_for(R_loc, 256)), dim3(256), 0, c->st, (const uint64_t*)P<uint64_t>(c->key[gen_local]), R_loc, (uint32_t)(2 * line_base), (const uint32_t*)P<uint32_t>(b->seq_len),
			                   (const uint32_t*)P<uint32_t>(b->name_len), (const uint32_t*)P<uint32_t>(b->name_pos), (NameRow*)xr);
			HIPCHK(hipMemcpyAsync(xb, b->names.p, nb_loc, hipMemcpyDeviceToDevice, c->st));
		}
		int rc = 0;
		do {
			if ((rc = dev_reserve(c, rows_all, (size_t)W * stride_rows * sizeof(NameRow) + 64)) != 0) break;
			if ((rc = dev_reserve(c, blobs_all, (size_t)W * stride_bytes + 64)) != 0) break;
			if ((rc = mahip_comm_all_gather(c, xr, rows_all.p, (size_t)stride_rows * sizeof(NameRow))) != 0) break;
			if ((rc = mahip_comm_all_gather(c, xb, blobs_all.p, stride_bytes)) != 0) break;
			// one table for all of them: at most sum_rows distinct names
			const uint32_t gcap = pow2_at_least(2 * (uint64_t)sum_rows + 1024);
			const size_t total_rows = (size_t)W * stride_rows;
			// aux: tab[gcap] u64 | gkey[gcap] u64 | slot_of[total_rows] u32 | gid_of_slot[gcap] u32 | n_rows[32] u32
			const size_t o_key = (size_t)gcap * 8, o_slot = o_key + (size_t)gcap * 8, o_gid = o_slot + ((total_rows * 4 + 7) & ~(size_t)7), o_n = o_gid + (size_t)gcap * 4;
			if ((rc = dev_reserve(c, aux, o_n + 32 * 4 + 64)) != 0) break;
			unsigned long long *gtab = (unsigned long long*)aux.p, *gkey = (unsigned long long*)((char*)aux.p + o_key);
			uint32_t *slot_of = (uint32_t*)((char*)aux.p + o_slot), *gid_of_slot = (uint32_t*)((char*)aux.p + o_gid), *d_rows = (uint32_t*)((char*)aux.p + o_n);
			HIPCHK(hipMemsetAsync(aux.p, 0xff, o_slot, c->st));
			HIPCHK(hipMemcpyAsync(d_rows, h_rows, (size_t)W * 4, hipMemcpyHostToDevice, c->st));
			CHK(ctr_zero(c));
			if (sum_rows) hipLaunchKernelGGL(k_dict_merge, dim3(grid_for(total_rows, 256, 8192)), dim3(256), 0, c->st, (const NameRow*)rows_all.p, (const unsigned char*)blobs_all.p, (const uint32_t*)d_rows, W,
			                                 stride_rows, stride_bytes, gtab, gkey, gcap - 1, slot_of, ctr);
			CHK(ctr_fetch(c));
			if (c->h_ctr[PC_OVERFLOW]) { mahip_set_error("mahip_paf_parse_sharded: name table overflow"); rc = -1; break; }
			// distinct names sorted by first appearance = ids
			if ((rc = dev_reserve(c, c->keep, ((size_t)gcap + 16) * 4)) != 0 || (rc = dev_reserve(c, c->pos, ((size_t)gcap + 16) * 4)) != 0) break;
			hipLaunchKernelGGL(k_merge_flag, dim3(grid_for(gcap, 256)), dim3(256), 0, c->st, (const unsigned long long*)gtab, gcap, P<uint32_t>(c->keep));
			if ((rc = scan_exclusive_u32(c, P<uint32_t>(c->keep), P<uint32_t>(c->pos), gcap, P<uint32_t>(b->scal))) != 0) break;
			uint32_t Rg = 0;
			HIPCHK(hipMemcpyAsync(&Rg, b->scal.p, 4, hipMemcpyDeviceToHost, c->st));
			HIPCHK(hipStreamSynchronize(c->st));
			for (int k = 0; k < 2 && rc == 0; ++k) { rc = dev_reserve(c, c->key[k], ((size_t)Rg + 1) * 8); if (rc == 0) rc = dev_reserve(c, c->val[k], ((size_t)Rg + 1) * 4); }
			if (rc) break;
			hipLaunchKernelGGL(k_merge_collect, dim3(grid_for(gcap, 256)), dim3(256), 0, c->st, (const uint32_t*)P<uint32_t>(c->keep), (const uint32_t*)P<uint32_t>(c->pos), (const unsigned long long*)gkey, gcap,
			                   P<uint64_t>(c->key[0]), P<uint32_t>(c->val[0]));
			int gen = 0;
			if ((rc = radix_sort_pairs(c, Rg, 0, bits_of(2ull * lines_total), 0, 0, &gen)) != 0) break;
			// the map for this rank's records BEFORE the local arrays are overwritten: local slot -> global id (k_merge_map needs the local "slot in use" flags)
			if (cap_used) {
				if ((rc = dev_reserve(c, b->excl, (size_t)cap_used * 4 + 16)) != 0) break; // (the -R flag array is free in this mode: the map lives there)
			}
			if ((rc = dev_reserve(c, b->seq_len, ((size_t)Rg + 4) * 4)) != 0 || (rc = dev_reserve(c, b->name_off, ((size_t)Rg + 4) * 8)) != 0 ||
			    (rc = dev_reserve(c, b->name_len, ((size_t)Rg + 4) * 4)) != 0 || (rc = dev_reserve(c, b->name_pos, ((size_t)Rg + 4) * 4)) != 0) break;
			// keep / pos are about to be reused for the name lengths: the local flags first
			DevBuf used_local;
			if (cap_used) {
				if ((rc = dev_reserve(c, used_local, (size_t)cap_used * 4 + 16)) != 0) break;
				hipLaunchKernelGGL(k_dict_flag, dim3(grid_for(cap_used, 256)), dim3(256), 0, c->st, (const uint32_t*)P<uint32_t>(b->tmin), cap_used, (uint32_t*)used_local.p);
			}
			if ((rc = dev_reserve(c, c->keep, ((size_t)Rg + 16) * 4)) != 0) { dev_free(c, used_local); break; }
			if (Rg) hipLaunchKernelGGL(k_merge_assign, dim3(grid_for(Rg, 256)), dim3(256), 0, c->st, (const uint32_t*)P<uint32_t>(c->val[gen]), Rg, (const unsigned long long*)gtab, (const unsigned long long*)gkey,
			                           (const NameRow*)rows_all.p, stride_rows, stride_bytes, gid_of_slot, P<uint32_t>(b->seq_len), P<uint64_t>(b->name_off), P<uint32_t>(b->name_len), P<uint32_t>(c->keep));
			if (cap_used) hipLaunchKernelGGL(k_merge_map, dim3(grid_for(cap_used, 256)), dim3(256), 0, c->st, (const uint32_t*)used_local.p, (const uint32_t*)P<uint32_t>(b->slot_id), cap_used,
			                                 (const uint32_t*)slot_of, (uint32_t)((size_t)me * stride_rows), (const uint32_t*)gid_of_slot, P<uint32_t>(b->excl));
			uint32_t nb = 0;
			if ((rc = scan_exclusive_u32(c, P<uint32_t>(c->keep), P<uint32_t>(b->name_pos), Rg, P<uint32_t>(b->scal))) != 0) { dev_free(c, used_local); break; }
			HIPCHK(hipMemcpyAsync(&nb, b->scal.p, 4, hipMemcpyDeviceToHost, c->st));
			HIPCHK(hipStreamSynchronize(c->st));
			dev_free(c, used_local);
			if ((rc = dev_reserve(c, b->names, (size_t)nb + 16)) != 0) break;
			if (Rg) hipLaunchKernelGGL(k_dict_names, dim3(grid_for(Rg, 256)), dim3(256), 0, c->st, (const unsigned char*)blobs_all.p, (const uint64_t*)P<uint64_t>(b->name_off), (const uint32_t*)P<uint32_t>(b->name_len),
			                           (const uint32_t*)P<uint32_t>(b->name_pos), Rg, P<char>(b->names));
			HIPCHK(hipStreamSynchronize(c->st)); // (the gathered blocks are released below)
			b->name_bytes = nb;
			R = Rg; b->n_seq = Rg;
			slot_to_id = (const uint32_t*)P<uint32_t>(b->excl);
		} while (0);
		dev_free(c, rows_all); dev_free(c, blobs_all); dev_free(c, aux);
		if (rc) return -1;
	}

	// ---- records: hit (+ mirrored hit) per stored line, in line order
	size_t n_hits = 0;
	if (n_pass && old_path) {
		CHK(dev_reserve(c, c->keep, ((size_t)L + 16) * 4)); CHK(dev_reserve(c, c->pos, ((size_t)L + 16) * 4));
		hipLaunchKernelGGL(k_paf_ids, dim3(grid_for(L, 256)), dim3(256), 0, c->st, o, slot_to_id, L, bi_dir, P<uint32_t>(c->keep));
		uint32_t nh = 0;
		CHK(scan_exclusive_u32(c, P<uint32_t>(c->keep), P<uint32_t>(c->pos), L, P<uint32_t>(b->scal)));
		HIPCHK(hipMemcpyAsync(&nh, b->scal.p, 4, hipMemcpyDeviceToHost, c->st));
		HIPCHK(hipStreamSynchronize(c->st));
		n_hits = nh;
		CHK(mahip_hits_adopt(c, nullptr, n_hits, R)); // resets the per-upload state and sizes the read arrays
		CHK(dev_reserve(c, c->aos_own, (n_hits + 1) * sizeof(ma_hit_t)));
		c->d_aos = (const ma_hit_t*)c->aos_own.p;
		if (n_hits) {
			ProfScope ps(c, "k_paf_emit", 45.0 * (double)n_pass + 32.0 * (double)n_hits);
			hipLaunchKernelGGL(k_paf_emit, dim3(grid_for(L, 256)), dim3(256), 0, c->st, o, (const uint32_t*)P<uint32_t>(c->keep), (const uint32_t*)P<uint32_t>(c->pos), L, (uint4*)c->aos_own.p);
		}
	} else if (n_pass) { // one pass: ids, record slots (chained tiles), records
		const size_t max_hits = bi_dir ? 2 * n_pass : n_pass;
		if (max_hits >= 0xffffffffull) { mahip_set_error("mahip_paf_parse: more than 2^32 records"); return -1; }
		CHK(dev_reserve(c, c->aos_own, (max_hits + 1) * sizeof(ma_hit_t)));
		const size_t nb = ((size_t)L + EM_TILE - 1) / EM_TILE;
		uint32_t *ticket; unsigned long long *state; uint32_t ticket_base, epoch;
		CHK(scan_chain_begin(c, nb, &state, &ticket, &ticket_base, &epoch));
		uint32_t nh = 0;
		{
			ProfScope ps(c, "k_paf_emit", 37.0 * (double)n_pass + 32.0 * (double)max_hits);
			hipLaunchKernelGGL(k_paf_emit_chain, dim3((unsigned)nb), dim3(256), 0, c->st, o, slot_to_id, L, bi_dir, (uint4*)c->aos_own.p, P<uint32_t>(b->scal), state, ticket, ticket_base, epoch);
		}
		HIPCHK(hipMemcpyAsync(&nh, b->scal.p, 4, hipMemcpyDeviceToHost, c->st));
		HIPCHK(hipStreamSynchronize(c->st));
		n_hits = nh;
		lap("records");
		CHK(mahip_hits_adopt(c, nullptr, n_hits, R)); // resets the per-upload state and sizes the read arrays
		c->d_aos = (const ma_hit_t*)c->aos_own.p;
		lap("adopt");
	} else {
		CHK(mahip_hits_adopt(c, nullptr, 0, R));
		CHK(dev_reserve(c, c->aos_own, sizeof(ma_hit_t)));
		c->d_aos = (const ma_hit_t*)c->aos_own.p;
	}
	HIPCHK(hipGetLastError());
	HIPCHK(hipStreamSynchronize(c->st));
	c->hint_max_qs = c->paf_max_qs = max_qs;
	c->run_stride = sharded ? 0 : bi_dir ? 2 : 1; // k_paf_emit wrote a line's record and its mirror side by side (hit.c:87-98): the sort may take RUNS of records (hits.hip)
	info->n_records = n_valid; info->n_stored_lines = n_pass; info->n_hits = n_hits; info->n_seq = R; info->max_qs = max_qs; info->name_bytes = b->name_bytes; info->n_lines = L;
	if (sharded) { info->n_records = valid_total; info->n_stored_lines = pass_total; info->n_lines = lines_total; }
	return 0;
}

extern "C" int mahip_paf_parse_excl(mahip_ctx_t *c, int min_span, int min_match, int bi_dir, int no_cont, int max_hang, float int_frac, mahip_paf_info_t *info)
{
	return paf_parse_impl(c, min_span, min_match, bi_dir, no_cont, max_hang, int_frac, info, false);
}
extern "C" int mahip_paf_parse_sharded(mahip_ctx_t *c, int min_span, int min_match, int bi_dir, mahip_paf_info_t *info)
{
	return paf_parse_impl(c, min_span, min_match, bi_dir, 0, 0, 0.f, info, mahip_comm_active(c) != 0);
}

extern "C" int mahip_paf_names(mahip_ctx_t *c, char *names, uint32_t *lens)
{
	HIPCHK(hipSetDevice(c->dev));
	PafBufs *b = paf_of(c);
	if (b->n_seq == 0) return 0;
	if (names) CHK(xfer_copy(c, b->names.p, names, b->name_bytes, 0));
	if (lens) CHK(xfer_copy(c, b->seq_len.p, lens, (size_t)b->n_seq * 4, 0));
	return 0;
}

// The host dictionary without per-name host work: sd_seq_t records (sdict.h:6-10: name pointer, len, aux:31 | del:1 -- 16 bytes) written by the
// device FOR the block the names are about to be copied into (names_host: its address on the host), so that both arrive as two plain copies.
__global__ __launch_bounds__(256) void k_dict_seqs(unsigned long long names_host, const uint32_t *__restrict__ name_pos, const uint32_t *__restrict__ seq_len, uint32_t R,
                                                    uint4 *__restrict__ out, unsigned long long *__restrict__ ctr)
{
	uint64_t tot = 0;
	for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < R; j += gridDim.x * 256u) {
		const unsigned long long p = names_host + name_pos[j];
		out[j] = make_uint4((uint32_t)p, (uint32_t)(p >> 32), seq_len[j], 0u);
		tot += seq_len[j];
	}
	blk_add_u64(&ctr[PC_VALID], tot);
}
extern "C" int mahip_paf_seqs(mahip_ctx_t *c, char *names, void *seqs16, uint64_t *tot_len)
{
	HIPCHK(hipSetDevice(c->dev));
	PafBufs *b = paf_of(c);
	if (tot_len) *tot_len = 0;
	if (b->n_seq == 0) return 0;
	const uint32_t R = b->n_seq;
	CHK(dev_reserve(c, c->key[0], ((size_t)R + 1) * 16));
	CHK(ctr_zero(c));
	hipLaunchKernelGGL(k_dict_seqs, dim3(grid_for(R, 256, 1024)), dim3(256), 0, c->st, (unsigned long long)(uintptr_t)names, (const uint32_t*)P<uint32_t>(b->name_pos),
	                   (const uint32_t*)P<uint32_t>(b->seq_len), R, (uint4*)c->key[0].p, P<unsigned long long>(c->ctr));
	CHK(ctr_fetch(c));
	if (tot_len) *tot_len = c->h_ctr[PC_VALID];
	CHK(xfer_copy(c, b->names.p, names, b->name_bytes, 0));
	CHK(xfer_copy(c, c->key[0].p, seqs16, (size_t)R * 16, 0));
	return 0;
}

// the text and the per-line columns are only needed until the records and the names are out
extern "C" int mahip_paf_release(mahip_ctx_t *c)
{
	HIPCHK(hipSetDevice(c->dev));
	HIPCHK(hipStreamSynchronize(c->st));
	paf_free(c);
	return 0;
}

extern "C" uint32_t mahip_paf_max_qs(mahip_ctx_t *c) { return c->paf_max_qs; }
