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
	DevBuf *all[] = { &b->text, &b->lstart, &b->tile, &b->flags, &b->tnoff, &b->qlen, &b->tlen, &b->hq, &b->ht, &b->qslot, &b->tslot, &b->tab, &b->tmin, &b->info,
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

// FNV-1a of a name column up to its first NUL (the reference handles names as C strings); *len = bytes hashed
template <typename PTR>
__device__ __forceinline__ uint64_t paf_name(PTR &p, uint32_t beg, uint32_t end, uint32_t *len)
{
	uint64_t h = FNV_OFF;
	uint32_t pos = beg;
	for (; pos < end; ++pos) { const unsigned ch = p[pos]; if (ch == 0) break; h = (h ^ ch) * FNV_PRIME; }
	*len = pos - beg;
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
// One thread per stored line: query name, then target name.  Slot word = tag(32) | occurrence of the name's first
// inserter; tmin[slot] = smallest occurrence (2*line + column) of the name = its first appearance in the file.
// info[slot] = text offset << 24 | length of the slot's name, written by the inserter right after its CAS: a prober that finds it compares
// the bytes after ONE dependent fetch; one that does not see it yet (the store is not ordered with the CAS, a stale L1 line) goes the
// long way through the occurrence (line start, column offset, length: three more random fetches) -- both ways read the same bytes.
#define PAF_INFO_LEN_BITS 24
// the slot of one name occurrence (probe, insert if new); 0xffffffff: the probe sequence ran out
__device__ __forceinline__ uint32_t dict_probe(const unsigned char *__restrict__ text, const uint64_t *__restrict__ lstart, const PafCols &o,
                                               unsigned long long *__restrict__ tab, uint32_t *__restrict__ tmin, unsigned long long *__restrict__ info, uint32_t mask,
                                               uint64_t h, uint32_t len, uint32_t occ, uint64_t noff, uint32_t *fresh)
{
	const unsigned char *nm = text + noff;
	const uint32_t tag = (uint32_t)(h >> 32);
	uint32_t s = (uint32_t)h & mask;
	for (uint32_t probe = 0; probe < PAF_PROBE_LIMIT; ++probe, s = (s + 1) & mask) {
		unsigned long long e = tab[s];
		if (e == PAF_EMPTY) {
			e = atomicCAS(&tab[s], PAF_EMPTY, (unsigned long long)tag << 32 | occ);
			if (e == PAF_EMPTY) {
				if (len < (1u << PAF_INFO_LEN_BITS) && noff < (1ull << (64 - PAF_INFO_LEN_BITS))) info[s] = noff << PAF_INFO_LEN_BITS | len;
				atomicMin(&tmin[s], occ); ++*fresh;
				return s;
			}
		}
		if ((uint32_t)(e >> 32) == tag) {
			const unsigned long long inf = info[s];
			bool same;
			if (inf != PAF_EMPTY) same = (uint32_t)(inf & ((1u << PAF_INFO_LEN_BITS) - 1)) == len && name_eq(nm, text + (inf >> PAF_INFO_LEN_BITS), len);
			else {
				const uint32_t r = (uint32_t)e, rl = r >> 1;
				const uint32_t rlen = (r & 1) ? o.tlen[rl] : o.qlen[rl];
				same = rlen == len && name_eq(nm, text + lstart[rl] + ((r & 1) ? o.tnoff[rl] : 0u), len);
			}
			if (same) {
				if (tmin[s] > occ) atomicMin(&tmin[s], occ);
				return s;
			}
		}
	}
	return 0xffffffffu;
}

// A PAF file lists a query's overlaps together (the reference's own all-vs-all pipeline writes them so; so does every overlapper that works query by
// query): the QUERY name of a line is, 49 times in 50 at BASELINE coverage, the previous line's.  A wave holds 64 consecutive lines; a lane whose query
// name equals its left neighbour's (same hash, same length, same bytes) does not probe but takes the slot of the nearest lane to its left that did (the
// head of its run: its occurrence number is the run's smallest, so tmin is right as well).  Round 2 probed once per name occurrence: 200 M probes and
// 68 GB of fetches for 100 M lines; the query column now costs one probe per run and wave.  Target names change from line to line and probe as before.
__global__ __launch_bounds__(256) void k_dict_insert(const unsigned char *__restrict__ text, const uint64_t *__restrict__ lstart, uint32_t L, PafCols o,
                                                      unsigned long long *__restrict__ tab, uint32_t *__restrict__ tmin, unsigned long long *__restrict__ info,
                                                      uint32_t mask, unsigned long long *__restrict__ ctr)
{
	uint32_t fail = 0, fresh = 0;
	const unsigned lane = threadIdx.x & 63;
	for (uint32_t base = blockIdx.x * 256u; base < L; base += gridDim.x * 256u) { // wave-uniform: the lanes talk to each other below
		const uint32_t i = base + threadIdx.x;
		const bool stored = i < L && (o.flags[i] & 2);
		const uint64_t ls = stored ? lstart[i] : 0;
		const uint64_t hq = stored ? o.hq[i] : 0;
		const uint32_t qlen = stored ? o.qlen[i] : 0;
		// does this line continue its left neighbour's run of one query name?
		const uint64_t hq_l = (uint64_t)__shfl_up((uint32_t)(hq >> 32), 1, 64) << 32 | __shfl_up((uint32_t)hq, 1, 64);
		const uint32_t qlen_l = __shfl_up(qlen, 1, 64);
		const uint64_t ls_l = (uint64_t)__shfl_up((uint32_t)(ls >> 32), 1, 64) << 32 | __shfl_up((uint32_t)ls, 1, 64);
		const int stored_l = __shfl_up((int)stored, 1, 64);
		const bool cont = stored && lane > 0 && stored_l && hq_l == hq && qlen_l == qlen && name_eq(text + ls, text + ls_l, qlen);
		uint32_t qslot = 0xffffffffu;
		if (stored && !cont) qslot = dict_probe(text, lstart, o, tab, tmin, info, mask, hq, qlen, i * 2u, ls, &fresh);
		const unsigned long long heads = __ballot(stored && !cont);
		{ // a continuing lane: the slot of the nearest head to its left (there is one: lane 0 never continues)
			const unsigned long long left = heads & ((1ull << lane) - 1ull);
			const int src = left ? 63 - __builtin_clzll(left) : (int)lane;
			const uint32_t got = __shfl(qslot, src, 64);
			if (cont) qslot = got;
		}
		if (stored) {
			const uint32_t tslot = dict_probe(text, lstart, o, tab, tmin, info, mask, o.ht[i], o.tlen[i], i * 2u + 1u, ls + o.tnoff[i], &fresh);
			if (qslot == 0xffffffffu || tslot == 0xffffffffu) fail = 1;
			o.qslot[i] = qslot; o.tslot[i] = tslot;
		}
	}
	blk_add_u64(&ctr[PC_OVERFLOW], fail);
	blk_add_u64(&ctr[PC_DISTINCT], fresh);
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

// a name gets an id if some STORED line carries it (with -R a name may sit in the table without such a line)
__global__ __launch_bounds__(256) void k_dict_flag(const unsigned long long *__restrict__ tab, const uint32_t *__restrict__ tmin, uint32_t cap, uint32_t *__restrict__ keep)
{
	uint32_t s = blockIdx.x * 256u + threadIdx.x;
	if (s < cap) keep[s] = tab[s] != PAF_EMPTY && tmin[s] != 0xffffffffu;
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

	// ---- line starts
	const size_t n_tiles = (n + PAF_TILE - 1) / PAF_TILE;
	if (n_tiles > 0x7fffffffull) { mahip_set_error("mahip_paf_parse: text too large"); return -1; }
	CHK(dev_reserve(c, b->tile, (n_tiles + 8) * 4));
	CHK(dev_reserve(c, b->scal, 64));
	CHK(ctr_zero(c));
	uint32_t L = 0;
	if (n) {
		{
			ProfScope ps(c, "k_paf_nl_count", (double)n);
			hipLaunchKernelGGL(k_paf_nl_count, dim3((unsigned)n_tiles), dim3(256), 0, c->st, text, n, P<uint32_t>(b->tile));
		}
		CHK(scan_exclusive_u32(c, P<uint32_t>(b->tile), P<uint32_t>(b->tile), n_tiles, P<uint32_t>(b->scal)));
		uint32_t n_nl = 0;
		HIPCHK(hipMemcpyAsync(&n_nl, b->scal.p, 4, hipMemcpyDeviceToHost, c->st));
		HIPCHK(hipStreamSynchronize(c->st));
		if ((uint64_t)n_nl + 1 >= 0x7fffffffull) { mahip_set_error("mahip_paf_parse: more than 2^31 lines"); return -1; }
		CHK(dev_reserve(c, b->lstart, ((size_t)n_nl + 4) * 8));
		{
			ProfScope ps(c, "k_paf_nl_pos", (double)n + 8.0 * (double)n_nl);
			hipLaunchKernelGGL(k_paf_nl_pos, dim3((unsigned)n_tiles), dim3(256), 0, c->st, text, n, (const uint32_t*)P<uint32_t>(b->tile), (const uint32_t*)P<uint32_t>(b->scal), P<uint64_t>(b->lstart), ctr);
		}
		int open_line = 0;
		{ // same decision as the kernel's, from the host's view of the counts: L = newlines + unterminated tail
			unsigned char last = 0;
			HIPCHK(hipMemcpyAsync(&last, text + n - 1, 1, hipMemcpyDeviceToHost, c->st));
			HIPCHK(hipStreamSynchronize(c->st));
			open_line = last != '\n';
		}
		L = n_nl + (uint32_t)open_line;
		if (!open_line) { // terminated text: line L-1 ends at the last newline; lstart[L] was written by the scatter
		}
	}
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
	size_t n_valid = 0, n_pass = 0, n_nobl = 0;
	uint32_t max_qs = 0;
	if (L) {
		CHK(dev_reserve(c, c->keep, ((size_t)L + 16) * 4)); CHK(dev_reserve(c, c->pos, ((size_t)L + 16) * 4));
		{
			ProfScope ps(c, "k_paf_parse", (double)n + 61.0 * (double)L);
			// LDS tile: 1.5 x the mean text of 256 lines, in 4 KiB steps (blocks whose lines are longer read global memory)
			uint32_t lds_bytes = (uint32_t)((double)n / (double)L * 256.0 * 1.5);
			lds_bytes = (lds_bytes + 4095u) & ~4095u;
			if (lds_bytes < 8192u) lds_bytes = 8192u;
			if (lds_bytes > PAF_LDS_BYTES) lds_bytes = PAF_LDS_BYTES;
			hipLaunchKernelGGL(k_paf_parse, dim3(grid_for(L, 256)), dim3(256), lds_bytes + 32, c->st, text, n, (const uint64_t*)P<uint64_t>(b->lstart), L, min_span, min_match, o, P<uint32_t>(c->keep), ctr, lds_bytes);
		}
		CHK(ctr_fetch(c));
		n_valid = (size_t)c->h_ctr[PC_VALID]; n_pass = (size_t)c->h_ctr[PC_PASS]; n_nobl = (size_t)c->h_ctr[PC_NOBL];
		max_qs = (uint32_t)c->h_ctr[PC_MAXQS];
		if (n_nobl && !sharded) { // stale bl: rare (PAF writers emit 12+ columns)
			CHK(dev_reserve(c, b->blv, ((size_t)L + 4) * 4));
			CHK(scan_exclusive_u32(c, P<uint32_t>(c->keep), P<uint32_t>(c->pos), L, nullptr));
			hipLaunchKernelGGL(k_paf_bl_compact, dim3(grid_for(L, 256)), dim3(256), 0, c->st, (const uint32_t*)P<uint32_t>(c->keep), (const uint32_t*)P<uint32_t>(c->pos), (const uint32_t*)o.bl, L, P<uint32_t>(b->blv));
			hipLaunchKernelGGL(k_paf_bl_fill, dim3(grid_for(L, 256)), dim3(256), 0, c->st, (const uint32_t*)P<uint32_t>(c->keep), (const uint32_t*)P<uint32_t>(c->pos), (const uint32_t*)P<uint32_t>(b->blv), L, o.bl);
		}
	}
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
		for (int attempt = 0;; ++attempt) {
			CHK(dev_reserve(c, b->tab, (size_t)cap * 8)); CHK(dev_reserve(c, b->tmin, (size_t)cap * 4)); CHK(dev_reserve(c, b->slot_id, (size_t)cap * 4));
			CHK(dev_reserve(c, b->info, (size_t)cap * 8));
			HIPCHK(hipMemsetAsync(b->tab.p, 0xff, (size_t)cap * 8, c->st));
			HIPCHK(hipMemsetAsync(b->info.p, 0xff, (size_t)cap * 8, c->st));
			HIPCHK(hipMemsetAsync(b->tmin.p, 0xff, (size_t)cap * 4, c->st));
			CHK(ctr_zero(c));
			{
				ProfScope ps(c, "k_dict_insert", 2.0 * 40.0 * (double)n_pass);
				hipLaunchKernelGGL(k_dict_insert, dim3(grid_for(L, 256, 8192)), dim3(256), 0, c->st, text, (const uint64_t*)P<uint64_t>(b->lstart), L, o,
				                   P<unsigned long long>(b->tab), P<uint32_t>(b->tmin), P<unsigned long long>(b->info), cap - 1, ctr);
			}
			CHK(ctr_fetch(c));
			const uint64_t distinct = c->h_ctr[PC_DISTINCT];
			if (c->h_ctr[PC_OVERFLOW] == 0 && 2 * distinct <= cap) break;
			if (attempt >= 3 || cap >= cap_max) { if (c->h_ctr[PC_OVERFLOW] == 0) break; mahip_set_error("mahip_paf_parse: name table overflow"); return -1; }
			uint32_t want = pow2_at_least(4 * distinct + 65536);
			if (want <= cap) want = cap < 0x10000000u ? cap << 3 : cap_max; // a probe sequence ran out: the count is incomplete
			cap = want < cap_max ? want : cap_max;
		}
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
		hipLaunchKernelGGL(k_dict_flag, dim3(grid_for(cap, 256)), dim3(256), 0, c->st, (const unsigned long long*)P<unsigned long long>(b->tab), (const uint32_t*)P<uint32_t>(b->tmin), cap, P<uint32_t>(c->keep));
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
				hipLaunchKernelGGL(k_dict_flag, dim3(grid_for(cap_used, 256)), dim3(256), 0, c->st, (const unsigned long long*)P<unsigned long long>(b->tab), (const uint32_t*)P<uint32_t>(b->tmin), cap_used, (uint32_t*)used_local.p);
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
	if (n_pass) {
		CHK(dev_reserve(c, c->keep, ((size_t)L + 16) * 4)); CHK(dev_reserve(c, c->pos, ((size_t)L + 16) * 4));
		hipLaunchKernelGGL(k_paf_ids, dim3(grid_for(L, 256)), dim3(256), 0, c->st, o, slot_to_id, L, bi_dir, P<uint32_t>(c->keep));
		uint32_t nh = 0;
		CHK(scan_exclusive_u32(c, P<uint32_t>(c->keep), P<uint32_t>(c->pos), L, P<uint32_t>(b->scal)));
		HIPCHK(hipMemcpyAsync(&nh, b->scal.p, 4, hipMemcpyDeviceToHost, c->st));
		HIPCHK(hipStreamSynchronize(c->st));
		n_hits = nh;
	}
	CHK(mahip_hits_adopt(c, nullptr, n_hits, R)); // resets the per-upload state and sizes the read arrays
	CHK(dev_reserve(c, c->aos_own, (n_hits + 1) * sizeof(ma_hit_t)));
	c->d_aos = (const ma_hit_t*)c->aos_own.p;
	if (n_hits) {
		ProfScope ps(c, "k_paf_emit", 45.0 * (double)n_pass + 32.0 * (double)n_hits);
		hipLaunchKernelGGL(k_paf_emit, dim3(grid_for(L, 256)), dim3(256), 0, c->st, o, (const uint32_t*)P<uint32_t>(c->keep), (const uint32_t*)P<uint32_t>(c->pos), L, (uint4*)c->aos_own.p);
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
