// graph.hip -- string-graph construction and reduction on a dense SoA edge list + CSR index in HBM.
//
//   arcs : au av alen aol(del<<31) u32 columns [n_arc], sorted by (u, len) like reference asg.c:22-25
//   idx  : u64 [2*n_seq], idx[v] = first<<32 | count      (reference asg.c:27-36)
//   sdel / slen : per read seq.del / seq.len               (reference asg.h:13-15)
// Vertex ids use the ORIGINAL read numbering; the squeeze map (sdict.c:69-86) is applied at export only
// (it is monotone, so every order is preserved).
//
//   mahip_sg_gen       : reference asm.c:9-39  ma_sg_gen + asg.c:72-80 asg_cleanup
//   mahip_asg_del_trans: reference asg.c:148-193 asg_arc_del_trans (Myers) : one wave per vertex, own
//                        neighbour list and an open-addressing mark table in LDS, neighbours' lists streamed
//   mahip_asg_symm     : reference asg.c:104-145 asg_arc_del_multi + asg_arc_del_asymm
//   mahip_asg_del_short: reference asg.c:83-101
#include "mahip_internal.hpp"
#include <type_traits>

#define DEAD 0x80000000u
#define ADEL 0x80000000u

struct HitColsG { const uint32_t *qid, *qs, *qe, *tn, *ts, *te, *ml, *bl; };
struct ArcCols { uint32_t *u, *v, *len, *ol; };

static HitColsG gcols_of(mahip_ctx *c)
{
	HitColsG h;
	h.qid = P<uint32_t>(c->col[0]); h.qs = P<uint32_t>(c->col[1]); h.qe = P<uint32_t>(c->col[2]); h.tn = P<uint32_t>(c->col[3]);
	h.ts = P<uint32_t>(c->col[4]); h.te = P<uint32_t>(c->col[5]); h.ml = P<uint32_t>(c->col[6]); h.bl = P<uint32_t>(c->col[7]);
	return h;
}
static ArcCols arcs_of(mahip_ctx *c, int g)
{
	ArcCols a;
	a.u = P<uint32_t>(c->au[g]); a.v = P<uint32_t>(c->av[g]); a.len = P<uint32_t>(c->alen[g]); a.ol = P<uint32_t>(c->aol[g]);
	return a;
}
static int reserve_arcs(mahip_ctx *c, size_t n)
{
	for (int g = 0; g < 2; ++g) {
		CHK(dev_reserve(c, c->au[g], (n + 4) * 4)); CHK(dev_reserve(c, c->av[g], (n + 4) * 4));
		CHK(dev_reserve(c, c->alen[g], (n + 4) * 4)); CHK(dev_reserve(c, c->aol[g], (n + 4) * 4));
	}
	CHK(dev_reserve(c, c->keep, (n + 16) * 4)); CHK(dev_reserve(c, c->pos, (n + 16) * 4));
	return 0;
}

// ------------------------------------------------------------------------------------------------ ma_sg_gen
// asm.c:14-17 : seq.len / seq.del per read.  ql used by the classifier keeps all 32 bits (asm.c:23-24).
__global__ __launch_bounds__(256) void k_sg_seq(const uint2 *__restrict__ sub, const uint8_t *__restrict__ r_del, const uint32_t *__restrict__ seq_len,
                                                 const uint8_t *__restrict__ seq_del, uint32_t n_seq, uint32_t *__restrict__ slen, uint8_t *__restrict__ sdel)
{
	uint32_t r = blockIdx.x * 256 + threadIdx.x;
	if (r >= n_seq) return;
	uint32_t len; int del;
	if (sub) { uint2 s = sub[r]; len = s.y - (s.x & 0x7fffffffu); del = s.x >> 31; }
	else len = seq_len[r], del = 0;
	if (seq_del) del |= seq_del[r];
	if (r_del) del |= r_del[r];
	slen[r] = len; sdel[r] = (uint8_t)del;
}

// asm.c:18-35 per hit.  The arcs a read pair yields are few next to the hit slots (after containment 99 % of the slots are
// dead), so nothing is materialised per slot: pass A only applies the seq.del side effects (asm.c:27-34) and takes the
// maxima and leaves one candidate bit per slot; once seq.del is final (after the exchange in the sharded mode) pass B counts the
// surviving arcs per tile of SG_TILE consecutive hits and pass C -- after a scan of the tile counts -- recomputes them and writes
// them densely, in slot order.  One sweep over the hits and two over the candidate bits instead of 16 B of arc per slot.
#define SG_TILE 16384u // hits per tile of passes B and C = 256 mask words

__device__ __forceinline__ int sg_candidate(const HitColsG &h, size_t i, const uint32_t *__restrict__ slen, int max_hang, float int_frac, int min_ovlp,
                                            const uint8_t *__restrict__ lazy_del, mc_arc_t *x, uint32_t *q_, uint32_t *t_, int *self_rc)
{ // <0: dead slot; else mc_hit2arc's verdict (r >= 0: arc in *x), *self_rc: the palindromic self hit of asm.c:27-30
	const uint32_t q = h.qid[i];
	if (lazy_del && lazy_del[q]) return -100; // first: after containment most READS are gone, and a group's lanes share q -- whole waves leave here with one column read
	if (h.bl[i] & DEAD) return -100;
	const uint32_t t = h.tn[i];
	if (lazy_del && lazy_del[t]) return -100;
	uint32_t qs = h.qs[i], qe = h.qe[i], ts = h.ts[i], te = h.te[i];
	int rev = h.ml[i] >> 31;
	*q_ = q; *t_ = t;
	*self_rc = q == t && qs == ts && qe == te && rev;
	return mc_hit2arc(q, qs, qe, t, ts, te, rev, (int)slen[q], (int)slen[t], max_hang, int_frac, min_ovlp, x);
}

__global__ __launch_bounds__(256) void k_sg_arcs(HitColsG h, size_t n, const uint32_t *__restrict__ slen, uint8_t *__restrict__ sdel,
                                                  int max_hang, float int_frac, int min_ovlp, unsigned long long *__restrict__ ctr, const uint8_t *__restrict__ lazy_del,
                                                  unsigned long long *__restrict__ cmask)
{ // pass A.  lazy_del != nullptr: the squeeze of ma_hit_contained was postponed; hits with a dropped endpoint are skipped here.
  // cmask: one bit per hit slot = "yields an arc" (before the endpoint test), so that passes B and C need not look at dead slots
	uint32_t mx = 0, n_live = 0;
	// (round 5, visit 15: four 256-slot rounds per trip -- the ids of four rounds fetched together, then the four flags -- ran at 1.02 ms against 0.50: left as it was)
	for (size_t base = (size_t)blockIdx.x * 256; base < n; base += (size_t)gridDim.x * 256) {
		const size_t i = base + threadIdx.x;
		int cand = 0;
		if (i < n) {
			mc_arc_t x;
			uint32_t q = 0, t = 0;
			int self_rc = 0, r = sg_candidate(h, i, slen, max_hang, int_frac, min_ovlp, lazy_del, &x, &q, &t, &self_rc);
			if (r != -100) {
				++n_live;
				if (r >= 0) {
					if (q == t) { if (self_rc) sdel[q] = 1; } // asm.c:27-31
					else cand = 1, mx = x.len > mx ? x.len : mx;
				} else if (r == MC_HT_QCONT) sdel[q] = 1; // asm.c:34
			}
		}
		const unsigned long long m = wv_ballot(cand);
		if ((threadIdx.x & 63) == 0) cmask[i >> 6] = m;
	}
	blk_max_u64(&ctr[CT_MAXLEN], mx);
	blk_add_u64(&ctr[CT_LIVE], n_live);
}

// passes B (out.u == nullptr: count per tile) and C (write): one thread per 64-slot word of the candidate mask, tile b = the words
// [b*256, (b+1)*256) = SG_TILE consecutive hits.  Candidates are rare (a handful per tile): a thread walks the set bits of its word
// and recomputes those arcs; threads, and the bits inside a word, are in hit order, so the dense slots are too.
// A word with many candidates (graph-heavy inputs: reads of one length, nothing contained -- every hit becomes an arc) is taken by the whole WAVE, a slot
// per lane: eight coalesced column loads per word instead of eight cache lines per candidate of a lane that walks its word alone (SG_DENSE: from how many
// candidates on).  Which form handles a word changes nothing about where its arcs go: the owner lane's count and offset are the same.
#ifndef SG_DENSE
#define SG_DENSE 6
#endif
__device__ __forceinline__ unsigned long long wv_bcast_u64(unsigned long long x, int src) { return (unsigned long long)__shfl((uint32_t)(x >> 32), src, 64) << 32 | __shfl((uint32_t)x, src, 64); }

__global__ __launch_bounds__(256) void k_sg_emit(HitColsG h, size_t n, const uint32_t *__restrict__ slen, const uint8_t *__restrict__ sdel,
                                                  int max_hang, float int_frac, int min_ovlp, const unsigned long long *cmask,
                                                  uint32_t *__restrict__ tile_cnt, const uint32_t *__restrict__ tile_off, ArcCols out, uint32_t *__restrict__ aslot,
                                                  uint32_t *__restrict__ wpos, unsigned long long *kmask)
{ // aslot (optional, pass C): the hit slot each pushed arc comes from (push order, tie-order repair)
  // wpos / kmask (optional, pass C): per 64-slot word the position of its first arc in the push sequence and the slots that yielded one -- the arc sort finds the
  // stretch of a read's arcs from them and the read's hit slots (arc_pos_of_slot) instead of a sweep over the arcs (k_arc_groups: 0.55 ms per 200 M arcs).
  // cmask and kmask MAY BE THE SAME buffer (mahip_sg_finish: c->sgmask; hence no __restrict__ on either): every thread reads its word before it writes it, and
  // from then on the buffer holds KEPT bits, not candidates -- mahip_sg_finish consumes the candidates exactly once (mahip_sg_flags makes them anew)
	__shared__ uint32_t s_w[4];
	const unsigned lane = threadIdx.x & 63;
	const size_t w = (size_t)blockIdx.x * 256 + threadIdx.x, n_words = (n + 63) >> 6, w0 = w - lane;
	unsigned long long m = w < n_words ? cmask[w] : 0ull, kept = 0;
	uint32_t cnt = 0;
	const bool dense = __popcll(m) >= SG_DENSE;
	const unsigned long long dense_lanes = wv_ballot(dense);
	// which candidates survive asg_arc_rm on the fresh arcs (asg.c:57-70): endpoints must be alive
	for (unsigned long long dl = dense_lanes; dl; dl &= dl - 1) {
		const int l = __ffsll((long long)dl) - 1;
		const unsigned long long ml = wv_bcast_u64(m, l);
		const size_t i = ((w0 + (size_t)l) << 6) + lane;
		const int k = (ml >> lane & 1ull) && i < n && !sdel[h.qid[i]] && !sdel[h.tn[i]];
		const unsigned long long kb = wv_ballot(k);
		if ((int)lane == l) kept = kb, cnt = (uint32_t)__popcll(kb);
	}
	if (!dense)
		for (unsigned long long rest = m; rest; rest &= rest - 1) {
			const int b = __ffsll((long long)rest) - 1;
			const size_t i = (w << 6) + (size_t)b;
			if (i < n && !sdel[h.qid[i]] && !sdel[h.tn[i]]) kept |= 1ull << b, ++cnt;
		}
	uint32_t tot, ex = block_excl_scan_256(cnt, s_w, &tot);
	if (!out.u) { if (threadIdx.x == 0) tile_cnt[blockIdx.x] = tot; return; }
	uint32_t p = tile_off[blockIdx.x] + ex;
	if (wpos && w < n_words) { wpos[w] = p; kmask[w] = kept; }
	for (unsigned long long dl = dense_lanes; dl; dl &= dl - 1) {
		const int l = __ffsll((long long)dl) - 1;
		const unsigned long long kl = wv_bcast_u64(kept, l);
		const uint32_t pl = __shfl(p, l, 64);
		if (kl >> lane & 1ull) {
			const size_t i = ((w0 + (size_t)l) << 6) + lane;
			const uint32_t pp = pl + (uint32_t)__popcll(kl & wv_lt(lane));
			mc_arc_t x;
			uint32_t q = 0, t = 0;
			int self_rc = 0;
			sg_candidate(h, i, slen, max_hang, int_frac, min_ovlp, nullptr, &x, &q, &t, &self_rc);
			out.u[pp] = x.u; out.v[pp] = x.v; out.len[pp] = x.len; out.ol[pp] = x.ol;
			if (aslot) aslot[pp] = (uint32_t)i;
		}
	}
	if (!dense)
		for (; kept; kept &= kept - 1, ++p) {
			const size_t i = (w << 6) + (size_t)(__ffsll((long long)kept) - 1);
			mc_arc_t x;
			uint32_t q = 0, t = 0;
			int self_rc = 0;
			sg_candidate(h, i, slen, max_hang, int_frac, min_ovlp, nullptr, &x, &q, &t, &self_rc);
			out.u[p] = x.u; out.v[p] = x.v; out.len[p] = x.len; out.ol[p] = x.ol;
			if (aslot) aslot[p] = (uint32_t)i;
		}
}

__global__ __launch_bounds__(256) void k_arc_keep(ArcCols a, size_t n, const uint8_t *__restrict__ sdel, uint32_t *__restrict__ keep, int use_keep_in)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	int k = use_keep_in ? (int)keep[i] : 1;
	if (k) k = !(a.ol[i] & ADEL) && !sdel[a.u[i] >> 1] && !sdel[a.v[i] >> 1];
	keep[i] = k;
}

__global__ __launch_bounds__(256) void k_arc_compact(ArcCols in, size_t n, const uint32_t *__restrict__ keep, const uint32_t *__restrict__ pos, ArcCols out)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n && keep[i]) {
		uint32_t p = pos[i];
		out.u[p] = in.u[i]; out.v[p] = in.v[i]; out.len[p] = in.len[i]; out.ol[p] = in.ol[i];
	}
}

// asg_arc_rm (asg.c:57-70) in ONE pass: a tile of RM_TILE consecutive arcs decides what stays (not deleted, both reads alive), learns where its
// survivors go from the tiles before it (chained look-back, mahip_internal.hpp: sc_look_back) and writes them -- every arc read once, the survivors
// written once (32 B per arc), no flag and position arrays in between.  The three-launch form above (flags, scan, compact) moves 70 B per
// arc (round 4, visit A: 1.25 against 3.6 ms at 200 M arcs) and serves the callers that bring a pre-filter of their own (keep_in).
#define RM_ITEMS 8
#define RM_TILE (256 * RM_ITEMS)
// The CLEAN form is THREE launches instead of a chain: a chain's ticket is ONE word that every block increments, and atomics on one address are served one after the
// other -- 12.7 ns each here, which at 98 k tiles WAS the launch (1.24 ms whether three columns were streamed or one: round 5, visit 4); chaining groups of tiles instead
// (one ticket per 32 k arcs) made every block of the launch publish at the same moment and look back over all the others at once (0.83 - 1.0 ms, visits 5 - 7).  So:
// count the survivors of every group of RMC_GROUP arcs from the overlap words, scan the (few thousand) counts, read the words again and write the survivors.
#define RMC_ITEMS 8
#define RMC_TILE (256 * RMC_ITEMS)
#define RMC_TILES 16u
#define RMC_GROUP ((size_t)RMC_TILES * RMC_TILE)
__global__ __launch_bounds__(256) void k_arc_rm_count(const uint32_t *__restrict__ aol, size_t n, uint32_t *__restrict__ cnt)
{
	__shared__ uint32_t s_wave[4];
	const size_t g0 = (size_t)blockIdx.x * RMC_GROUP, g1 = g0 + RMC_GROUP < n ? g0 + RMC_GROUP : n;
	uint32_t mine = 0;
	for (size_t b0 = g0; b0 < g1; b0 += 8 * 1024) { // 4 words per lane and load (coalesced 16-byte loads), 8 independent loads in flight; n's tail one by one
		uint4 w[8];
#pragma unroll
		for (int j = 0; j < 8; ++j) { const size_t b4 = b0 + (size_t)j * 1024 + (size_t)threadIdx.x * 4; w[j] = b4 + 4 <= g1 ? *(const uint4*)(aol + b4) : make_uint4(ADEL, ADEL, ADEL, ADEL); }
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			const size_t b4 = b0 + (size_t)j * 1024 + (size_t)threadIdx.x * 4;
			mine += !(w[j].x & ADEL) + !(w[j].y & ADEL) + !(w[j].z & ADEL) + !(w[j].w & ADEL);
			if (b4 < g1 && b4 + 4 > g1) for (size_t i = b4; i < g1; ++i) mine += !(aol[i] & ADEL);
		}
	}
	uint32_t gtot;
	(void)block_excl_scan_256(mine, s_wave, &gtot);
	if (threadIdx.x == 0) cnt[blockIdx.x] = gtot;
}
__global__ __launch_bounds__(256) void k_arc_rm_write(ArcCols in, size_t n, ArcCols out, const uint32_t *__restrict__ pre)
{
	__shared__ uint32_t s_wave[4];
	const size_t g0 = (size_t)blockIdx.x * RMC_GROUP, g1 = g0 + RMC_GROUP < n ? g0 + RMC_GROUP : n;
	uint32_t p0 = pre[blockIdx.x]; // first output slot of the current tile
	for (size_t tb = g0; tb < g1; tb += RMC_TILE) {
		const size_t base = tb + (size_t)threadIdx.x * RMC_ITEMS;
		uint32_t ol[RMC_ITEMS], keep = 0;
		if (base + RMC_ITEMS <= n) {
			const uint4 *po = (const uint4*)(in.ol + base);
			const uint4 c0 = po[0], c1 = po[1];
			ol[0] = c0.x, ol[1] = c0.y, ol[2] = c0.z, ol[3] = c0.w, ol[4] = c1.x, ol[5] = c1.y, ol[6] = c1.z, ol[7] = c1.w;
		} else {
#pragma unroll
			for (int i = 0; i < RMC_ITEMS; ++i) ol[i] = base + i < n ? in.ol[base + i] : ADEL;
		}
#pragma unroll
		for (int i = 0; i < RMC_ITEMS; ++i) if (!(ol[i] & ADEL)) keep |= 1u << i;
		uint32_t tot;
		const uint32_t ex = block_excl_scan_256((uint32_t)__popc(keep), s_wave, &tot);
		uint32_t p = p0 + ex;
#pragma unroll
		for (int i = 0; i < RMC_ITEMS; ++i)
			if (keep >> i & 1u) { out.u[p] = in.u[base + i]; out.v[p] = in.v[base + i]; out.len[p] = in.len[base + i]; out.ol[p] = ol[i]; ++p; }
		p0 += tot;
	}
}

// (the form for arcs that have to be checked against seq.del: an imported graph, a cleanup behind a cleaner that deleted reads; the clean case is k_arc_rm_count / _write above)
__global__ __launch_bounds__(256) void k_arc_rm_chain(ArcCols in, size_t n, const uint8_t *__restrict__ sdel, ArcCols out, uint32_t *__restrict__ d_total,
                                                       unsigned long long *state, uint32_t *ticket, uint32_t ticket_base, uint32_t epoch)
{
	__shared__ uint32_t s_wave[4];
	__shared__ uint32_t s_tile, s_prefix;
	if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u) - ticket_base;
	__syncthreads();
	const uint32_t tile = s_tile;
	const size_t base = (size_t)tile * RM_TILE + (size_t)threadIdx.x * RM_ITEMS;
	uint32_t u[RM_ITEMS], v[RM_ITEMS], ol[RM_ITEMS], keep = 0;
	if (base + RM_ITEMS <= n) {
		const uint4 *pu = (const uint4*)(in.u + base), *pv = (const uint4*)(in.v + base), *po = (const uint4*)(in.ol + base);
		const uint4 a0 = pu[0], a1 = pu[1], b0 = pv[0], b1 = pv[1], c0 = po[0], c1 = po[1];
		u[0] = a0.x, u[1] = a0.y, u[2] = a0.z, u[3] = a0.w, u[4] = a1.x, u[5] = a1.y, u[6] = a1.z, u[7] = a1.w;
		v[0] = b0.x, v[1] = b0.y, v[2] = b0.z, v[3] = b0.w, v[4] = b1.x, v[5] = b1.y, v[6] = b1.z, v[7] = b1.w;
		ol[0] = c0.x, ol[1] = c0.y, ol[2] = c0.z, ol[3] = c0.w, ol[4] = c1.x, ol[5] = c1.y, ol[6] = c1.z, ol[7] = c1.w;
	} else {
#pragma unroll
		for (int i = 0; i < RM_ITEMS; ++i) { const bool in_r = base + i < n; u[i] = in_r ? in.u[base + i] : 0; v[i] = in_r ? in.v[base + i] : 0; ol[i] = in_r ? in.ol[base + i] : ADEL; }
	}
#pragma unroll
	for (int i = 0; i < RM_ITEMS; ++i)
		if (base + i < n && !(ol[i] & ADEL) && !sdel[u[i] >> 1] && !sdel[v[i] >> 1]) keep |= 1u << i;
	const uint32_t cnt = (uint32_t)__popc(keep);
	uint32_t tot;
	const uint32_t ex = block_excl_scan_256(cnt, s_wave, &tot);
	if (threadIdx.x == 0) {
		SC_PUBLISH(&state[tile], sc_pack(epoch, tile == 0 ? SC_INCL : SC_AGG, tot));
		if (tile == 0) s_prefix = 0;
	}
	if (tile > 0 && threadIdx.x < 64) {
		const uint32_t prefix = sc_look_back(state, tile, epoch, threadIdx.x);
		if (threadIdx.x == 0) { s_prefix = prefix; SC_PUBLISH(&state[tile], sc_pack(epoch, SC_INCL, prefix + tot)); }
	}
	__syncthreads();
	uint32_t p = s_prefix + ex;
#pragma unroll
	for (int i = 0; i < RM_ITEMS; ++i)
		if (keep >> i & 1u) { out.u[p] = u[i]; out.v[p] = v[i]; out.len[p] = in.len[base + i]; out.ol[p] = ol[i]; ++p; }
	if (base < n && base + RM_ITEMS >= n) *d_total = p; // the last thread with arcs: its end is the total
}

__global__ __launch_bounds__(256) void k_arc_keys(ArcCols a, size_t n, uint64_t *__restrict__ key, uint32_t *__restrict__ val)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) { key[i] = (uint64_t)a.u[i] << 32 | a.len[i]; val[i] = (uint32_t)i; }
}

__global__ __launch_bounds__(256) void k_arc_permute(ArcCols in, size_t n, const uint32_t *__restrict__ perm, ArcCols out)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) {
		uint32_t j = perm[i];
		out.u[i] = in.u[j]; out.v[i] = in.v[j]; out.len[i] = in.len[j]; out.ol[i] = in.ol[j];
	}
}

// ---- tie order of asg_arc_sort (asg.c:22-25) ----
// census over the stably sorted keys: groups of >= 2 equal (u,len) keys and their members
__global__ __launch_bounds__(256) void k_arc_tie_census(const uint64_t *__restrict__ skey, size_t n, unsigned long long *__restrict__ ctr)
{
	uint32_t groups = 0, members = 0;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
		const uint64_t k = skey[i];
		const int eq_prev = i > 0 && skey[i - 1] == k, eq_next = i + 1 < n && skey[i + 1] == k;
		groups += eq_next && !eq_prev;
		members += eq_prev || eq_next;
	}
	blk_add_u64(&ctr[ST_ARC_TIE_GROUPS], groups);
	blk_add_u64(&ctr[ST_ARC_TIE_ARCS], members);
}
// the key ma_hit_sort ordered the hit of a pushed arc by: the ORIGINAL qns of the slot's record
__global__ __launch_bounds__(256) void k_arc_slot_keys(const uint32_t *__restrict__ aslot, size_t n, const ma_hit_t *__restrict__ h, const uint32_t *__restrict__ sidx,
                                                        uint64_t *__restrict__ key, uint32_t *__restrict__ val)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) { key[i] = h[sidx[aslot[i]]].qns; val[i] = (uint32_t)i; }
}
// consecutive pushed arcs (stable push order) whose hits had the same original (qid,qs) key: only then does the order of tied hits reach the arcs
__global__ __launch_bounds__(256) void k_arc_push_conflicts(const uint64_t *__restrict__ skey, size_t n, unsigned long long *__restrict__ ctr)
{
	uint32_t cnt = 0;
	for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x + 1; p < n; p += (size_t)gridDim.x * 256) cnt += skey[p] == skey[p - 1];
	blk_add_u64(&ctr[ST_PUSH_CONFLICTS], cnt);
}
// ---- which push conflicts can the reference's arc sort SEE?  (round 6)
// A conflict = two arcs x, y, neighbours in the stable push order, from hits with equal (qid,qs): only the reference's (unstable) hit sort knows which of them ma_sg_gen
// pushes first.  asg_arc_sort is an MSD radix sort (ksort.h:149-183): as long as x and y carry the same digit, a level's walk only tells elements apart by their digit,
// so pushing y before x lands every OTHER element where it landed before and x, y in each other's places.  At the level of their first differing byte they sit in one
// bucket B (the arcs that share the key bytes above it); there the walk may take another course -- for the elements of B only, and what it leaves open is only the order
// inside groups of EQUAL keys.  So the swap can change the result only if B holds a tie group.  If B has at most RS_MIN_SIZE = 64 elements, it -- or a bucket above it
// that is as small -- is not walked but insertion-sorted (ksort.h:182), stably: equal keys keep the order they came in.  x and y have come in in each other's places
// (not as neighbours any more: the walks above do not keep the order of their input), so what the swap can change there is the order of x, or of y, against ANOTHER arc
// with the same key that came in between the two: visible if x or y belongs to a tie group, invisible otherwise.  (Until the last day of round 6 the rule read "a small
// bucket never shows a swap of two different keys" -- which forgot that third arc.)  x and y
// with equal keys are a tie group themselves.  A conflict that fails the test is invisible; if all of them are, the stable push order gives the reference's graph and
// the walk over the hit keys (seconds at BASELINE configs[4], 0.48 of the 0.94 s of the 50 M-line realistic input) is not needed.  Runs of more than two tied hits: the
// bucket of any two members lies inside the larger of the buckets of the neighbouring pairs between them (common prefixes are an ultrametric), so neighbours suffice.
// S = the arcs' keys with squeezed ids, ascending (the stable sort's output); tp[i] = pairs j < i with S[j] == S[j+1].
__global__ __launch_bounds__(256) void k_tie_pair_flags(const uint64_t *__restrict__ S, size_t n, uint32_t *__restrict__ f)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) f[i] = i + 1 < n && S[i] == S[i + 1];
}
__device__ __forceinline__ uint32_t lower_bound_u64(const uint64_t *__restrict__ S, uint32_t n, uint64_t x)
{ // first position whose key is >= x
	uint32_t lo = 0, hi = n;
	while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (S[mid] < x) lo = mid + 1; else hi = mid; }
	return lo;
}
__global__ __launch_bounds__(256) void k_arc_push_conflicts_seen(const uint64_t *__restrict__ skey, const uint32_t *__restrict__ perm, ArcCols a /* push sequence */, const int32_t *__restrict__ map,
                                                                  const uint64_t *__restrict__ S, const uint32_t *__restrict__ tp, uint32_t n, unsigned long long *__restrict__ ctr,
                                                                  uint32_t *__restrict__ want)
{ // want: a bit per read id (the ids of the hit keys), set for the reads that have a conflict in sight -- the hit walk's order is taken inside their hit groups only
	uint32_t cnt = 0;
	for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x + 1; p < n; p += (size_t)gridDim.x * 256) {
		if (skey[p] != skey[p - 1]) continue;
		const uint32_t ia = perm[p - 1], ib = perm[p];
		uint32_t ua = a.u[ia], ub = a.u[ib];
		if (map) { ua = (uint32_t)map[ua >> 1] << 1 | (ua & 1); ub = (uint32_t)map[ub >> 1] << 1 | (ub & 1); }
		const uint64_t ka = (uint64_t)ua << 32 | a.len[ia], kb = (uint64_t)ub << 32 | a.len[ib];
		const uint32_t rd = (uint32_t)(skey[p] >> 32);
		if (ka == kb) { ++cnt; atomicOr(&want[rd >> 5], 1u << (rd & 31)); continue; }
		const int sh = ((63 - __clzll((long long)(ka ^ kb))) / 8 + 1) * 8; // bits below the shared prefix
		uint32_t r0 = 0, r1 = n;
		if (sh < 64) {
			const uint64_t lo = ka >> sh << sh, hi = lo + (1ull << sh);
			r0 = lower_bound_u64(S, n, lo);
			r1 = hi > lo ? lower_bound_u64(S, n, hi) : n; // (the top bucket ends with the array)
		}
		bool seen;
		if (r1 - r0 > 64u) seen = tp[r1 - 1] > tp[r0]; // a walked bucket with a tie group in it
		else { // insertion-sorted: is x or y one of several arcs with its key?
			const uint32_t pa = lower_bound_u64(S, n, ka), pb = lower_bound_u64(S, n, kb);
			seen = (pa + 1 < n && S[pa + 1] == ka) || (pb + 1 < n && S[pb + 1] == kb);
		}
		if (seen) { ++cnt; atomicOr(&want[rd >> 5], 1u << (rd & 31)); }
	}
	blk_add_u64(&ctr[ST_PUSH_SEEN], cnt);
}
// sort key of a pushed arc in the reference's hit order
__global__ __launch_bounds__(256) void k_arc_push_keys(const uint32_t *__restrict__ aslot, const uint32_t *__restrict__ hrank, size_t n, uint64_t *__restrict__ key, uint32_t *__restrict__ val)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) { key[i] = hrank[aslot[i]]; val[i] = (uint32_t)i; }
}
// the keys the reference sorts: ul = u<<32 | len with the SQUEEZED read ids (ma_sg_gen runs after ma_hit_contained renumbered the reads)
__global__ __launch_bounds__(256) void k_arc_keys_ref(ArcCols a, size_t n, const int32_t *__restrict__ map, uint64_t *__restrict__ key)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) {
		uint32_t u = a.u[i];
		if (map) u = (uint32_t)map[u >> 1] << 1 | (u & 1);
		key[i] = (uint64_t)u << 32 | a.len[i];
	}
}

// asg.c:27-36 asg_arc_index_core on a zeroed idx: idx[u] = first<<32 | count.  The first arc of a run contributes first<<32 - first, the last one
// its end: the sum is first<<32 | (end - first) (the borrow of the first term is paid back by the carry of the second).  Round 3 let the first arc's
// thread count its run in a loop -- one dependent load per arc of the run, 2.2 ms per launch at 200 M arcs; this form is one streaming read of the u column.
__global__ __launch_bounds__(256) void k_arc_index(const uint32_t *__restrict__ au, size_t n, unsigned long long *__restrict__ idx)
{
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
		const uint32_t u = au[i];
		const bool first = i == 0 || au[i - 1] != u, last = i + 1 == n || au[i + 1] != u;
		if (first && last) idx[u] = (unsigned long long)i << 32 | 1ull; // a run of one: nobody else writes the word
		else if (first) atomicAdd(&idx[u], ((unsigned long long)i << 32) - (unsigned long long)i);
		else if (last) atomicAdd(&idx[u], (unsigned long long)(i + 1));
	}
}

// ---- asg_arc_sort + asg_arc_index (asg.c:22-42) for arcs that arrive GROUPED BY READ ----
// ma_sg_gen pushes the arcs in hit order and the hit slots are grouped by query id, so the arcs of read q -- the lists of its two vertices 2q and 2q+1
// -- are one contiguous stretch of the push sequence.  Sorting by (u, len) is then a sort INSIDE each stretch: one wave per read, the (strand, len,
// position in the stretch) keys in registers, the network of the coverage sweeps (wave_sort_regs), the rows fetched through the sorted positions.
// One read and one write of every arc ("arc sort 32 B per arc") instead of keys + four 12-byte radix passes + a permutation, and the
// wave that has a read's sorted keys in registers also writes the two CSR words (asg_arc_index) and counts the equal (u,len) keys (the tie census).
// The position rides in the low bits, so the order is total and equal keys stay in push order -- the stable order radix.hip gave.
#define AG_IB 9        // bits of an arc's position inside its read's stretch
#define AG_MAX 512u    // arcs per read the register network takes; longer stretches (or lengths of more than 21 bits) send the whole sort to the radix path

// position in the push sequence of the first arc that comes from hit slot s or a later one (k_sg_emit, pass C, left wpos / kmask behind); s = n_slots: all of them
__device__ __forceinline__ uint32_t arc_pos_of_slot(const uint32_t *__restrict__ wpos, const unsigned long long *__restrict__ kmask, uint32_t s, uint32_t n_slots, uint32_t n_arc)
{
	if (s >= n_slots) return n_arc;
	return wpos[s >> 6] + (uint32_t)__popcll(kmask[s >> 6] & ((1ull << (s & 63u)) - 1ull));
}

struct ArcPre { uint32_t u[2], len[2], v[2], ol[2]; }; // the four columns of a read's (up to 128) arcs, arc r * 64 + lane in slot r

template <int ITEMS>
__device__ __forceinline__ void arc_group_sort_regs(const ArcCols &in, const ArcCols &out, uint32_t q, uint32_t beg, uint32_t n, int bl, unsigned lane,
                                                    unsigned long long *__restrict__ idx, uint32_t &tie_groups, uint32_t &tie_arcs, uint32_t &foreign, const ArcPre *pre)
{ // pre (SMALL tier): the read's rows, fetched a read ahead; else loaded here
	uint32_t x[ITEMS];
#pragma unroll
	for (int r = 0; r < ITEMS; ++r) { // any arrangement will do on the way in: the position is part of the key
		const uint32_t i = (uint32_t)r * 64u + lane;
		x[r] = 0xffffffffu;
		if (i < n) {
			const uint32_t u = pre ? pre->u[r < 2 ? r : 0] : in.u[beg + i];
			foreign += (u >> 1) != q; // the stretch goff / wpos / kmask name must hold read q's arcs and nothing else: true when the hit slots ascend by query id (hits sorted here);
			                          // hits indexed as the caller had them (mahip_hits_index: any order of the groups) can break it -> the caller falls back to the general sort
			x[r] = (u & 1u) << (bl + AG_IB) | (pre ? pre->len[r < 2 ? r : 0] : in.len[beg + i]) << AG_IB | i;
		}
	}
	wave_sort_regs<ITEMS>(x, lane); // sorted element p sits in lane p / ITEMS, register p % ITEMS
	const uint32_t lmask = (1u << bl) - 1u;
	uint32_t n0 = 0;
	const uint32_t up = __shfl_up(x[ITEMS - 1], 1, 64), down = __shfl_down(x[0], 1, 64); // the neighbours across the lane borders
#pragma unroll
	for (int r = 0; r < ITEMS; ++r) {
		const uint32_t p = lane * ITEMS + (uint32_t)r;
		const uint32_t k = x[r], at = k & ((1u << AG_IB) - 1u), strand = k >> (bl + AG_IB) & 1u;
		uint32_t v, ol;
		if (pre) { // the row's other two columns sit in lane (at & 63), slot (at >> 6) of the prefetched registers: through the LDS crossbar, no second trip to memory
			const uint32_t v0 = __shfl(pre->v[0], (int)(at & 63u), 64), o0 = __shfl(pre->ol[0], (int)(at & 63u), 64);
			v = v0; ol = o0;
			if (ITEMS > 1) { const uint32_t v1 = __shfl(pre->v[1], (int)(at & 63u), 64), o1 = __shfl(pre->ol[1], (int)(at & 63u), 64); if (at >> 6) v = v1, ol = o1; }
		}
		if (p < n) {
			if (!pre) { v = in.v[beg + at]; ol = in.ol[beg + at]; }
			out.u[beg + p] = q << 1 | strand; out.v[beg + p] = v; out.len[beg + p] = (k >> AG_IB) & lmask; out.ol[beg + p] = ol;
			n0 += strand ^ 1u;
			// tie census: runs of equal (u,len) among neighbours in sorted order
			const uint32_t kp = r > 0 ? x[r > 0 ? r - 1 : 0] : up, kn = r + 1 < ITEMS ? x[r + 1 < ITEMS ? r + 1 : 0] : down;
			const bool eq_prev = p > 0 && (kp >> AG_IB) == (k >> AG_IB), eq_next = p + 1 < n && (kn >> AG_IB) == (k >> AG_IB);
			tie_groups += eq_next && !eq_prev;
			tie_arcs += eq_prev || eq_next;
		}
	}
	n0 = wv_sum_u32(n0);
	if (lane == 0) { // asg_arc_index: vertices without arcs keep the zero of the cleared array (asg.c:29)
		if (n0) idx[2 * (size_t)q] = (unsigned long long)beg << 32 | n0;
		if (n - n0) idx[2 * (size_t)q + 1] = (unsigned long long)(beg + n0) << 32 | (n - n0);
	}
}

__device__ __forceinline__ void arc_prefetch(const ArcCols &in, uint32_t beg, uint32_t n, unsigned lane, ArcPre &p)
{
#pragma unroll
	for (int r = 0; r < 2; ++r) {
		const uint32_t i = (uint32_t)r * 64u + lane;
		p.u[r] = p.len[r] = p.v[r] = p.ol[r] = 0;
		if (i < n) { p.u[r] = in.u[beg + i]; p.len[r] = in.len[beg + i]; p.v[r] = in.v[beg + i]; p.ol[r] = in.ol[beg + i]; }
	}
}

// one wave per read with arcs; a wave takes 64 consecutive reads at a time (their bounds: one coalesced load) and visits the ones that have arcs --
// after containment most reads have none.  SMALL: stretches of <= 128 arcs (2 rows per lane: most of them).  A read is a chain load -> sort -> rows through
// the sorted positions -> store; the rows of the NEXT read are in flight while the current one is sorted, and the rows follow the sorted positions through the
// LDS crossbar instead of a second, dependent trip to memory (round 4: the key columns alone a read ahead changed nothing, 2.5 ms per 200 M arcs either way).
// !SMALL: 129 .. AG_MAX arcs, loaded where they are needed.
template <bool SMALL>
__global__ __launch_bounds__(256) void k_arc_group_sort(ArcCols in, ArcCols out, const uint32_t *__restrict__ goff, const uint32_t *__restrict__ wpos, const unsigned long long *__restrict__ kmask,
                                                         uint32_t n_slots, uint32_t n_arc, uint32_t q_lo, uint32_t q_hi, int bl,
                                                         unsigned long long *__restrict__ idx, unsigned long long *__restrict__ ctr)
{
	const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t tg = 0, ta = 0, big = 0, foreign = 0;
	for (uint64_t qb = (uint64_t)q_lo + (uint64_t)(blockIdx.x * 4 + wave) * 64; qb < q_hi; qb += (uint64_t)gridDim.x * 256) {
		// the arcs of read q come from its hit slots goff[q] .. goff[q+1]: lane l holds the start of read qb + l; its end is the next read's start
		const bool have = qb + lane < q_hi;
		const uint32_t a0 = have ? arc_pos_of_slot(wpos, kmask, goff[qb + lane], n_slots, n_arc) : 0u;
		uint32_t a1 = __shfl_down(a0, 1, 64);
		const uint32_t last = (uint32_t)((qb + 64 < q_hi ? qb + 64 : q_hi) - qb) - 1u; // the chunk's last read: its end is not in a neighbour lane
		if (lane == last) a1 = arc_pos_of_slot(wpos, kmask, goff[qb + lane + 1], n_slots, n_arc);
		const uint2 g = have ? make_uint2(a0, a1) : make_uint2(0, 0);
		const uint32_t n_l = g.y - g.x;
		if (SMALL) big += n_l > AG_MAX; // counted once (by the SMALL launch)
		unsigned long long todo = wv_ballot(SMALL ? (n_l != 0 && n_l <= 128u) : (n_l > 128u && n_l <= AG_MAX));
		if (SMALL) {
			ArcPre cur, nxt;
			int b = todo ? __ffsll((long long)todo) - 1 : -1;
			uint32_t beg = 0, n = 0;
			if (b >= 0) { beg = __shfl(g.x, b, 64); n = __shfl(n_l, b, 64); arc_prefetch(in, beg, n, lane, cur); }
			while (b >= 0) {
				todo &= todo - 1;
				const int bn = todo ? __ffsll((long long)todo) - 1 : -1;
				uint32_t begn = 0, nn = 0;
				if (bn >= 0) { begn = __shfl(g.x, bn, 64); nn = __shfl(n_l, bn, 64); arc_prefetch(in, begn, nn, lane, nxt); } // first: in flight while this read is sorted
				const uint32_t q = (uint32_t)qb + (uint32_t)b;
				if (n <= 64) arc_group_sort_regs<1>(in, out, q, beg, n, bl, lane, idx, tg, ta, foreign, &cur);
				else arc_group_sort_regs<2>(in, out, q, beg, n, bl, lane, idx, tg, ta, foreign, &cur);
				b = bn; beg = begn; n = nn; cur = nxt;
			}
		} else
		while (todo) {
			const int b = __ffsll((long long)todo) - 1;
			todo &= todo - 1;
			const uint32_t q = (uint32_t)qb + (uint32_t)b, beg = __shfl(g.x, b, 64), n = __shfl(n_l, b, 64);
			if (n <= 256) arc_group_sort_regs<4>(in, out, q, beg, n, bl, lane, idx, tg, ta, foreign, nullptr);
			else arc_group_sort_regs<8>(in, out, q, beg, n, bl, lane, idx, tg, ta, foreign, nullptr);
		}
	}
	blk_add_u64(&ctr[ST_ARC_TIE_GROUPS], tg);
	blk_add_u64(&ctr[ST_ARC_TIE_ARCS], ta);
	blk_add_u64(&ctr[CT_OVF2], (SMALL ? big : 0u) + foreign); // either: the general sort takes over (mahip_sg_finish)
}

// ------------------------------------------------------------------------------------------------ asg_arc_del_trans
#ifndef TR_PRE
#define TR_PRE 128u // neighbours whose CSR words are fetched with the vertex's own list (SMALL); the rest wait until they are expanded.  128 = all of them
#endif
#define TR_CAP 512
#define TR_HASH 1024
#define TR_EMPTY 0xffffffffu

__device__ __forceinline__ uint32_t tr_hash(uint32_t x, uint32_t hbits) { return (x * 0x9E3779B1u) >> (32 - hbits); }
__device__ __forceinline__ int tr_find(const uint32_t *hk, uint32_t x, uint32_t hbits)
{
	uint32_t mask = (1u << hbits) - 1, s = tr_hash(x, hbits);
	for (;;) {
		uint32_t k = hk[s];
		if (k == x) return (int)s;
		if (k == TR_EMPTY) return -1;
		s = (s + 1) & mask;
	}
}

// One wave per vertex.  Two instantiations split the vertices by out-degree so that the common case (<= 128 arcs)
// runs with a small LDS footprint at full occupancy: SMALL handles 1..128 arcs and also prefetches the CSR entries of
// all neighbours (one coalesced gather instead of one dependent load per expansion); !SMALL handles 129..TR_CAP arcs
// and sends larger vertices to the block-per-vertex tier.
//
// The reference walks v's arcs in order and expands neighbour i only if mark[target_i] is still 1 (asg.c:168).  Marks
// only ever go 1 -> 2, so "the next arc to expand" is simply the lowest not-yet-passed arc whose target is still
// marked 1 NOW: each lane watches the mark of its own arc and a ballot finds that arc -- the serial walk shrinks from
// one step per arc to one step per expansion (about one per vertex on clean data).
struct TrPre { uint32_t v[2], l[2], o[2]; uint32_t dead; }; // a vertex's (up to 128) arcs -- target, length, overlap word -- and its read's seq.del, a vertex ahead

template <int CAP, int HASH, bool SMALL>
__global__ __launch_bounds__(256) void k_asg_trans(const uint32_t *__restrict__ av, const uint32_t *__restrict__ alen, uint32_t *__restrict__ aol,
                                                    const unsigned long long *__restrict__ idx, const uint8_t *__restrict__ sdel, uint32_t v_beg, uint32_t n_vtx,
                                                    uint32_t fuzz, uint32_t *__restrict__ ovf, unsigned long long *__restrict__ ctr)
{ // processes the vertices [v_beg, n_vtx): the whole graph on one GPU, a rank's own read range in the sharded mode
	__shared__ uint32_t s_v[4][CAP], s_l[4][CAP], s_slot[4][CAP], s_hk[4][HASH], s_hm[4][HASH];
	__shared__ uint32_t s_ws[4][SMALL ? CAP : 1], s_nw[4][SMALL ? CAP : 1];
	const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t *lv = s_v[wave], *ll = s_l[wave], *slot = s_slot[wave], *hk = s_hk[wave], *hm = s_hm[wave];
	uint32_t n_red = 0, n_inner = 0; // n_inner: bodies of the loop at asg.c:169 this lane executed (SURVEY 8(d) prices the reduction at 16 (A + I) bytes)
	// a wave takes 64 consecutive vertices at a time: one coalesced load of their CSR entries, then only the vertices that have arcs (of this
	// instantiation's size class) are visited -- after containment most vertices have none, and a dependent load per vertex is pure latency.
	// A vertex is a chain of dependent trips to memory (own list -> the neighbours' CSR words -> a neighbour's list -> the overlap words to flag): at 200 M
	// arcs, where every vertex has work, the chain is what a launch costs (round 4, visit A: 3.8 ms, 0.21 of the roofline).  SMALL therefore fetches the NEXT
	// vertex's rows (with the overlap words, so that a flag is a plain store) and its read's seq.del while the current vertex is reduced, and a neighbour's
	// targets travel with its lengths: two trips per vertex are left.
	for (uint64_t vb = (uint64_t)v_beg + (uint64_t)(blockIdx.x * 4 + wave) * 64; vb < n_vtx; vb += (uint64_t)gridDim.x * 256) {
	const unsigned long long xl = vb + lane < n_vtx ? idx[vb + lane] : 0ull;
	const uint32_t nvl = (uint32_t)xl;
	unsigned long long todo = wv_ballot(nvl != 0 && !(SMALL ? nvl > (uint32_t)CAP : nvl <= 128u));
	TrPre cur, nxt;
	int vbit = todo ? __ffsll((long long)todo) - 1 : -1;
	auto preload = [&](int bit, TrPre &p) {
		const unsigned long long x = __shfl(xl, bit, 64);
		const uint32_t st = (uint32_t)(x >> 32), nv = (uint32_t)x;
#pragma unroll
		for (int r = 0; r < 2; ++r) {
			const uint32_t i = (uint32_t)r * 64u + lane;
			p.v[r] = p.l[r] = p.o[r] = 0;
			if (i < nv) { p.v[r] = av[st + i]; p.l[r] = alen[st + i]; p.o[r] = aol[st + i]; }
		}
		p.dead = sdel[((uint32_t)vb + (uint32_t)bit) >> 1];
	};
	if (SMALL && vbit >= 0) preload(vbit, cur);
	while (vbit >= 0) {
		todo &= todo - 1;
		const int vnext = todo ? __ffsll((long long)todo) - 1 : -1;
		if (SMALL && vnext >= 0) preload(vnext, nxt);
		const uint32_t v = (uint32_t)vb + (uint32_t)vbit;
		const unsigned long long x = __shfl(xl, vbit, 64);
		uint32_t st = (uint32_t)(x >> 32), nv = (uint32_t)x;
		if (SMALL ? cur.dead != 0 : sdel[v >> 1] != 0) { // asg.c:158-161
			if (SMALL) {
#pragma unroll
				for (int r = 0; r < 2; ++r) { const uint32_t i = (uint32_t)r * 64u + lane; if (i < nv) aol[st + i] = cur.o[r] | ADEL, ++n_red; }
			} else for (uint32_t i = lane; i < nv; i += 64) aol[st + i] |= ADEL, ++n_red;
			vbit = vnext; cur = nxt;
			continue;
		}
		if (nv > (uint32_t)CAP) { if (lane == 0) { unsigned long long k = atomicAdd(&ctr[CT_OVF2], 1ull); ovf[k] = v; } vbit = vnext; cur = nxt; continue; }
		uint32_t hbits = 6; while ((1u << hbits) < 2 * nv) ++hbits;
		uint32_t hsize = 1u << hbits, hmask = hsize - 1;
		if (SMALL) {
#pragma unroll
			for (int r = 0; r < 2; ++r) {
				const uint32_t i = (uint32_t)r * 64u + lane;
				if (i < nv) {
					const uint32_t w = cur.v[r];
					lv[i] = w, ll[i] = cur.l[r];
					if (i < TR_PRE) { unsigned long long xw = idx[w]; s_ws[wave][i] = (uint32_t)(xw >> 32); s_nw[wave][i] = (uint32_t)xw; }
				}
			}
		} else
		for (uint32_t i = lane; i < nv; i += 64) {
			uint32_t w = av[st + i];
			lv[i] = w, ll[i] = alen[st + i];
		}
		// hm[slot] = (index of the FIRST arc to this target) << 2 | mark
		for (uint32_t s = lane; s < hsize; s += 64) hk[s] = TR_EMPTY, hm[s] = 0xffffffffu;
		wv_sync();
		for (uint32_t i = lane; i < nv; i += 64) { // mark all neighbours 1 (asg.c:162); duplicates share a slot
			uint32_t key = lv[i], s = tr_hash(key, hbits);
			for (;;) {
				uint32_t old = atomicCAS(&hk[s], TR_EMPTY, key);
				if (old == TR_EMPTY || old == key) { atomicMin(&hm[s], i << 2 | 1u); slot[i] = s; break; }
				s = (s + 1) & hmask;
			}
		}
		wv_sync();
		const uint32_t L = ll[nv - 1] + fuzz; // asg.c:163
		for (uint32_t base = 0; base < nv; base += 64) {
			const uint32_t i = base + lane;
			const uint32_t myslot = i < nv ? slot[i] : 0;
			uint64_t passed = 0; // lanes of this chunk the walk has gone past
			for (;;) {
				uint64_t cand = wv_ballot(i < nv && (hm[myslot] & 3u) == 1u) & ~passed;
				if (!cand) break;
				const unsigned l0 = (unsigned)(__ffsll((long long)cand) - 1);
				passed |= l0 == 63 ? ~0ull : ((2ull << l0) - 1ull);
				const uint32_t i0 = base + l0, li = ll[i0];
				uint32_t ws, nw;
				if (SMALL && i0 < TR_PRE) ws = s_ws[wave][i0], nw = s_nw[wave][i0];
				else { unsigned long long xw = idx[lv[i0]]; ws = (uint32_t)(xw >> 32); nw = (uint32_t)xw; }
				for (uint32_t j0 = 0; j0 < nw; j0 += 64) { // lanes over w's arcs; sorted by len => the loop of asg.c:169 is a prefix
					uint32_t j = j0 + lane;
					int ok = j < nw;
					const uint32_t lx = ok ? alen[ws + j] : 0, y = ok ? av[ws + j] : 0; // both columns at once: the target is wanted by (almost) every lane that has a length
					int cond = ok && lx + li <= L;
					uint64_t fail = wv_ballot(ok && !cond);
					if (fail) cond = cond && lane < (unsigned)(__ffsll((long long)fail) - 1);
					if (cond) {
						++n_inner;
						int sx = tr_find(hk, y, hbits);
						if (sx >= 0) hm[sx] = (hm[sx] & ~3u) | 2u; // every writer stores the same word
					}
					if (fail) break;
				}
				wv_sync();
			}
		}
		// asg.c:181-184: the sweep resets mark[target] at the first arc to a target, so of several arcs to one
		// reduced target (multi-arcs are still present here) only the first is deleted
		if (SMALL) {
#pragma unroll
			for (int r = 0; r < 2; ++r) { const uint32_t i = (uint32_t)r * 64u + lane; if (i < nv && hm[slot[i]] == (i << 2 | 2u)) aol[st + i] = cur.o[r] | ADEL, ++n_red; }
		} else
		for (uint32_t i = lane; i < nv; i += 64)
			if (hm[slot[i]] == (i << 2 | 2u)) aol[st + i] |= ADEL, ++n_red;
		wv_sync();
		vbit = vnext; cur = nxt;
	}
	}
	blk_add_u64(&ctr[CT_NRED], n_red);
	blk_add_u64(&ctr[CT_TRINNER], n_inner);
}

// ---- the first tier (1 .. 128 arcs) as a four-stage software pipeline (round 5) ----
// A vertex is a chain of dependent trips to memory: its own rows -> its neighbours' CSR words -> the list of the first neighbour (which asg.c:168 ALWAYS expands:
// nothing has been marked 2 yet) -> the lists of the neighbours that are still marked 1 after that.  Round 4 fetched the rows a vertex ahead and paid the other trips per
// vertex: at 200 M arcs, where every vertex has work, a wave spent 6 us per vertex, nearly all of it waiting (2.95 ms per launch, 0.27 of the roofline).  Here every trip but
// the last is issued for a LATER vertex of the wave's chunk: stage 1 loads the rows of vertex t+3, stage 2 the CSR words of the neighbours of t+2, stage 3 the first
// neighbour's list of t+1, while vertex t is reduced from registers and LDS.  What is left on the critical path are the expansions after the first one, and those are
// batched: the lists of up to FOUR pending candidates are fetched at once, 16 entries each (a quarter of the wave per candidate; the far neighbours that survive the first
// expansion have short prefixes inside L), and then replayed in the reference's order -- a candidate that an earlier one of its batch marked is skipped, as asg.c:168 would.
// The overlap words are not carried through the pipeline: they are fetched when a vertex is taken up and used when its flags are written.
struct TrItem { uint32_t st, nv, nvp; uint32_t v[2], l[2], dead; uint32_t ws[2], nw[2]; uint32_t el[2], ev[2], n0; }; // nvp: arcs this kernel will expand from (0: dead read, no vertex)

__global__ __launch_bounds__(256) void k_asg_trans_pipe(const uint32_t *__restrict__ av, const uint32_t *__restrict__ alen, uint32_t *__restrict__ aol,
                                                         const unsigned long long *__restrict__ idx, const uint8_t *__restrict__ sdel, uint32_t v_beg, uint32_t n_vtx,
                                                         uint32_t fuzz, unsigned long long *__restrict__ ctr)
{
	constexpr int CAP = 128, HASH = 256;
	__shared__ uint32_t s_l[4][CAP], s_slot[4][CAP], s_hk[4][HASH], s_hm[4][HASH], s_ws[4][CAP], s_nw[4][CAP];
	const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t *ll = s_l[wave], *slot = s_slot[wave], *hk = s_hk[wave], *hm = s_hm[wave], *lws = s_ws[wave], *lnw = s_nw[wave];
	uint32_t n_red = 0, n_inner = 0;
	for (uint64_t vb = (uint64_t)v_beg + (uint64_t)(blockIdx.x * 4 + wave) * 64; vb < n_vtx; vb += (uint64_t)gridDim.x * 256) {
		const unsigned long long xl = vb + lane < n_vtx ? idx[vb + lane] : 0ull;
		const uint32_t nvl = (uint32_t)xl;
		unsigned long long todo = wv_ballot(nvl != 0 && nvl <= (uint32_t)CAP);
		// the stages; an item without a vertex (bit < 0) has nv = 0 and issues nothing -- no branch around the loads, so that the waits stay counted
		auto stage1 = [&](int bit, TrItem &p) {
			const unsigned long long x = __shfl(xl, bit < 0 ? 0 : bit, 64);
			p.st = (uint32_t)(x >> 32); p.nv = bit < 0 ? 0u : (uint32_t)x;
#pragma unroll
			for (int r = 0; r < 2; ++r) {
				const uint32_t i = (uint32_t)r * 64u + lane;
				p.v[r] = p.l[r] = 0;
				if (i < p.nv) { p.v[r] = av[p.st + i]; p.l[r] = alen[p.st + i]; }
			}
			p.dead = bit < 0 ? 0u : sdel[((uint32_t)vb + (uint32_t)bit) >> 1];
		};
		auto stage2 = [&](TrItem &p) { // the rows have arrived: the CSR words of the neighbours
			p.nvp = p.dead ? 0u : p.nv;
#pragma unroll
			for (int r = 0; r < 2; ++r) {
				const uint32_t i = (uint32_t)r * 64u + lane;
				p.ws[r] = p.nw[r] = 0;
				if (i < p.nvp) { const unsigned long long xw = idx[p.v[r]]; p.ws[r] = (uint32_t)(xw >> 32); p.nw[r] = (uint32_t)xw; }
			}
		};
		auto stage3 = [&](TrItem &p) { // the CSR words have arrived: the list of the first neighbour (lane 0, row 0), up to 128 entries of it
			const uint32_t ws0 = __shfl(p.ws[0], 0, 64);
			p.n0 = p.nvp ? __shfl(p.nw[0], 0, 64) : 0u;
			const uint32_t m = p.n0 < 128u ? p.n0 : 128u;
#pragma unroll
			for (int r = 0; r < 2; ++r) {
				const uint32_t j = (uint32_t)r * 64u + lane;
				p.el[r] = p.ev[r] = 0;
				if (j < m) { p.el[r] = alen[ws0 + j]; p.ev[r] = av[ws0 + j]; }
			}
		};
		auto next_bit = [&]() { int b = -1; if (todo) { b = __ffsll((long long)todo) - 1; todo &= todo - 1; } return b; };
		TrItem c0, c1, c2, c3; // c0: being reduced; c1: stage 3 issued; c2: stage 2 issued; c3: stage 1 issued
		int b0 = next_bit(), b1 = next_bit(), b2 = next_bit(), b3;
		stage1(b0, c0); stage1(b1, c1); stage1(b2, c2);
		stage2(c0); stage2(c1);
		stage3(c0);
		while (b0 >= 0) {
			b3 = next_bit();
			stage1(b3, c3); stage2(c2); stage3(c1);
			__builtin_amdgcn_sched_barrier(0);
			const uint32_t st = c0.st, nv = c0.nv;
			if (c0.dead) { // asg.c:158-161: all arcs of a deleted read go
				for (uint32_t i = lane; i < nv; i += 64) aol[st + i] |= ADEL, ++n_red;
			} else {
				// ROWS = 1: the vertex AND its first neighbour have at most 64 arcs (most vertices: ~ 50 arcs at BASELINE coverage): everything that runs over "two rows
				// of 64" runs over one -- the kernel is bound by the instructions it issues (round 5, visit 1: 2.75 ms per 200 M arcs with every trip to memory but one
				// off the critical path, 2.95 with all of them on it), and the second row of a 50-arc vertex is all instructions and no work
				auto reduce = [&](auto rows_tag) {
				constexpr int ROWS = decltype(rows_tag)::value;
				uint32_t o[ROWS];
#pragma unroll
				for (int r = 0; r < ROWS; ++r) { const uint32_t i = (uint32_t)r * 64u + lane; o[r] = i < nv ? aol[st + i] : 0u; } // used when the flags are written
				uint32_t hbits = 6; while ((1u << hbits) < 2 * nv) ++hbits;
				const uint32_t hsize = 1u << hbits, hmask = hsize - 1;
#pragma unroll
				for (int r = 0; r < ROWS; ++r) {
					const uint32_t i = (uint32_t)r * 64u + lane;
					if (i < nv) { ll[i] = c0.l[r]; lws[i] = c0.ws[r]; lnw[i] = c0.nw[r]; }
				}
				for (uint32_t s = lane; s < hsize; s += 64) hk[s] = TR_EMPTY, hm[s] = 0xffffffffu; // hm[slot] = (index of the FIRST arc to this target) << 2 | mark
				wv_sync();
#pragma unroll
				for (int r = 0; r < ROWS; ++r) { // mark all neighbours 1 (asg.c:162); duplicates share a slot
					const uint32_t i = (uint32_t)r * 64u + lane;
					if (i < nv) {
						const uint32_t key = c0.v[r];
						uint32_t s = tr_hash(key, hbits);
						for (;;) {
							const uint32_t old = atomicCAS(&hk[s], TR_EMPTY, key);
							if (old == TR_EMPTY || old == key) { atomicMin(&hm[s], i << 2 | 1u); slot[i] = s; break; }
							s = (s + 1) & hmask;
						}
					}
				}
				wv_sync();
				const uint32_t L = ll[nv - 1] + fuzz; // asg.c:163
				auto expand_rest = [&](uint32_t ws, uint32_t nw, uint32_t li, uint32_t from) { // lanes over w's arcs from entry `from` on; sorted by len => the loop of asg.c:169 is a prefix
					for (uint32_t j0 = from; j0 < nw; j0 += 64) {
						const uint32_t j = j0 + lane;
						const int ok = j < nw;
						const uint32_t lx = ok ? alen[ws + j] : 0, y = ok ? av[ws + j] : 0;
						int cond = ok && lx + li <= L;
						const uint64_t fail = wv_ballot(ok && !cond);
						if (fail) cond = cond && lane < (unsigned)(__ffsll((long long)fail) - 1);
						if (cond) { ++n_inner; const int sx = tr_find(hk, y, hbits); if (sx >= 0) hm[sx] = (hm[sx] & ~3u) | 2u; } // every writer stores the same word
						if (fail) break;
					}
				};
				{ // the first neighbour: always expanded (asg.c:168: its mark is still 1), its list is in registers
					const uint32_t li = ll[0], n0 = c0.n0;
					bool more = true;
#pragma unroll
					for (int r = 0; r < ROWS; ++r) {
						const uint32_t j = (uint32_t)r * 64u + lane;
						const int ok = more && j < n0 && j < 128u;
						int cond = ok && c0.el[r] + li <= L;
						const uint64_t fail = wv_ballot(ok && !cond);
						if (fail) cond = cond && lane < (unsigned)(__ffsll((long long)fail) - 1);
						if (cond) { ++n_inner; const int sx = tr_find(hk, c0.ev[r], hbits); if (sx >= 0) hm[sx] = (hm[sx] & ~3u) | 2u; }
						if (fail) more = false;
					}
					if (more && n0 > 128u) expand_rest(lws[0], n0, li, 128u);
					wv_sync();
				}
				for (uint32_t base = 0; base < (uint32_t)ROWS * 64u && base < nv; base += 64) {
					const uint32_t i = base + lane;
					const uint32_t myslot = i < nv ? slot[i] : 0;
					uint64_t passed = base == 0 ? 1ull : 0ull; // lanes of this chunk the walk has gone past
					for (;;) {
						uint64_t cand = wv_ballot(i < nv && (hm[myslot] & 3u) == 1u) & ~passed;
						if (!cand) break;
						// up to four pending candidates, a quarter of the wave each: 16 entries of each list in ONE trip
						int cb[4];
#pragma unroll
						for (int k = 0; k < 4; ++k) { cb[k] = cand ? __ffsll((long long)cand) - 1 : -1; cand &= cand - 1; }
						const unsigned g = lane >> 4, sub = lane & 15u;
						const int mine = g == 0 ? cb[0] : g == 1 ? cb[1] : g == 2 ? cb[2] : cb[3];
						const uint32_t i0 = base + (uint32_t)(mine < 0 ? 0 : mine);
						const uint32_t li_g = ll[i0], ws_g = lws[i0], nw_g = mine < 0 ? 0u : lnw[i0];
						const int ok = sub < nw_g;
						const uint32_t lx = ok ? alen[ws_g + sub] : 0, y = ok ? av[ws_g + sub] : 0;
#pragma unroll
						for (int k = 0; k < 4; ++k) {
							if (cb[k] < 0) break; // (uniform)
							const uint32_t ik = base + (uint32_t)cb[k];
							passed |= cb[k] == 63 ? ~0ull : ((2ull << cb[k]) - 1ull);
							if ((hm[slot[ik]] & 3u) != 1u) continue; // an earlier candidate of this batch marked it: asg.c:168 skips it (uniform: every lane reads the same word)
							const int in = g == (unsigned)k;
							int cond = in && ok && lx + li_g <= L;
							const uint64_t fail = wv_ballot(in && ok && !cond);
							if (fail) cond = cond && lane < (unsigned)(__ffsll((long long)fail) - 1);
							if (cond) { ++n_inner; const int sx = tr_find(hk, y, hbits); if (sx >= 0) hm[sx] = (hm[sx] & ~3u) | 2u; }
							const uint32_t nwk = lnw[ik];
							if (!fail && nwk > 16u) expand_rest(lws[ik], nwk, ll[ik], 16u); // a long prefix: the rest of the list, the whole wave over it
							wv_sync();
						}
					}
				}
				// asg.c:181-184: the sweep resets mark[target] at the first arc to a target, so of several arcs to one
				// reduced target (multi-arcs are still present here) only the first is deleted
#pragma unroll
				for (int r = 0; r < ROWS; ++r) { const uint32_t i = (uint32_t)r * 64u + lane; if (i < nv && hm[slot[i]] == (i << 2 | 2u)) aol[st + i] = o[r] | ADEL, ++n_red; }
				wv_sync();
				};
				if (nv <= 64u && c0.n0 <= 64u) reduce(std::integral_constant<int, 1>()); else reduce(std::integral_constant<int, 2>());
			}
			b0 = b1; b1 = b2; b2 = b3;
			c0 = c1; c1 = c2; c2 = c3;
		}
	}
	blk_add_u64(&ctr[CT_NRED], n_red);
	blk_add_u64(&ctr[CT_TRINNER], n_inner);
}

// second tier: vertices with more than TR_CAP arcs; one block per vertex with a private global mark array
// (the reference's own data structure, asg.c:153), lanes over the inner loop.
__global__ __launch_bounds__(256) void k_asg_trans_big(const uint32_t *__restrict__ av, const uint32_t *__restrict__ alen, uint32_t *__restrict__ aol,
                                                        const unsigned long long *__restrict__ idx, uint32_t n_vtx, uint32_t fuzz,
                                                        const uint32_t *__restrict__ ovf, uint32_t n_ovf, uint8_t *__restrict__ marks, unsigned long long *__restrict__ ctr)
{
	uint8_t *mark = marks + (size_t)blockIdx.x * n_vtx;
	__shared__ uint32_t s_go;
	for (uint32_t k = blockIdx.x; k < n_ovf; k += gridDim.x) {
		uint32_t v = ovf[k];
		unsigned long long x = idx[v];
		uint32_t st = (uint32_t)(x >> 32), nv = (uint32_t)x;
		for (uint32_t i = threadIdx.x; i < nv; i += 256) mark[av[st + i]] = 1;
		__syncthreads();
		uint32_t L = alen[st + nv - 1] + fuzz;
		for (uint32_t i = 0; i < nv; ++i) {
			uint32_t w = av[st + i], li = alen[st + i];
			if (threadIdx.x == 0) s_go = mark[w] == 1; // asg.c:168
			__syncthreads();
			if (s_go) {
				unsigned long long xw = idx[w];
				uint32_t ws = (uint32_t)(xw >> 32), nw = (uint32_t)xw;
				for (uint32_t j = threadIdx.x; j < nw; j += 256)
					if (alen[ws + j] + li <= L) { uint32_t y = av[ws + j]; if (mark[y]) mark[y] = 2; } // lengths ascending: a prefix
			}
			__syncthreads();
		}
		if (threadIdx.x == 0) { // asg.c:181-184 literally (the reset makes the sweep order dependent for multi-arcs)
			uint32_t cnt = 0;
			for (uint32_t i = 0; i < nv; ++i) {
				uint32_t y = av[st + i];
				if (mark[y] == 2) aol[st + i] |= ADEL, ++cnt;
				mark[y] = 0;
			}
			if (cnt) atomicAdd(&ctr[CT_NRED], (unsigned long long)cnt);
		}
		__syncthreads();
	}
}

// ------------------------------------------------------------------------------------------------ asg_symm
// asg.c:104-121: within one vertex keep the first arc to each target, delete the later ones
__global__ __launch_bounds__(256) void k_asg_multi(ArcCols a, size_t n, const unsigned long long *__restrict__ idx, unsigned long long *__restrict__ ctr)
{
	uint32_t cnt = 0;
	for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
		uint32_t st = (uint32_t)(idx[a.u[e]] >> 32), v = a.v[e];
		int del = 0;
		for (uint32_t j = st; j < e; ++j) if (a.v[j] == v) { del = 1; break; }
		if (del) a.ol[e] |= ADEL, ++cnt;
	}
	blk_add_u64(&ctr[CT_NMULTI], cnt);
}

// asg.c:124-138: u->v survives only if v^1 -> u^1 is present (del bits are not consulted, as in the reference)
__global__ __launch_bounds__(256) void k_asg_asymm(ArcCols a, size_t n, const unsigned long long *__restrict__ idx, unsigned long long *__restrict__ ctr)
{
	uint32_t cnt = 0;
	uint64_t probed = 0;
	for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
		uint32_t v = a.v[e] ^ 1, u = a.u[e] ^ 1;
		unsigned long long x = idx[v];
		uint32_t st = (uint32_t)(x >> 32), nv = (uint32_t)x, i;
		for (i = 0; i < nv; ++i) if (a.v[st + i] == u) break;
		probed += i < nv ? i + 1 : nv;
		if (i == nv) a.ol[e] |= ADEL, ++cnt; // only the ol column is written; the v column read above is never modified
	}
	blk_add_u64(&ctr[CT_NASYMM], cnt);
	blk_add_u64(&ctr[CT_PROBED], probed);
}

// asg.c:83-101
__global__ __launch_bounds__(256) void k_asg_short(ArcCols a, const unsigned long long *__restrict__ idx, uint32_t n_vtx, float drop_ratio, unsigned long long *__restrict__ ctr)
{
	uint32_t cnt = 0;
	for (uint32_t v = blockIdx.x * 256 + threadIdx.x; v < n_vtx; v += gridDim.x * 256) {
		unsigned long long x = idx[v];
		uint32_t st = (uint32_t)(x >> 32), nv = (uint32_t)x;
		if (nv >= 2) {
			float p = (float)(int32_t)(a.ol[st] & 0x7fffffffu) * drop_ratio;
			uint32_t thres = (uint32_t)((double)p + .499), i;
			for (i = nv - 1; i >= 1 && (a.ol[st + i] & 0x7fffffffu) < thres; --i);
			for (i = i + 1; i < nv; ++i) a.ol[st + i] |= ADEL, ++cnt;
		}
	}
	blk_add_u64(&ctr[CT_NSHORT], cnt);
}

// ------------------------------------------------------------------------------------------------ import / export
__global__ __launch_bounds__(256) void k_arc_from_aos(const asg_arc_t *__restrict__ in, size_t n, ArcCols a)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) { uint4 r = *(const uint4*)(in + i); a.len[i] = r.x; a.u[i] = r.y; a.v[i] = r.z; a.ol[i] = r.w; }
}
__global__ __launch_bounds__(256) void k_seq_from_aos(const uint32_t *__restrict__ seq, uint32_t n_seq, uint32_t *__restrict__ slen, uint8_t *__restrict__ sdel)
{
	uint32_t r = blockIdx.x * 256 + threadIdx.x;
	if (r < n_seq) { uint32_t x = seq[r]; slen[r] = x & 0x7fffffffu; sdel[r] = (uint8_t)(x >> 31); }
}
__global__ __launch_bounds__(256) void k_arc_to_aos(ArcCols a, size_t n, const int32_t *__restrict__ map, asg_arc_t *__restrict__ out)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	uint32_t u = a.u[i], v = a.v[i];
	if (map) u = (uint32_t)map[u >> 1] << 1 | (u & 1), v = (uint32_t)map[v >> 1] << 1 | (v & 1);
	*(uint4*)(out + i) = make_uint4(a.len[i], u, v, a.ol[i]);
}
__global__ __launch_bounds__(256) void k_seq_to_aos(const uint32_t *__restrict__ slen, const uint8_t *__restrict__ sdel, const unsigned long long *__restrict__ idx,
                                                     const int32_t *__restrict__ map, uint32_t n_seq, uint32_t *__restrict__ seq, unsigned long long *__restrict__ idx_out)
{
	uint32_t r = blockIdx.x * 256 + threadIdx.x;
	if (r >= n_seq) return;
	int32_t m = map ? map[r] : (int32_t)r;
	if (m < 0) return;
	seq[m] = (slen[r] & 0x7fffffffu) | (uint32_t)sdel[r] << 31;
	idx_out[2 * (size_t)m] = idx[2 * (size_t)r]; idx_out[2 * (size_t)m + 1] = idx[2 * (size_t)r + 1];
}

// ================================================================================================ host side

// ---- sharded mode: arcs as packed rows {u, v, len, ol} ----
__global__ __launch_bounds__(256) void k_arc_rows_out(ArcCols a, size_t n, uint4 *__restrict__ rows)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) rows[i] = make_uint4(a.u[i], a.v[i], a.len[i], a.ol[i]);
}
__global__ __launch_bounds__(256) void k_arc_rows_in(const uint4 *__restrict__ rows, size_t n, size_t dst0, ArcCols a)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) { uint4 r = rows[i]; a.u[dst0 + i] = r.x; a.v[dst0 + i] = r.y; a.len[dst0 + i] = r.z; a.ol[dst0 + i] = r.w; }
}

__global__ __launch_bounds__(256) void k_rows_permute(const uint4 *__restrict__ in, size_t n, const uint32_t *__restrict__ perm, uint4 *__restrict__ out)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) out[i] = in[perm[i]];
}

// reads of the graph / the map that still has to be applied when it leaves the device
uint32_t graph_nseq(const mahip_ctx *c) { return c->gsq ? c->n_seq_new : c->n_seq; }
static const int32_t *graph_map(mahip_ctx *c) { return c->has_map && !c->gsq ? (const int32_t*)P<int32_t>(c->map) : (const int32_t*)nullptr; }

static int arc_reindex(mahip_ctx *c)
{
	size_t V = 2 * (size_t)graph_nseq(c);
	CHK(dev_reserve(c, c->idx, (V + 2) * 8));
	HIPCHK(hipMemsetAsync(c->idx.p, 0, V * 8, c->st));
	if (c->n_arc) {
		ProfScope ps(c, "k_arc_index", 16.0 * (double)c->n_arc);
		hipLaunchKernelGGL(k_arc_index, dim3(grid_for(c->n_arc, 256)), dim3(256), 0, c->st, (const uint32_t*)P<uint32_t>(c->au[c->ag]), (size_t)c->n_arc, P<unsigned long long>(c->idx));
	}
	HIPCHK(hipGetLastError());
	return 0;
}

// asg.c:72-80 asg_cleanup on the dense list: drop (del | deleted endpoint), keep order, re-index if anything went.
// keep_in: keep[] already holds a pre-filter (ma_sg_gen's candidate flags).
// index_mode: 0 = re-index when something was removed, 1 = always, -1 = never (caller indexes later)
static int arc_cleanup(mahip_ctx *c, size_t n_in, int keep_in, int index_mode)
{
	uint32_t *d_tot = (uint32_t*)(P<unsigned long long>(c->ctr) + CT_TOTAL);
	if (n_in == 0) { c->n_arc = 0; return index_mode < 0 ? 0 : arc_reindex(c); }
	ArcCols in = arcs_of(c, c->ag), out = arcs_of(c, c->ag ^ 1);
	if (!keep_in) {
		ProfScope ps(c, "k_arc_rm", 32.0 * (double)n_in); // asg_arc_rm 32 B per arc
		// arcs_clean: nothing has deleted a read since the arcs were last checked against seq.del (ma_sg_gen's own asg_arc_rm, an earlier cleanup): only the arcs'
		// del bits can have changed -- the cleanup behind the transitive reduction and behind asg_symm -- and the two look-ups per arc are moot
		if (c->arcs_clean) {
			const size_t ng = (n_in + RMC_GROUP - 1) / RMC_GROUP;
			hipLaunchKernelGGL(k_arc_rm_count, dim3((unsigned)ng), dim3(256), 0, c->st, (const uint32_t*)in.ol, n_in, P<uint32_t>(c->keep));
			CHK(scan_exclusive_u32(c, P<uint32_t>(c->keep), P<uint32_t>(c->pos), ng, d_tot));
			hipLaunchKernelGGL(k_arc_rm_write, dim3((unsigned)ng), dim3(256), 0, c->st, in, n_in, out, (const uint32_t*)P<uint32_t>(c->pos));
		} else {
			const size_t nb = (n_in + RM_TILE - 1) / RM_TILE;
			uint32_t *ticket; unsigned long long *state; uint32_t ticket_base, epoch;
			CHK(scan_chain_begin(c, nb, &state, &ticket, &ticket_base, &epoch));
			hipLaunchKernelGGL(k_arc_rm_chain, dim3((unsigned)nb), dim3(256), 0, c->st, in, n_in, (const uint8_t*)P<uint8_t>(c->sdel), out, d_tot, state, ticket, ticket_base, epoch);
		}
	} else {
		ProfScope ps(c, "k_arc_rm", 32.0 * (double)n_in); // asg_arc_rm 32 B per arc
		hipLaunchKernelGGL(k_arc_keep, dim3(grid_for(n_in, 256)), dim3(256), 0, c->st, in, n_in, (const uint8_t*)P<uint8_t>(c->sdel), P<uint32_t>(c->keep), keep_in);
		CHK(scan_exclusive_u32(c, P<uint32_t>(c->keep), P<uint32_t>(c->pos), n_in, d_tot));
		hipLaunchKernelGGL(k_arc_compact, dim3(grid_for(n_in, 256)), dim3(256), 0, c->st, in, n_in, (const uint32_t*)P<uint32_t>(c->keep), (const uint32_t*)P<uint32_t>(c->pos), out);
	}
	CHK(ctr_fetch(c));
	uint32_t n_out = (uint32_t)(c->h_ctr[CT_TOTAL] & 0xffffffffu);
	c->ag ^= 1;
	int changed = n_out != n_in;
	c->n_arc = n_out;
	c->arcs_clean = true;
	if (index_mode > 0 || (index_mode == 0 && changed)) CHK(arc_reindex(c));
	return 0;
}

static int bitlen_u64(uint64_t x) { int b = 0; while (x) ++b, x >>= 1; return b; }

extern "C" int mahip_sg_flags(mahip_ctx_t *c, const ma_opt_t *opt, int use_sub, const uint32_t *seq_len, const uint8_t *seq_del)
{ // asm.c:14-35: seq.len/seq.del, one candidate arc per hit at the hit's slot, local seq.del side effects
	HIPCHK(hipSetDevice(c->dev));
	CHK(hits_need_cols(c, "mahip_sg_gen"));
	size_t n = c->n_hits;
	uint32_t R = c->n_seq;
	CHK(dev_reserve(c, c->slen, ((size_t)R + 4) * 4)); CHK(dev_reserve(c, c->sdel, (size_t)R + 16));
	c->sg_max_hang = opt->max_hang; c->sg_int_frac = opt->int_frac; c->sg_min_ovlp = opt->min_ovlp;
	c->gsq = false; c->arcs_clean = false;
	CHK(ctr_zero(c));
	unsigned long long *ctr = P<unsigned long long>(c->ctr);
	// host-side per-read arrays (per-symbol path) go through the scratch buffers
	const uint32_t *d_len = nullptr; const uint8_t *d_del = nullptr;
	if (seq_len) { CHK(dev_reserve(c, c->big0, ((size_t)R + 4) * 4)); HIPCHK(hipMemcpyAsync(c->big0.p, seq_len, (size_t)R * 4, hipMemcpyHostToDevice, c->st)); d_len = P<uint32_t>(c->big0); }
	if (seq_del) { CHK(dev_reserve(c, c->big1, (size_t)R + 16)); HIPCHK(hipMemcpyAsync(c->big1.p, seq_del, R, hipMemcpyHostToDevice, c->st)); d_del = P<uint8_t>(c->big1); }
	if (!use_sub && !d_len) { mahip_set_error("mahip_sg_gen: need sub or seq_len"); return -1; }
	if (R) hipLaunchKernelGGL(k_sg_seq, dim3(grid_for(R, 256)), dim3(256), 0, c->st, use_sub ? (const uint2*)P<uint2>(c->sub[0]) : (const uint2*)nullptr,
	                          c->has_map ? (const uint8_t*)P<uint8_t>(c->r_del) : (const uint8_t*)nullptr, d_len, d_del, R, P<uint32_t>(c->slen), P<uint8_t>(c->sdel));
	c->ag = 0;
	if (n) CHK(dev_reserve(c, c->sgmask, (n / 64 + 8) * 8)); // candidate bit per hit slot
	if (n) {
		ProfScope ps(c, "k_sg_arcs", 64.0 * (double)c->n_live); // ma_sg_gen 32 r + 16 look-ups + 16 w
		hipLaunchKernelGGL(k_sg_arcs, dim3(grid_for(n, 256, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, gcols_of(c), n, (const uint32_t*)P<uint32_t>(c->slen), P<uint8_t>(c->sdel),
		                   opt->max_hang, opt->int_frac, opt->min_ovlp, ctr,
		                   c->lazy_squeeze ? (const uint8_t*)P<uint8_t>(c->r_del) : (const uint8_t*)nullptr, P<unsigned long long>(c->sgmask));
	}
	HIPCHK(hipGetLastError());
	return 0;
}

// ---- push order (asm.c:18-35: arcs are pushed in hit order) ----
// The slots are grouped by query id only (hits.hip), so the arcs leave k_sg_emit in (qid, input position) order; the reference pushes them
// in ma_hit_sort's order.  That order matters only when arcs with equal (u,len) exist -- or in the stable mode, which is DEFINED as the
// stable sort of the stable push order -- and then it is made here, for the arcs alone: a stable sort of the m arcs by the original
// (qid,qs) of their hits = the (qid, qs, input position) order.  c->val[*gen] <- the permutation; *conflicts <- consecutive arcs with equal keys.
static int push_stable_order(mahip_ctx *c, size_t m, int *gen, uint64_t *conflicts)
{
	unsigned long long *ctr = P<unsigned long long>(c->ctr);
	const int bs = hits_qs_bits(c);
	int bq = 0;
	for (uint64_t x = c->n_seq ? c->n_seq - 1 : 0xffffffffull; x; x >>= 1) ++bq;
	for (int k = 0; k < 2; ++k) { CHK(dev_reserve(c, c->key[k], (m + 1) * 8)); CHK(dev_reserve(c, c->val[k], (m + 1) * 4)); }
	hipLaunchKernelGGL(k_arc_slot_keys, dim3(grid_for(m, 256)), dim3(256), 0, c->st, (const uint32_t*)P<uint32_t>(c->aslot), m, c->d_aos, (const uint32_t*)P<uint32_t>(c->sidx),
	                   P<uint64_t>(c->key[0]), P<uint32_t>(c->val[0]));
	*gen = 0;
	CHK(radix_sort_pairs(c, m, 0, bs, 32, 32 + (bq ? bq : 1), gen));
	HIPCHK(hipMemsetAsync(ctr + ST_PUSH_CONFLICTS, 0, 8, c->st));
	hipLaunchKernelGGL(k_arc_push_conflicts, dim3(grid_for(m, 256, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, (const uint64_t*)P<uint64_t>(c->key[*gen]), m, ctr);
	CHK(ctr_fetch(c));
	*conflicts = c->h_ctr[ST_PUSH_CONFLICTS];
	return 0;
}

// exact: where two consecutive arcs come from hits with equal keys, the reference's (unstable) hit order decides: walk over the hits -- unless the arc sort cannot
// see any of those pairs (k_arc_push_conflicts_seen).  pushed / sorted: the arcs in push-sequence (slot) order and stably sorted by (u,len), when the caller has both
static int push_order(mahip_ctx *c, size_t m, bool exact, int *gen, const ArcCols *pushed = nullptr, const ArcCols *sorted = nullptr)
{
	uint64_t conf = 0;
	bool want_known = false; // c->wantb says which reads' hit order matters
	TieLaps tl(c);
	CHK(push_stable_order(c, m, gen, &conf));
	tl.lap("stable push order + conflicts");
	c->tie.push_conflicts = conf;
	c->tie.push_conflicts_seen = conf;
	static const bool filter_on = !(getenv("MA_TIE_NO_FILTER") && atoi(getenv("MA_TIE_NO_FILTER")) != 0); // A/B handle: walk whenever there is a conflict (until round 5)
	if (exact && conf && pushed && sorted && filter_on && m < 0xffffffffull) {
		unsigned long long *ctr = P<unsigned long long>(c->ctr);
		const int32_t *map = c->has_map ? (const int32_t*)P<int32_t>(c->map) : (const int32_t*)nullptr;
		uint64_t *S = P<uint64_t>(c->key[*gen ^ 1]); // (the other generation was the sort's scratch)
		CHK(dev_reserve(c, c->keep, (m + 16) * 4)); CHK(dev_reserve(c, c->pos, (m + 16) * 4));
		hipLaunchKernelGGL(k_arc_keys_ref, dim3(grid_for(m, 256)), dim3(256), 0, c->st, *sorted, m, map, S);
		hipLaunchKernelGGL(k_tie_pair_flags, dim3(grid_for(m, 256)), dim3(256), 0, c->st, (const uint64_t*)S, m, P<uint32_t>(c->keep));
		CHK(scan_exclusive_u32(c, P<uint32_t>(c->keep), P<uint32_t>(c->pos), m, nullptr));
		HIPCHK(hipMemsetAsync(ctr + ST_PUSH_SEEN, 0, 8, c->st));
		const size_t want_bytes = ((size_t)c->n_seq / 32 + 2) * 4;
		CHK(dev_reserve(c, c->wantb, want_bytes));
		HIPCHK(hipMemsetAsync(c->wantb.p, 0, want_bytes, c->st));
		hipLaunchKernelGGL(k_arc_push_conflicts_seen, dim3(grid_for(m, 256, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, (const uint64_t*)P<uint64_t>(c->key[*gen]), (const uint32_t*)P<uint32_t>(c->val[*gen]),
		                   *pushed, map, (const uint64_t*)S, (const uint32_t*)P<uint32_t>(c->pos), (uint32_t)m, ctr, P<uint32_t>(c->wantb));
		CHK(ctr_fetch(c));
		c->tie.push_conflicts_seen = c->h_ctr[ST_PUSH_SEEN];
		want_known = true;
		tl.lap("conflicts the arc sort can see");
	}
	if (exact && c->tie.push_conflicts_seen) {
		const int g_keep = *gen; (void)g_keep;
		static const bool walk_all = getenv("MA_TIE_WALK_ALL") && atoi(getenv("MA_TIE_WALK_ALL")) != 0; // A/B handle: the whole hit order although only some reads' is needed
		CHK(hits_reference_rank(c, false, want_known && !walk_all)); // uses key[]/val[] as scratch
		for (int k = 0; k < 2; ++k) { CHK(dev_reserve(c, c->key[k], (m + 1) * 8)); CHK(dev_reserve(c, c->val[k], (m + 1) * 4)); }
		hipLaunchKernelGGL(k_arc_push_keys, dim3(grid_for(m, 256)), dim3(256), 0, c->st, (const uint32_t*)P<uint32_t>(c->aslot), (const uint32_t*)P<uint32_t>(c->hrank), m,
		                   P<uint64_t>(c->key[0]), P<uint32_t>(c->val[0]));
		*gen = 0;
		int bh = 0;
		for (uint64_t x = c->n_in; x; x >>= 1) ++bh;
		CHK(radix_sort_pairs(c, m, 0, bh ? bh : 1, 0, 0, gen)); // hrank = position among all input records
	}
	return 0;
}

extern "C" int mahip_sg_finish(mahip_ctx_t *c, uint32_t *n_arc)
{ // asg_cleanup (asg.c:72-80) on the candidates: arc_rm (order preserving), sort by (u,len), index
	HIPCHK(hipSetDevice(c->dev));
	size_t n = c->n_hits;
	uint32_t R = c->n_seq;
	CHK(ctr_fetch(c)); // CT_MAXLEN / CT_LIVE of pass A
	c->n_live = (size_t)c->h_ctr[CT_LIVE];
	if (c->prof) prof_patch_last(c, "k_sg_arcs", 64.0 * (double)c->n_live); // units = hits left after containment (64 B each)
	c->n_arc = 0; c->ag = 0;
	const bool sharded = c->q_beg > 0 || (c->n_seq && c->q_end < c->n_seq);
	const bool want_slots = c->sorted_here && c->sidx.p != nullptr; // the arcs can be put into the reference's push order (on a shard: by the orchestrator, after the exchange)
	const int blen = bitlen_u64(c->h_ctr[CT_MAXLEN]);
	{ const uint64_t keep_hit_ties = c->tie.hit_ties; const int keep_walk = c->hrank_ready; memset(&c->tie, 0, sizeof(c->tie)); c->tie.hit_ties = keep_hit_ties; c->tie.hit_walk = keep_walk; }
	if (n) {
		const size_t n_tiles = (n + SG_TILE - 1) / SG_TILE;
		const unsigned long long *lazy = (const unsigned long long*)P<unsigned long long>(c->sgmask); // pass A's candidate bits
		uint32_t *d_tot = (uint32_t*)(P<unsigned long long>(c->ctr) + CT_TOTAL);
		ArcCols none = { nullptr, nullptr, nullptr, nullptr };
		CHK(dev_reserve(c, c->keep, (n + 16) * 4)); CHK(dev_reserve(c, c->pos, (n + 16) * 4)); // also what reserve_arcs(n_arc <= n) asks for: no reallocation between the passes
		{
			ProfScope ps(c, "k_sg_emit", 4.0 * (double)n + 64.0 * (double)c->n_live);
			hipLaunchKernelGGL(k_sg_emit, dim3((unsigned)n_tiles), dim3(256), 0, c->st, gcols_of(c), n, (const uint32_t*)P<uint32_t>(c->slen), (const uint8_t*)P<uint8_t>(c->sdel),
			                   c->sg_max_hang, c->sg_int_frac, c->sg_min_ovlp, lazy, P<uint32_t>(c->keep), (const uint32_t*)nullptr, none, (uint32_t*)nullptr, (uint32_t*)nullptr, (unsigned long long*)nullptr);
		}
		CHK(scan_exclusive_u32(c, P<uint32_t>(c->keep), P<uint32_t>(c->pos), n_tiles, d_tot));
		CHK(ctr_fetch(c));
		c->n_arc = (uint32_t)(c->h_ctr[CT_TOTAL] & 0xffffffffu);
		CHK(reserve_arcs(c, c->n_arc));
		if (want_slots) CHK(dev_reserve(c, c->aslot, ((size_t)c->n_arc + 4) * 4));
		CHK(dev_reserve(c, c->apos, (n / 64 + 8) * 4)); // first arc of every 64-slot word (pass C -> k_arc_group_sort)
		if (c->n_arc) {
			ProfScope ps(c, "k_sg_emit", 4.0 * (double)n + 64.0 * (double)c->n_live + 16.0 * (double)c->n_arc);
			hipLaunchKernelGGL(k_sg_emit, dim3((unsigned)n_tiles), dim3(256), 0, c->st, gcols_of(c), n, (const uint32_t*)P<uint32_t>(c->slen), (const uint8_t*)P<uint8_t>(c->sdel),
			                   c->sg_max_hang, c->sg_int_frac, c->sg_min_ovlp, lazy, (uint32_t*)nullptr, (const uint32_t*)P<uint32_t>(c->pos), arcs_of(c, 0),
			                   want_slots ? P<uint32_t>(c->aslot) : (uint32_t*)nullptr, P<uint32_t>(c->apos), P<unsigned long long>(c->sgmask)); // (the candidate bits become "kept" bits in place: every thread owns its word)
		}
	} else CHK(reserve_arcs(c, 0));
	c->n_push = 0; c->push_ordered = false;
	if (sharded && c->tie_mode != 0) { // keep this rank's arcs in push order: the exchange overwrites the arc arrays, a tie repair needs them (sharded.c)
		CHK(dev_reserve(c, c->pushrows[0], ((size_t)c->n_arc + 1) * 16));
		if (c->n_arc) hipLaunchKernelGGL(k_arc_rows_out, dim3(grid_for(c->n_arc, 256)), dim3(256), 0, c->st, arcs_of(c, c->ag), (size_t)c->n_arc, (uint4*)c->pushrows[0].p);
		c->n_push = c->n_arc;
	}
	if (c->n_arc > 1) {
		size_t m = c->n_arc;
		for (int k = 0; k < 2; ++k) { CHK(dev_reserve(c, c->key[k], (m + 1) * 8)); CHK(dev_reserve(c, c->val[k], (m + 1) * 4)); }
		ArcCols in = arcs_of(c, c->ag), out = arcs_of(c, c->ag ^ 1);
		unsigned long long *ctr = P<unsigned long long>(c->ctr);
		const int32_t *map = c->has_map ? (const int32_t*)P<int32_t>(c->map) : (const int32_t*)nullptr;
		int gen = 0, need_walk = c->tie_mode == 1 && !sharded;
		TieLaps tl(c);
		if (c->tie_mode == 0 && want_slots) { // the documented stable order: the stable sort of the stable push order
			int g0 = 0;
			CHK(push_order(c, m, false, &g0));
			hipLaunchKernelGGL(k_arc_permute, dim3(grid_for(m, 256)), dim3(256), 0, c->st, in, m, (const uint32_t*)P<uint32_t>(c->val[g0]), out);
			c->ag ^= 1;
			in = arcs_of(c, c->ag); out = arcs_of(c, c->ag ^ 1);
		}
		bool idx_done = false;
		if (c->tie_mode != 1 || sharded) { // stable sort (asg.c:24 up to the order of equal keys), then the census (on a shard: after the exchange, sharded.c)
			// fast path: the arcs are grouped by read (push order): one in-register sort per read that also writes the CSR index and takes the census
			bool fast = blen + AG_IB + 1 <= 31 && !getenv("MA_ARC_RADIX");
			if (fast) {
				const size_t V = 2 * (size_t)R;
				const uint32_t q_lo = sharded ? c->q_beg : 0u, q_hi = sharded && c->q_end < R ? c->q_end : R, Rr = q_hi > q_lo ? q_hi - q_lo : 1;
				CHK(dev_reserve(c, c->idx, (V + 2) * 8));
				HIPCHK(hipMemsetAsync(c->idx.p, 0, V * 8, c->st));
				HIPCHK(hipMemsetAsync(ctr + CT_OVF2, 0, 8, c->st));
				HIPCHK(hipMemsetAsync(ctr + CT_STICKY, 0, (64 - CT_STICKY) * 8, c->st));
				{
					ProfScope ps(c, "k_arc_group_sort", 48.0 * (double)m); // arc sort 32 + index 16 B per arc
					const unsigned grid = grid_for(((size_t)Rr + 63) / 64, 4, MA_STREAM_BLOCKS);
					const uint32_t *goff = (const uint32_t*)P<uint32_t>(c->goff), *wpos = (const uint32_t*)P<uint32_t>(c->apos);
					const unsigned long long *km = (const unsigned long long*)P<unsigned long long>(c->sgmask);
					hipLaunchKernelGGL(k_arc_group_sort<true>, dim3(grid), dim3(256), 0, c->st, in, out, goff, wpos, km, (uint32_t)n, (uint32_t)m, q_lo, q_hi, blen ? blen : 1, P<unsigned long long>(c->idx), ctr);
					hipLaunchKernelGGL(k_arc_group_sort<false>, dim3(grid), dim3(256), 0, c->st, in, out, goff, wpos, km, (uint32_t)n, (uint32_t)m, q_lo, q_hi, blen ? blen : 1, P<unsigned long long>(c->idx), ctr);
				}
				CHK(ctr_fetch(c));
				if (c->h_ctr[CT_OVF2]) fast = false; // a read with more than AG_MAX arcs, or a stretch that holds another read's arcs (hit groups not in ascending id order): the general sort below
				else {
					idx_done = true;
					if (c->tie_mode == 2 && !sharded) {
						c->tie.arc_tie_groups = c->h_ctr[ST_ARC_TIE_GROUPS]; c->tie.arc_tie_arcs = c->h_ctr[ST_ARC_TIE_ARCS];
						need_walk = c->tie.arc_tie_groups > 0;
					}
				}
			}
			if (!fast) {
			hipLaunchKernelGGL(k_arc_keys, dim3(grid_for(m, 256)), dim3(256), 0, c->st, in, m, P<uint64_t>(c->key[0]), P<uint32_t>(c->val[0]));
			c->radix_arcs = true; // (profile names: the arc sort's digit passes apart from the hit sort's)
			CHK(radix_sort_pairs(c, m, 0, blen, 32, 32 + bitlen_u64(2ull * R), &gen));
			c->radix_arcs = false;
			{
				ProfScope ps(c, "k_arc_permute", 36.0 * (double)m);
				hipLaunchKernelGGL(k_arc_permute, dim3(grid_for(m, 256)), dim3(256), 0, c->st, in, m, (const uint32_t*)P<uint32_t>(c->val[gen]), out);
			}
			if (c->tie_mode == 2 && !sharded) { // without equal (u,len) keys the sorted sequence is unique: whatever order the arcs were pushed in
				HIPCHK(hipMemsetAsync(ctr + CT_STICKY, 0, (64 - CT_STICKY) * 8, c->st));
				ProfScope ps(c, "k_arc_tie_census", 8.0 * (double)m);
				hipLaunchKernelGGL(k_arc_tie_census, dim3(grid_for(m, 256, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, (const uint64_t*)P<uint64_t>(c->key[gen]), m, ctr);
				CHK(ctr_fetch(c));
				c->tie.arc_tie_groups = c->h_ctr[ST_ARC_TIE_GROUPS]; c->tie.arc_tie_arcs = c->h_ctr[ST_ARC_TIE_ARCS];
				need_walk = c->tie.arc_tie_groups > 0;
			}
			}
			tl.lap("stable arc sort + census");
		}
		if (need_walk) {
			// (1) the push order: arcs leave ma_sg_gen in the order ma_hit_sort left the hits in (asm.c:18-35)
			if (want_slots) {
				int g2 = 0;
				const bool have_sorted = c->tie_mode != 1 || sharded; // `out` holds the stable sort's result (forced mode skips that sort: it walks unconditionally)
				CHK(push_order(c, m, true, &g2, &in, have_sorted ? &out : nullptr));
				hipLaunchKernelGGL(k_arc_permute, dim3(grid_for(m, 256)), dim3(256), 0, c->st, in, m, (const uint32_t*)P<uint32_t>(c->val[g2]), out);
				c->ag ^= 1; // `out` now holds the arcs in the reference's push order
				in = arcs_of(c, c->ag); out = arcs_of(c, c->ag ^ 1);
				tl.lap("push order (all of it)");
			}
			// (2) the reference's sort of that sequence (ksort.h:134-183 on ul with squeezed ids): permutation from the host walk
			hipLaunchKernelGGL(k_arc_keys_ref, dim3(grid_for(m, 256)), dim3(256), 0, c->st, in, m, map, P<uint64_t>(c->key[0]));
			CHK(reference_order(c, P<uint64_t>(c->key[0]), m, P<uint32_t>(c->val[1])));
			{
				ProfScope ps(c, "k_arc_permute", 36.0 * (double)m);
				hipLaunchKernelGGL(k_arc_permute, dim3(grid_for(m, 256)), dim3(256), 0, c->st, in, m, (const uint32_t*)P<uint32_t>(c->val[1]), out);
			}
			c->tie.arc_walk = 1;
			tl.lap("arc walk (all of it)");
			walk_scratch_release(c);
			idx_done = false; // the walk's order moved arcs inside their lists: same runs, but index them from what is there now
		}
		c->ag ^= 1;
		if (!idx_done) CHK(arc_reindex(c));
	} else CHK(arc_reindex(c));
	HIPCHK(hipGetLastError());
	c->graph_ready = true;
	c->arcs_clean = true; // k_sg_emit kept only arcs between live reads (asg.c:57-70 on the fresh arcs)
	if (n_arc) *n_arc = c->n_arc;
	return 0;
}

extern "C" int mahip_sg_gen(mahip_ctx_t *c, const ma_opt_t *opt, int use_sub, const uint32_t *seq_len, const uint8_t *seq_del, uint32_t *n_arc)
{
	CHK(mahip_sg_flags(c, opt, use_sub, seq_len, seq_del));
	return mahip_sg_finish(c, n_arc);
}

extern "C" int mahip_asg_export_rows(mahip_ctx_t *c, void *d_dst)
{
	HIPCHK(hipSetDevice(c->dev));
	if (!c->graph_ready) { mahip_set_error("mahip_asg_export_rows: no graph"); return -1; }
	if (c->n_arc) hipLaunchKernelGGL(k_arc_rows_out, dim3(grid_for(c->n_arc, 256)), dim3(256), 0, c->st, arcs_of(c, c->ag), (size_t)c->n_arc, (uint4*)d_dst);
	if (xchg_needs_sync(c)) HIPCHK(hipStreamSynchronize(c->st)); // the exchange runs on somebody else's stream (mahip_internal.hpp)
	return 0;
}

extern "C" int mahip_asg_import_rows(mahip_ctx_t *c, const void *d_src, const uint32_t *counts, int n_ranks, size_t stride)
{ // blocks arrive in rank order = read-range order, each sorted by (u,len): the concatenation is the sorted global list
	HIPCHK(hipSetDevice(c->dev));
	size_t tot = 0;
	for (int r = 0; r < n_ranks; ++r) tot += counts[r];
	if (tot >= 0x7fffffffull) { mahip_set_error("mahip_asg_import_rows: too many arcs"); return -1; }
	CHK(reserve_arcs(c, tot));
	c->ag = 0; c->gsq = false; c->arcs_clean = false;
	ArcCols a = arcs_of(c, 0);
	size_t off = 0;
	for (int r = 0; r < n_ranks; ++r) {
		if (counts[r]) hipLaunchKernelGGL(k_arc_rows_in, dim3(grid_for(counts[r], 256)), dim3(256), 0, c->st, (const uint4*)d_src + (size_t)r * stride, (size_t)counts[r], off, a);
		off += counts[r];
	}
	c->n_arc = (uint32_t)tot;
	CHK(arc_reindex(c));
	memset(&c->tie, 0, sizeof(c->tie));
	if (c->tie_mode != 0 && tot > 1) { // every rank now holds the whole sorted graph: the tie census of the single-GPU path, reported (the repair
		// needs the global push order and the global hit order: not available on shards)
		unsigned long long *ctr = P<unsigned long long>(c->ctr);
		CHK(dev_reserve(c, c->key[0], (tot + 1) * 8)); CHK(dev_reserve(c, c->val[0], (tot + 1) * 4));
		hipLaunchKernelGGL(k_arc_keys, dim3(grid_for(tot, 256)), dim3(256), 0, c->st, a, tot, P<uint64_t>(c->key[0]), P<uint32_t>(c->val[0]));
		HIPCHK(hipMemsetAsync(ctr + CT_STICKY, 0, (64 - CT_STICKY) * 8, c->st));
		hipLaunchKernelGGL(k_arc_tie_census, dim3(grid_for(tot, 256, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, (const uint64_t*)P<uint64_t>(c->key[0]), tot, ctr);
		CHK(ctr_fetch(c));
		c->tie.arc_tie_groups = c->h_ctr[ST_ARC_TIE_GROUPS]; c->tie.arc_tie_arcs = c->h_ctr[ST_ARC_TIE_ARCS];
		c->tie.unrepaired = c->tie.arc_tie_groups > 0;
	}
	if (xchg_needs_sync(c)) HIPCHK(hipStreamSynchronize(c->st)); // the exchange runs on somebody else's stream (mahip_internal.hpp)
	c->graph_ready = true;
	return 0;
}

// ---- tie repair on shards (host/sharded.c) ----
// this rank's pushed arcs into the stable push order (pushrows[1]); *n_conf = consecutive arcs from hits with equal original (qid,qs) keys
extern "C" int mahip_sg_push_conflicts(mahip_ctx_t *c, uint64_t *n_conf)
{
	HIPCHK(hipSetDevice(c->dev));
	*n_conf = 0;
	const size_t m = c->n_push;
	CHK(dev_reserve(c, c->pushrows[1], (m + 1) * 16));
	if (m < 2 || !c->aslot.p || !c->sidx.p) {
		if (m) HIPCHK(hipMemcpyAsync(c->pushrows[1].p, c->pushrows[0].p, m * 16, hipMemcpyDeviceToDevice, c->st));
		c->push_ordered = true;
		return 0;
	}
	int g = 0;
	CHK(push_stable_order(c, m, &g, n_conf));
	hipLaunchKernelGGL(k_rows_permute, dim3(grid_for(m, 256)), dim3(256), 0, c->st, (const uint4*)c->pushrows[0].p, m, (const uint32_t*)P<uint32_t>(c->val[g]), (uint4*)c->pushrows[1].p);
	HIPCHK(hipGetLastError());
	c->push_ordered = true;
	return 0;
}

// this rank's pushed arcs into the order the reference's hit order gives them (needs the whole input on this context: mahip_set_full_input)
extern "C" int mahip_sg_push_fix(mahip_ctx_t *c)
{
	HIPCHK(hipSetDevice(c->dev));
	const size_t m = c->n_push;
	if (m < 2) return 0;
	CHK(hits_reference_rank(c, true)); // the one entry that may run the ranks' collective form
	for (int k = 0; k < 2; ++k) { CHK(dev_reserve(c, c->key[k], (m + 1) * 8)); CHK(dev_reserve(c, c->val[k], (m + 1) * 4)); }
	CHK(dev_reserve(c, c->pushrows[1], (m + 1) * 16));
	hipLaunchKernelGGL(k_arc_push_keys, dim3(grid_for(m, 256)), dim3(256), 0, c->st, (const uint32_t*)P<uint32_t>(c->aslot), (const uint32_t*)P<uint32_t>(c->hrank), m, P<uint64_t>(c->key[0]), P<uint32_t>(c->val[0]));
	int g = 0;
	CHK(radix_sort_pairs(c, m, 0, 32, 0, 0, &g)); // hrank holds positions in the global order: up to 32 bits
	hipLaunchKernelGGL(k_rows_permute, dim3(grid_for(m, 256)), dim3(256), 0, c->st, (const uint4*)c->pushrows[0].p, m, (const uint32_t*)P<uint32_t>(c->val[g]), (uint4*)c->pushrows[1].p);
	HIPCHK(hipGetLastError());
	c->push_ordered = true;
	return 0;
}

extern "C" int mahip_asg_export_rows_push(mahip_ctx_t *c, void *d_dst)
{
	HIPCHK(hipSetDevice(c->dev));
	if (!c->push_ordered) { uint64_t conf; CHK(mahip_sg_push_conflicts(c, &conf)); }
	if (c->n_push) HIPCHK(hipMemcpyAsync(d_dst, c->pushrows[1].p, (size_t)c->n_push * 16, hipMemcpyDeviceToDevice, c->st));
	return 0;
}

// Replace the graph by the reference's sort of the GLOBAL push sequence: n_ranks blocks of rows in push order (rank order = hit order), keys from
// the squeezed ids, the host walk (identical on every rank), permute, index.  Every rank calls it with the same data and gets the same graph.
extern "C" int mahip_asg_import_push_rows(mahip_ctx_t *c, const void *d_src, const uint32_t *counts, int n_ranks, size_t stride)
{
	HIPCHK(hipSetDevice(c->dev));
	size_t tot = 0;
	for (int r = 0; r < n_ranks; ++r) tot += counts[r];
	if (tot >= 0x7fffffffull) { mahip_set_error("mahip_asg_import_push_rows: too many arcs"); return -1; }
	CHK(reserve_arcs(c, tot));
	c->ag = 0; c->gsq = false; c->arcs_clean = false;
	ArcCols in = arcs_of(c, 0), out = arcs_of(c, 1);
	size_t off = 0;
	for (int r = 0; r < n_ranks; ++r) {
		if (counts[r]) hipLaunchKernelGGL(k_arc_rows_in, dim3(grid_for(counts[r], 256)), dim3(256), 0, c->st, (const uint4*)d_src + (size_t)r * stride, (size_t)counts[r], off, in);
		off += counts[r];
	}
	c->n_arc = (uint32_t)tot;
	if (tot > 1) {
		for (int k = 0; k < 2; ++k) { CHK(dev_reserve(c, c->key[k], (tot + 1) * 8)); CHK(dev_reserve(c, c->val[k], (tot + 1) * 4)); }
		hipLaunchKernelGGL(k_arc_keys_ref, dim3(grid_for(tot, 256)), dim3(256), 0, c->st, in, tot, graph_map(c), P<uint64_t>(c->key[0]));
		CHK(reference_order(c, P<uint64_t>(c->key[0]), tot, P<uint32_t>(c->val[1])));
		hipLaunchKernelGGL(k_arc_permute, dim3(grid_for(tot, 256)), dim3(256), 0, c->st, in, tot, (const uint32_t*)P<uint32_t>(c->val[1]), out);
		c->ag = 1;
	}
	CHK(arc_reindex(c));
	walk_scratch_release(c);
	c->tie.arc_walk = 1; c->tie.unrepaired = 0;
	c->graph_ready = true;
	return 0;
}

extern "C" int mahip_asg_flags_out(mahip_ctx_t *c, void *d_dst, size_t first, size_t count)
{
	HIPCHK(hipSetDevice(c->dev));
	if (first + count > c->n_arc) { mahip_set_error("mahip_asg_flags_out: bad range"); return -1; }
	if (count) HIPCHK(hipMemcpyAsync(d_dst, P<uint32_t>(c->aol[c->ag]) + first, count * 4, hipMemcpyDeviceToDevice, c->st));
	if (xchg_needs_sync(c)) HIPCHK(hipStreamSynchronize(c->st)); // the exchange runs on somebody else's stream (mahip_internal.hpp)
	return 0;
}

extern "C" int mahip_asg_flags_in(mahip_ctx_t *c, const void *d_src, size_t first, size_t count)
{
	HIPCHK(hipSetDevice(c->dev));
	if (first + count > c->n_arc) { mahip_set_error("mahip_asg_flags_in: bad range"); return -1; }
	if (count) HIPCHK(hipMemcpyAsync(P<uint32_t>(c->aol[c->ag]) + first, d_src, count * 4, hipMemcpyDeviceToDevice, c->st));
	if (xchg_needs_sync(c)) HIPCHK(hipStreamSynchronize(c->st)); // the exchange runs on somebody else's stream (mahip_internal.hpp)
	return 0;
}

int graph_cleanup(mahip_ctx *c) { return arc_cleanup(c, c->n_arc, 0, 0); }

// ---- the graph in the squeezed numbering (sdict.c:69-86 applied to the graph, as the reference has it from ma_sg_gen on) ----
__global__ __launch_bounds__(256) void k_arc_squeeze_ids(uint32_t *__restrict__ au, uint32_t *__restrict__ av, size_t n, const int32_t *__restrict__ map)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) {
		uint32_t u = au[i], v = av[i];
		au[i] = (uint32_t)map[u >> 1] << 1 | (u & 1); av[i] = (uint32_t)map[v >> 1] << 1 | (v & 1);
	}
}
__global__ __launch_bounds__(256) void k_seq_squeeze(const uint32_t *__restrict__ slen, const uint8_t *__restrict__ sdel, const unsigned long long *__restrict__ idx,
                                                      const int32_t *__restrict__ map, uint32_t n_seq, uint32_t *__restrict__ slen2, uint8_t *__restrict__ sdel2, unsigned long long *__restrict__ idx2)
{
	uint32_t r = blockIdx.x * 256 + threadIdx.x;
	if (r >= n_seq) return;
	int32_t m = map[r];
	if (m < 0) return;
	slen2[m] = slen[r]; sdel2[m] = sdel[r];
	idx2[2 * (size_t)m] = idx[2 * (size_t)r]; idx2[2 * (size_t)m + 1] = idx[2 * (size_t)r + 1];
}

// Renumber the device graph to the squeezed read ids.  After ma_sg_gen only surviving reads carry arcs, so this is a relabelling:
// arc order, CSR positions and every later result are unchanged; the cleaners and the unitig pass then sweep n_seq_new reads
// instead of n_seq.
extern "C" int mahip_asg_squeeze(mahip_ctx_t *c)
{
	HIPCHK(hipSetDevice(c->dev));
	if (!c->graph_ready) { mahip_set_error("mahip_asg_squeeze: no graph"); return -1; }
	if (!c->has_map || c->gsq) return 0;
	const uint32_t R = c->n_seq, Rn = c->n_seq_new;
	const int32_t *map = (const int32_t*)P<int32_t>(c->map);
	CHK(dev_reserve(c, c->big0, ((size_t)Rn + 4) * 4)); CHK(dev_reserve(c, c->big1, (size_t)Rn + 16));
	CHK(dev_reserve(c, c->pos, (2 * (size_t)Rn + 2) * 8));
	if (c->n_arc) hipLaunchKernelGGL(k_arc_squeeze_ids, dim3(grid_for(c->n_arc, 256)), dim3(256), 0, c->st, P<uint32_t>(c->au[c->ag]), P<uint32_t>(c->av[c->ag]), (size_t)c->n_arc, map);
	if (R) hipLaunchKernelGGL(k_seq_squeeze, dim3(grid_for(R, 256)), dim3(256), 0, c->st, (const uint32_t*)P<uint32_t>(c->slen), (const uint8_t*)P<uint8_t>(c->sdel),
	                          (const unsigned long long*)P<unsigned long long>(c->idx), map, R, P<uint32_t>(c->big0), P<uint8_t>(c->big1), P<unsigned long long>(c->pos));
	if (Rn) {
		HIPCHK(hipMemcpyAsync(c->slen.p, c->big0.p, (size_t)Rn * 4, hipMemcpyDeviceToDevice, c->st));
		HIPCHK(hipMemcpyAsync(c->sdel.p, c->big1.p, (size_t)Rn, hipMemcpyDeviceToDevice, c->st));
		HIPCHK(hipMemcpyAsync(c->idx.p, c->pos.p, 2 * (size_t)Rn * 8, hipMemcpyDeviceToDevice, c->st));
	}
	HIPCHK(hipGetLastError());
	c->gsq = true;
	return 0;
}

__global__ __launch_bounds__(256) void k_sub_squeeze_g(const uint2 *__restrict__ sub, const int32_t *__restrict__ map, uint32_t n_seq, uint2 *__restrict__ out)
{ // sdict.c:78-85 on the interval array: survivors move to their new ids
	uint32_t r = blockIdx.x * 256 + threadIdx.x;
	if (r < n_seq && map[r] >= 0) out[map[r]] = sub[r];
}

extern "C" int mahip_tail_handoff(mahip_ctx_t *s, mahip_ctx_t *d)
{
	if (s == d || s->dev != d->dev) { mahip_set_error("mahip_tail_handoff: two contexts on one device"); return -1; }
	HIPCHK(hipSetDevice(s->dev));
	if (!s->graph_ready) { mahip_set_error("mahip_tail_handoff: no graph"); return -1; }
	CHK(mahip_asg_squeeze(s));
	const uint32_t Rn = s->has_map ? s->n_seq_new : s->n_seq;
	const size_t n = s->n_arc;
	const bool have_sub = s->sub[0].p != nullptr && s->n_seq > 0;
	// the receiving buffers (allocation only: nothing of `d` is in flight -- its owner finished the previous batch before asking for the next)
	CHK(reserve_arcs(d, n));
	CHK(dev_reserve(d, d->slen, ((size_t)Rn + 4) * 4)); CHK(dev_reserve(d, d->sdel, (size_t)Rn + 16));
	CHK(dev_reserve(d, d->idx, (2 * (size_t)Rn + 2) * 8));
	CHK(dev_reserve(d, d->sub[0], ((size_t)Rn + 1) * 8)); CHK(dev_reserve(d, d->surv, ((size_t)Rn + 1) * 4));
	// everything below is queued on the sender's stream, behind the kernels that made the graph
	if (n) {
		HIPCHK(hipMemcpyAsync(d->au[0].p, s->au[s->ag].p, n * 4, hipMemcpyDeviceToDevice, s->st));
		HIPCHK(hipMemcpyAsync(d->av[0].p, s->av[s->ag].p, n * 4, hipMemcpyDeviceToDevice, s->st));
		HIPCHK(hipMemcpyAsync(d->alen[0].p, s->alen[s->ag].p, n * 4, hipMemcpyDeviceToDevice, s->st));
		HIPCHK(hipMemcpyAsync(d->aol[0].p, s->aol[s->ag].p, n * 4, hipMemcpyDeviceToDevice, s->st));
	}
	if (Rn) {
		HIPCHK(hipMemcpyAsync(d->slen.p, s->slen.p, (size_t)Rn * 4, hipMemcpyDeviceToDevice, s->st));
		HIPCHK(hipMemcpyAsync(d->sdel.p, s->sdel.p, (size_t)Rn, hipMemcpyDeviceToDevice, s->st));
		HIPCHK(hipMemcpyAsync(d->idx.p, s->idx.p, 2 * (size_t)Rn * 8, hipMemcpyDeviceToDevice, s->st));
		if (have_sub) {
			if (s->has_map) hipLaunchKernelGGL(k_sub_squeeze_g, dim3(grid_for(s->n_seq, 256)), dim3(256), 0, s->st, (const uint2*)P<uint2>(s->sub[0]), (const int32_t*)P<int32_t>(s->map), s->n_seq, P<uint2>(d->sub[0]));
			else HIPCHK(hipMemcpyAsync(d->sub[0].p, s->sub[0].p, (size_t)Rn * 8, hipMemcpyDeviceToDevice, s->st));
		}
		if (s->has_map || s->surv_ready) HIPCHK(hipMemcpyAsync(d->surv.p, s->surv.p, (size_t)Rn * 4, hipMemcpyDeviceToDevice, s->st));
	}
	HIPCHK(hipGetLastError());
	HIPCHK(hipStreamSynchronize(s->st));
	d->n_seq = Rn; d->n_seq_new = Rn; d->has_map = false; d->surv_ready = s->has_map || s->surv_ready; d->gsq = true; d->lazy_squeeze = false;
	d->soa_ready = false; d->gather_pending = false; d->sorted_here = false; d->n_hits = 0; d->n_live = 0;
	d->ag = 0; d->n_arc = (uint32_t)n; d->graph_ready = true;
	d->arcs_clean = s->arcs_clean; // the copied sdel / arcs are as checked against each other as the source's were (a reused tail context must not keep the last batch's answer)
	d->tie = s->tie;
	return 0;
}

extern "C" int mahip_asg_cleanup(mahip_ctx_t *c, uint32_t *n_arc)
{
	HIPCHK(hipSetDevice(c->dev));
	if (!c->graph_ready) { mahip_set_error("mahip_asg_cleanup: no graph"); return -1; }
	CHK(ctr_zero(c));
	CHK(arc_cleanup(c, c->n_arc, 0, 0));
	if (n_arc) *n_arc = c->n_arc;
	return 0;
}

extern "C" int mahip_asg_upload(mahip_ctx_t *c, const asg_t *g)
{
	HIPCHK(hipSetDevice(c->dev));
	size_t n = g->n_arc;
	uint32_t R = g->n_seq;
	c->n_seq = R; c->n_seq_new = R; c->has_map = false; c->surv_ready = false; c->soa_ready = false; c->gather_pending = false; c->gsq = false; c->arcs_clean = false;
	CHK(reserve_arcs(c, n));
	CHK(dev_reserve(c, c->slen, ((size_t)R + 4) * 4)); CHK(dev_reserve(c, c->sdel, (size_t)R + 16));
	CHK(dev_reserve(c, c->idx, (2 * (size_t)R + 2) * 8));
	CHK(dev_reserve(c, c->key[0], (n + 1) * 16 + ((size_t)R + 4) * 4));
	if (n) {
		HIPCHK(hipMemcpyAsync(c->key[0].p, g->arc, n * 16, hipMemcpyHostToDevice, c->st));
		hipLaunchKernelGGL(k_arc_from_aos, dim3(grid_for(n, 256)), dim3(256), 0, c->st, (const asg_arc_t*)c->key[0].p, n, arcs_of(c, 0));
	}
	if (R) {
		uint32_t *d_seq = (uint32_t*)((char*)c->key[0].p + (n + 1) * 16);
		HIPCHK(hipMemcpyAsync(d_seq, g->seq, (size_t)R * 4, hipMemcpyHostToDevice, c->st));
		hipLaunchKernelGGL(k_seq_from_aos, dim3(grid_for(R, 256)), dim3(256), 0, c->st, (const uint32_t*)d_seq, R, P<uint32_t>(c->slen), P<uint8_t>(c->sdel));
	}
	c->ag = 0; c->n_arc = (uint32_t)n;
	if (g->idx) { if (R) HIPCHK(hipMemcpyAsync(c->idx.p, g->idx, 2 * (size_t)R * 8, hipMemcpyHostToDevice, c->st)); }
	else CHK(arc_reindex(c));
	HIPCHK(hipGetLastError());
	c->graph_ready = true;
	return 0;
}

extern "C" int mahip_asg_del_trans_range(mahip_ctx_t *c, int fuzz, uint32_t v_beg, uint32_t v_end, uint32_t *n_reduced)
{ // marking only (asg.c:148-186) for the vertices [v_beg, v_end)
	HIPCHK(hipSetDevice(c->dev));
	if (!c->graph_ready) { mahip_set_error("mahip_asg_del_trans: no graph"); return -1; }
	uint32_t V = 2 * graph_nseq(c);
	if (v_end > V) v_end = V;
	CHK(ctr_zero(c));
	CHK(dev_reserve(c, c->ovf, ((size_t)V + 1) * 4));
	ArcCols a = arcs_of(c, c->ag);
	unsigned long long *ctr = P<unsigned long long>(c->ctr);
	if (v_end > v_beg && c->n_arc) {
		ProfScope ps(c, "k_asg_trans", 32.0 * (double)c->n_arc); // 16*(A+I)/A per arc, I ~ A on clean data
		static const bool old_small = getenv("MA_TRANS_OLD") != nullptr; // A/B handle: round 4's first tier (rows a vertex ahead, every other trip paid per vertex)
		if (old_small)
		hipLaunchKernelGGL((k_asg_trans<128, 256, true>), dim3(grid_for(((size_t)(v_end - v_beg) + 63) / 64, 4, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, (const uint32_t*)a.v, (const uint32_t*)a.len, a.ol,
		                   (const unsigned long long*)P<unsigned long long>(c->idx), (const uint8_t*)P<uint8_t>(c->sdel), v_beg, v_end, (uint32_t)fuzz, P<uint32_t>(c->ovf), ctr);
		else
		hipLaunchKernelGGL(k_asg_trans_pipe, dim3(grid_for(((size_t)(v_end - v_beg) + 63) / 64, 4, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, (const uint32_t*)a.v, (const uint32_t*)a.len, a.ol,
		                   (const unsigned long long*)P<unsigned long long>(c->idx), (const uint8_t*)P<uint8_t>(c->sdel), v_beg, v_end, (uint32_t)fuzz, ctr);
		hipLaunchKernelGGL((k_asg_trans<TR_CAP, TR_HASH, false>), dim3(grid_for(((size_t)(v_end - v_beg) + 63) / 64, 4, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, (const uint32_t*)a.v, (const uint32_t*)a.len, a.ol,
		                   (const unsigned long long*)P<unsigned long long>(c->idx), (const uint8_t*)P<uint8_t>(c->sdel), v_beg, v_end, (uint32_t)fuzz, P<uint32_t>(c->ovf), ctr);
	}
	CHK(ctr_fetch(c));
	c->tr_inner = c->h_ctr[CT_TRINNER];
	if (c->prof && c->n_arc) prof_patch_last(c, "k_asg_trans", 16.0 * ((double)c->n_arc + (double)c->tr_inner)); // 16 (A + I) with the I this launch really ran
	uint32_t n_ovf = (uint32_t)c->h_ctr[CT_OVF2];
	if (n_ovf) {
		unsigned nblk = n_ovf < 64 ? n_ovf : 64;
		CHK(dev_reserve(c, c->marks, (size_t)nblk * V + 16));
		HIPCHK(hipMemsetAsync(c->marks.p, 0, (size_t)nblk * V, c->st));
		ProfScope ps(c, "k_asg_trans_big", 0);
		hipLaunchKernelGGL(k_asg_trans_big, dim3(nblk), dim3(256), 0, c->st, (const uint32_t*)a.v, (const uint32_t*)a.len, a.ol,
		                   (const unsigned long long*)P<unsigned long long>(c->idx), V, (uint32_t)fuzz, (const uint32_t*)P<uint32_t>(c->ovf), n_ovf, P<uint8_t>(c->marks), ctr);
		CHK(ctr_fetch(c));
	}
	HIPCHK(hipGetLastError());
	if (n_reduced) *n_reduced = (uint32_t)c->h_ctr[CT_NRED];
	return 0;
}

extern "C" int mahip_asg_del_trans(mahip_ctx_t *c, int fuzz, uint32_t *n_reduced)
{
	uint32_t nr = 0;
	CHK(mahip_asg_del_trans_range(c, fuzz, 0, 2 * graph_nseq(c), &nr));
	if (n_reduced) *n_reduced = nr;
	if (nr) CHK(arc_cleanup(c, c->n_arc, 0, 0)); // asg.c:188-189
	return 0;
}

// asg.c:104-121 asg_arc_del_multi (+ asg_cleanup when it removed arcs)
extern "C" int mahip_asg_del_multi(mahip_ctx_t *c, uint32_t *n_multi)
{
	HIPCHK(hipSetDevice(c->dev));
	if (!c->graph_ready) { mahip_set_error("mahip_asg_del_multi: no graph"); return -1; }
	CHK(ctr_zero(c));
	if (c->n_arc) {
		ProfScope ps(c, "k_asg_multi", 16.0 * (double)c->n_arc);
		hipLaunchKernelGGL(k_asg_multi, dim3(grid_for(c->n_arc, 256, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, arcs_of(c, c->ag), (size_t)c->n_arc,
		                   (const unsigned long long*)P<unsigned long long>(c->idx), P<unsigned long long>(c->ctr));
	}
	CHK(ctr_fetch(c));
	const uint32_t nm = (uint32_t)c->h_ctr[CT_NMULTI];
	if (nm) CHK(arc_cleanup(c, c->n_arc, 0, 0));
	if (n_multi) *n_multi = nm;
	return 0;
}

// asg.c:124-138 asg_arc_del_asymm (+ asg_cleanup when it removed arcs)
extern "C" int mahip_asg_del_asymm(mahip_ctx_t *c, uint32_t *n_asymm)
{
	HIPCHK(hipSetDevice(c->dev));
	if (!c->graph_ready) { mahip_set_error("mahip_asg_del_asymm: no graph"); return -1; }
	CHK(ctr_zero(c));
	if (c->n_arc) {
		ProfScope ps(c, "k_asg_asymm", 32.0 * (double)c->n_arc);
		hipLaunchKernelGGL(k_asg_asymm, dim3(grid_for(c->n_arc, 256, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, arcs_of(c, c->ag), (size_t)c->n_arc,
		                   (const unsigned long long*)P<unsigned long long>(c->idx), P<unsigned long long>(c->ctr));
	}
	CHK(ctr_fetch(c));
	if (c->prof && c->n_arc) prof_patch_last(c, "k_asg_asymm", 16.0 * (double)c->n_arc + 16.0 * (double)c->h_ctr[CT_PROBED]); // 16 + 16 x list entries probed
	const uint32_t na = (uint32_t)c->h_ctr[CT_NASYMM];
	if (na) CHK(arc_cleanup(c, c->n_arc, 0, 0));
	if (n_asymm) *n_asymm = na;
	return 0;
}

extern "C" int mahip_asg_symm(mahip_ctx_t *c, uint32_t *n_multi, uint32_t *n_asymm)
{
	CHK(mahip_asg_del_multi(c, n_multi));
	return mahip_asg_del_asymm(c, n_asymm);
}

extern "C" int mahip_asg_del_short(mahip_ctx_t *c, float drop_ratio, uint32_t *n_short)
{
	HIPCHK(hipSetDevice(c->dev));
	if (!c->graph_ready) { mahip_set_error("mahip_asg_del_short: no graph"); return -1; }
	uint32_t V = 2 * graph_nseq(c);
	CHK(ctr_zero(c));
	if (V && c->n_arc) hipLaunchKernelGGL(k_asg_short, dim3(grid_for(V, 256, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, arcs_of(c, c->ag), (const unsigned long long*)P<unsigned long long>(c->idx), V, drop_ratio, P<unsigned long long>(c->ctr));
	CHK(ctr_fetch(c));
	uint32_t ns = (uint32_t)c->h_ctr[CT_NSHORT];
	if (n_short) *n_short = ns;
	if (ns) CHK(arc_cleanup(c, c->n_arc, 0, 0));
	return 0;
}

extern "C" uint32_t mahip_asg_n_arc(mahip_ctx_t *c) { return c->n_arc; }
// iterations of the loop at asg.c:169 the last reduction on this context ran (the reduction moves 16 (A + I) bytes)
extern "C" uint64_t mahip_asg_trans_inner(mahip_ctx_t *c) { return c->tr_inner; }

extern "C" int mahip_asg_download(mahip_ctx_t *c, asg_t *g)
{
	HIPCHK(hipSetDevice(c->dev));
	if (!c->graph_ready) { mahip_set_error("mahip_asg_download: no graph"); return -1; }
	size_t n = c->n_arc;
	uint32_t R = graph_nseq(c), Rn = c->has_map ? c->n_seq_new : R;
	const int32_t *map = graph_map(c);
	size_t off_seq = (n + 1) * 16, off_idx = off_seq + (((size_t)Rn + 4) * 4 + 15) / 16 * 16;
	CHK(dev_reserve(c, c->key[0], off_idx + (2 * (size_t)Rn + 2) * 8));
	char *stg = (char*)c->key[0].p;
	HIPCHK(hipMemsetAsync(stg + off_idx, 0, 2 * (size_t)Rn * 8 + 8, c->st));
	if (n) hipLaunchKernelGGL(k_arc_to_aos, dim3(grid_for(n, 256)), dim3(256), 0, c->st, arcs_of(c, c->ag), n, map, (asg_arc_t*)stg);
	if (R) hipLaunchKernelGGL(k_seq_to_aos, dim3(grid_for(R, 256)), dim3(256), 0, c->st, (const uint32_t*)P<uint32_t>(c->slen), (const uint8_t*)P<uint8_t>(c->sdel),
	                          (const unsigned long long*)P<unsigned long long>(c->idx), map, R, (uint32_t*)(stg + off_seq), (unsigned long long*)(stg + off_idx));
	HIPCHK(hipGetLastError());
	g->arc = (asg_arc_t*)malloc((n ? n : 1) * sizeof(asg_arc_t));
	g->seq = (asg_seq_t*)malloc(((size_t)Rn ? Rn : 1) * sizeof(asg_seq_t));
	g->idx = (uint64_t*)malloc((2 * (size_t)Rn + 1) * 8);
	if (!g->arc || !g->seq || !g->idx) { mahip_set_error("mahip_asg_download: out of host memory"); return -1; }
	if (n) HIPCHK(hipMemcpyAsync(g->arc, stg, n * 16, hipMemcpyDeviceToHost, c->st));
	if (Rn) HIPCHK(hipMemcpyAsync(g->seq, stg + off_seq, (size_t)Rn * 4, hipMemcpyDeviceToHost, c->st));
	if (Rn) HIPCHK(hipMemcpyAsync(g->idx, stg + off_idx, 2 * (size_t)Rn * 8, hipMemcpyDeviceToHost, c->st));
	HIPCHK(hipStreamSynchronize(c->st));
	g->n_arc = (uint32_t)n; g->m_arc = (uint32_t)(n ? n : 1);
	g->n_seq = Rn; g->m_seq = Rn ? Rn : 1;
	g->is_srt = 1;
	return 0;
}
