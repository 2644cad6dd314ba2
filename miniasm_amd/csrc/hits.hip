// hits.hip -- the hit ingest/filter passes of reference hit.c on SoA columns resident in HBM.
//
// Layout: 8 u32 columns [n_hits] (qid qs qe tn ts te ml|rev<<31 bl|dead<<31), grouped by query id after the
// sort, goff[n_seq+1] = first slot of each query group.  Passes never move records: a removed hit gets its
// dead bit set (the reference compacts in place after every pass; compaction here happens only at export),
// so group boundaries stay valid from the sort to ma_sg_gen.  sub slots are uint2 {s | del<<31, e} = the
// bits of ma_sub_t (reference miniasm.h:38-40).
#include "mahip_internal.hpp"

#define COL_QID 0
#define COL_QS 1
#define COL_QE 2
#define COL_TN 3
#define COL_TS 4
#define COL_TE 5
#define COL_ML 6
#define COL_BL 7
#define DEAD 0x80000000u

// ------------------------------------------------------------------------------------------------ sort
// ma_hit_sort (hit.c:12-22) orders the hits by qns = qid<<32 | qs.  Everything between the sort and ma_sg_gen only needs the hits of a
// read to be ADJACENT: ma_hit_sub sorts its own events, cut / flt / contained are per hit.  The order inside a read's group shows in two
// places only: a hit dump, and the push order of the (few) arcs when arcs with equal (u,len) exist.  So the resident layout is grouped by
// query id, input order inside a group -- a sort on the id bits alone (BASELINE configs[3]: 21 instead of 37 key bits, 3 digit passes
// instead of 5) -- and the two consumers re-establish the (qid, qs, input position) order for what they look at: hits_order_rank() for a
// dump, push_stable_order() (graph.hip) for the arcs.
//   key = qid << bi | input position   (8-byte elements, no value array; qid and position are 32-bit: always fits)
// keep[i] = hit belongs to this context's read range (sharded mode).
__global__ __launch_bounds__(256) void k_hit_keys(const ma_hit_t *__restrict__ h, size_t n, uint64_t *__restrict__ key,
                                                   uint32_t *__restrict__ val, uint32_t *__restrict__ keep, unsigned long long *__restrict__ ctr,
                                                   uint32_t q_beg, uint32_t q_end, int bi, int full)
{ // full: key = qns, val = input position (the walk of the reference's sort over the original keys)
	uint32_t cnt = 0;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
		uint64_t k = h[i].qns;
		uint32_t q = (uint32_t)(k >> 32);
		int in = q >= q_beg && q < q_end;
		cnt += in;
		if (full) key[i] = k, val[i] = (uint32_t)i;
		else key[i] = (uint64_t)q << bi | i;
		if (keep) keep[i] = in;
	}
	blk_add_u64(&ctr[CT_LIVE], cnt);
}

// The same for an unsharded context, one block per radix tile: the block also counts the first sort digit of its keys, which
// is exactly the per-tile histogram the first radix pass needs (saves one sweep over the keys).
__global__ __launch_bounds__(256) void k_hit_keys_tiled(const ma_hit_t *__restrict__ h, size_t n, uint64_t *__restrict__ key, int bi,
                                                         uint32_t *__restrict__ hist, unsigned nb, unsigned tile, int shift, unsigned mask)
{
	__shared__ uint32_t s_cnt[1 << RS_MAXBITS];
	for (unsigned d = threadIdx.x; d <= mask; d += 256) s_cnt[d] = 0;
	__syncthreads();
	const size_t base = (size_t)blockIdx.x * tile;
	for (unsigned it = 0; it < tile / 256; ++it) {
		const size_t i = base + (size_t)it * 256 + threadIdx.x;
		if (i < n) {
			const uint64_t kk = (uint64_t)(uint32_t)(h[i].qns >> 32) << bi | i;
			key[i] = kk;
			atomicAdd(&s_cnt[(unsigned)(kk >> shift) & mask], 1u);
		}
	}
	__syncthreads();
	for (unsigned d = threadIdx.x; d <= mask; d += 256) hist[(size_t)blockIdx.x * (mask + 1u) + d] = s_cnt[d]; // the tile's row (radix.hip: RsOffsets)
}

// largest query id / query start (only when the caller gave no hints: the per-symbol ma_hit_sort)
__global__ __launch_bounds__(256) void k_hit_bounds(const ma_hit_t *__restrict__ h, size_t n, unsigned long long *__restrict__ ctr)
{
	uint32_t mq = 0, ms = 0;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
		uint64_t k = h[i].qns;
		uint32_t q = (uint32_t)(k >> 32), qs = (uint32_t)k;
		mq = q > mq ? q : mq; ms = qs > ms ? qs : ms;
	}
	blk_max_u64(&ctr[CT_MAXQID], mq);
	blk_max_u64(&ctr[CT_MAXQS], ms);
}

__global__ __launch_bounds__(256) void k_key_compact(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin, size_t n,
                                                      const uint32_t *__restrict__ keep, const uint32_t *__restrict__ pos,
                                                      uint64_t *__restrict__ kout, uint32_t *__restrict__ vout)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n && keep[i]) { kout[pos[i]] = kin[i]; if (vin) vout[pos[i]] = vin[i]; }
}

// ---- sorting RUNS of records instead of records (round 5) ----
// ma_hit_read stores a PAF line's record and its mirror side by side (hit.c:87-98), and an overlapper lists a query's overlaps together: the records of one
// query's own lines stand in the input at stride 2 (stride 1 without mirrors, `-b`), a RUN of ~ 50 with one query id.  A run is one element of the sort:
//   key = qid << (bi + bl) | position of its first record << bl | (length - 1)
// -- the mirrored records, whose ids are spread over the whole dictionary, stay runs of one.  At BASELINE configs[3] that is 102 M keys through the three digit
// passes instead of 200 M.  The stable sort on the id bits keeps a read's runs in the order of their first positions = input order, as long as no two runs of one
// read interleave (two runs of one id at the two parities over the same stretch: consecutive records of ONE query id at stride 1 under a stride-2 reading, e.g. a
// row of self hits): k_runs_expand counts such pairs and the caller then sorts records as before.  k_runs_expand turns the sorted runs back into what every
// consumer wants -- sidx[slot] = input position of the record in each slot, goff[id] = first slot of a read -- so the gather reads 4 bytes per slot instead of an
// 8-byte key and does not write sidx itself.
// k_hit_keys_runs: a wave takes 1024 consecutive records (16 x 64), a run never crosses that border (cut there: 0.2 % more runs); heads by comparing with the
// record STRIDE before; lengths from the head masks, walking the 16 rounds backwards; the keys leave in input order at the tile's offset in the run sequence
// (chained look-back, mahip_internal.hpp: sc_look_back).
#define RUN_SLAB 1024u
#define RUN_TILE (4u * RUN_SLAB)
#define RUN_MIN_LEN_BITS 10 // a run has at most RUN_SLAB records
template <int STRIDE>
__global__ __launch_bounds__(256) void k_hit_keys_runs(const ma_hit_t *__restrict__ h, size_t n, uint64_t *__restrict__ key, int bi, int bl, unsigned long long *__restrict__ d_total,
                                                        unsigned long long *state, uint32_t *ticket, uint32_t ticket_base, uint32_t epoch, uint32_t n_seq, unsigned long long *__restrict__ d_bad)
{
	__shared__ uint32_t s_wave[4];
	__shared__ uint32_t s_tile, s_prefix;
	if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u) - ticket_base;
	__syncthreads();
	const uint32_t tile = s_tile;
	const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const size_t wbase = (size_t)tile * RUN_TILE + (size_t)wave * RUN_SLAB;
	const uint32_t nvalid = wbase >= n ? 0u : (uint32_t)(n - wbase < RUN_SLAB ? n - wbase : RUN_SLAB);
	uint32_t q[16];
	unsigned long long hm[16];
#pragma unroll
	for (int it = 0; it < 16; ++it) { const uint32_t p = (uint32_t)it * 64u + lane; q[it] = p < nvalid ? (uint32_t)(h[wbase + p].qns >> 32) : 0u; }
	{ // an id outside the dictionary (a caller's contract violation) would lose its high bits in the packed key and look like another read's: say so, the caller sorts records then
		uint32_t qmax = 0;
#pragma unroll
		for (int it = 0; it < 16; ++it) qmax = q[it] > qmax ? q[it] : qmax;
		if (wv_ballot(qmax >= n_seq) && lane == 0) atomicAdd(d_bad, 1ull);
	}
	uint32_t cnt = 0, pre[16]; // pre[it]: heads of the wave in front of round it
#pragma unroll
	for (int it = 0; it < 16; ++it) {
		const uint32_t p = (uint32_t)it * 64u + lane;
		uint32_t pred = __shfl_up(q[it], STRIDE, 64);
		if (it > 0) { const uint32_t carry = __shfl(q[it > 0 ? it - 1 : 0], (int)((lane + 64u - STRIDE) & 63u), 64); if (lane < (unsigned)STRIDE) pred = carry; }
		const bool has_pred = it > 0 || lane >= (unsigned)STRIDE;
		hm[it] = wv_ballot(p < nvalid && (!has_pred || pred != q[it]));
		pre[it] = cnt; cnt += (uint32_t)__popcll(hm[it]);
	}
	if (lane == 0) s_wave[wave] = cnt;
	__syncthreads();
	uint32_t wex = 0, tot = 0;
	for (unsigned w = 0; w < 4; ++w) { const uint32_t v = s_wave[w]; if (w < wave) wex += v; tot += v; }
	if (threadIdx.x == 0) {
		SC_PUBLISH(&state[tile], sc_pack(epoch, tile == 0 ? SC_INCL : SC_AGG, tot));
		if (tile == 0) s_prefix = 0;
	}
	if (tile > 0 && threadIdx.x < 64) {
		const uint32_t prefix = sc_look_back(state, tile, epoch, threadIdx.x);
		if (threadIdx.x == 0) { s_prefix = prefix; SC_PUBLISH(&state[tile], sc_pack(epoch, SC_INCL, prefix + tot)); }
	}
	__syncthreads();
	const uint32_t out0 = s_prefix + wex;
	if ((size_t)tile * RUN_TILE < n && (size_t)(tile + 1) * RUN_TILE >= n && threadIdx.x == 0) *d_total = (unsigned long long)s_prefix + tot; // the last tile
	// lengths, last round first: nxt[par] = the nearest head of the parity class behind the current round (or the slab's end, rounded up to the class)
	const unsigned par = STRIDE == 1 ? 0u : (lane & 1u);
	const unsigned long long cls = STRIDE == 1 ? ~0ull : (par ? 0xAAAAAAAAAAAAAAAAull : 0x5555555555555555ull);
	uint32_t nxt = STRIDE == 1 ? nvalid : nvalid + ((par - nvalid) & 1u); // (per lane: its own class)
#pragma unroll
	for (int it = 15; it >= 0; --it) {
		const uint32_t p = (uint32_t)it * 64u + lane;
		const unsigned long long same = hm[it] & cls, above = same & ~wv_le(lane);
		const uint32_t np = above ? (uint32_t)it * 64u + (uint32_t)(__ffsll((long long)above) - 1) : nxt;
		if (hm[it] >> lane & 1ull) {
			const uint32_t len = (np - p) / (uint32_t)STRIDE;
			key[out0 + pre[it] + (uint32_t)__popcll(hm[it] & wv_lt(lane))] = (uint64_t)q[it] << (bi + bl) | (uint64_t)(wbase + p) << bl | (uint64_t)(len - 1u);
		}
		if (same) nxt = (uint32_t)it * 64u + (uint32_t)(__ffsll((long long)same) - 1);
	}
}

// sorted runs -> sidx (input position of the record in every slot) + goff (first slot of every read with records; the others: radix_group_starts_finish).
// Three launches: the records of every GROUP of RX_GROUP x RX_TILE consecutive runs are counted (k_runs_count), the few thousand counts scanned, and k_runs_expand writes
// a group's stretch of slots.  Inside a group, run (tile kk, round j, thread t) = group * ... + kk * RX_TILE + j * 256 + t, so that the lanes of a wave hold consecutive
// runs -- coalesced key loads, and the single-record runs of neighbouring lanes (the mirrored half of the records) write neighbouring slots; the lengths are scanned in
// that order (wave scans + 32 partial sums).  A run of fewer than RX_LONG records is written by its own lane; a longer one (a query's own lines: ~ 50 records) by the
// whole wave, a record per lane.
// (Tried on the way, round 5: a thread per SLOT with a binary search over the runs' offsets in LDS -- 1.07 ms per 200 M records, eleven dependent LDS reads per slot;
// one launch with the tiles chained by look-back -- 0.88 ms: the ticket is ONE word every block increments, 12 - 17 ns per atomic; groups of tiles chained -- 0.78 ms,
// and 0.96 with a group's keys held in registers: every block of the launch publishes at the same moment and looks back over all the others at once.)
#ifndef RX_ITEMS
#define RX_ITEMS 4 // runs per thread and tile (8: 148 registers, three blocks per CU; 4: see tools/isa_stats.sh)
#endif
#define RX_TILE (256 * RX_ITEMS)
#define RX_LONG 16u
#define RX_GROUP (16384u / RX_TILE) // consecutive tiles per block: 16 k runs
__global__ __launch_bounds__(256) void k_runs_count(const uint64_t *__restrict__ rkey, uint32_t n_runs, int bl, uint32_t *__restrict__ cnt)
{
	__shared__ uint32_t s_wave[4];
	const uint32_t g00 = blockIdx.x * (RX_GROUP * RX_TILE);
	const uint64_t lmask = (1ull << bl) - 1ull;
	uint32_t mine = 0;
	for (uint32_t kk = 0; kk < RX_GROUP; kk += 2) { // 16 independent loads in flight
		uint64_t kk2[2 * RX_ITEMS];
#pragma unroll
		for (int j = 0; j < 2 * RX_ITEMS; ++j) { const uint32_t r = g00 + kk * RX_TILE + (uint32_t)j * 256u + threadIdx.x; kk2[j] = r < n_runs ? rkey[r] : ~0ull; }
#pragma unroll
		for (int j = 0; j < 2 * RX_ITEMS; ++j) mine += kk2[j] == ~0ull ? 0u : (uint32_t)(kk2[j] & lmask) + 1u; // (no key is all ones: the id field is < n_seq)
	}
	uint32_t gtot;
	(void)block_excl_scan_256(mine, s_wave, &gtot);
	if (threadIdx.x == 0) cnt[blockIdx.x] = gtot;
}
template <int STRIDE>
__global__ __launch_bounds__(256) void k_runs_expand(const uint64_t *__restrict__ rkey, uint32_t n_runs, int bi, int bl, uint32_t n_seq, uint32_t n_slots, const uint32_t *__restrict__ gpre,
                                                      uint32_t *__restrict__ sidx, uint32_t *__restrict__ goff, unsigned long long *__restrict__ ctr)
{
	__shared__ uint32_t s_part[RX_ITEMS][4];
	__shared__ uint64_t s_last[RX_ITEMS][4];
	const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t g00 = blockIdx.x * (RX_GROUP * RX_TILE);
	const uint64_t lmask = (1ull << bl) - 1ull, pmask = bi >= 64 ? ~0ull : (1ull << bi) - 1ull;
	uint32_t bad = 0;
	uint32_t S0 = gpre[blockIdx.x]; // first slot of the current tile
	for (uint32_t kk = 0; kk < RX_GROUP; ++kk) {
	const uint32_t r00 = g00 + kk * RX_TILE;
	if (r00 >= n_runs) break; // (uniform)
	uint64_t k[RX_ITEMS];
	uint32_t len[RX_ITEMS], incl[RX_ITEMS];
#pragma unroll
	for (int j = 0; j < RX_ITEMS; ++j) {
		const uint32_t r = r00 + (uint32_t)j * 256u + threadIdx.x;
		k[j] = r < n_runs ? rkey[r] : 0ull;
		len[j] = r < n_runs ? (uint32_t)(k[j] & lmask) + 1u : 0u;
	}
	const uint64_t kfirst = r00 > 0 && threadIdx.x == 0 ? rkey[r00 - 1] : 0ull; // the run in front of the tile
#pragma unroll
	for (int j = 0; j < RX_ITEMS; ++j) {
		const uint32_t x = (uint32_t)wv_scan_incl_i32((int)len[j], lane); // (DPP row shifts + row broadcasts: the wave is full; six ds_bpermute round trips per scan before)
		incl[j] = x;
		if (lane == 63) { s_part[j][wave] = x; s_last[j][wave] = k[j]; }
	}
	__syncthreads();
	uint32_t tot = 0, base[RX_ITEMS];
#pragma unroll
	for (int j = 0; j < RX_ITEMS; ++j)
		for (unsigned w = 0; w < 4; ++w) { if (w == wave) base[j] = tot; tot += s_part[j][w]; }
#pragma unroll
	for (int j = 0; j < RX_ITEMS; ++j) {
		const uint32_t r = r00 + (uint32_t)j * 256u + threadIdx.x;
		const uint32_t off = S0 + base[j] + incl[j] - len[j], pos = (uint32_t)(k[j] >> bl & pmask);
		// the run in front of this one: the lane to the left, the wave to the left, the round before, the tile before
		uint64_t kp = (uint64_t)__shfl_up((uint32_t)(k[j] >> 32), 1, 64) << 32 | __shfl_up((uint32_t)k[j], 1, 64);
		if (lane == 0) kp = wave > 0 ? s_last[j][wave - 1] : j > 0 ? s_last[j > 0 ? j - 1 : 0][3] : kfirst;
		if (r < n_runs) {
			const uint32_t qid = (uint32_t)(k[j] >> (bi + bl));
			if (r == 0 || (uint32_t)(kp >> (bi + bl)) != qid) { if (qid < n_seq) goff[qid] = off; else ++bad; } // ids are < n_seq by contract
			else bad += (k[j] >> bl & pmask) <= (kp >> bl & pmask) + (uint64_t)STRIDE * (kp & lmask); // the read's previous run must end in front of this one (no interleaving, see above)
			if (r == n_runs - 1) bad += off + len[j] != n_slots; // the slots must add up to the records
		}
		const bool is_long = len[j] >= RX_LONG;
		if (len[j] && !is_long)
			for (uint32_t x = 0; x < len[j]; ++x) sidx[off + x] = pos + (uint32_t)STRIDE * x;
		for (unsigned long long todo = wv_ballot(is_long); todo; todo &= todo - 1) { // the whole wave over one long run at a time
			const int src = __ffsll((long long)todo) - 1;
			const uint32_t o2 = __shfl(off, src, 64), p2 = __shfl(pos, src, 64), l2 = __shfl(len[j], src, 64);
			for (uint32_t x = lane; x < l2; x += 64) sidx[o2 + x] = p2 + (uint32_t)STRIDE * x;
		}
	}
	S0 += tot;
	__syncthreads(); // (s_part / s_last are the next tile's)
	}
	blk_add_u64(&ctr[CT_OVF2], bad);
}

struct HitCols { uint32_t *qid, *qs, *qe, *tn, *ts, *te, *ml, *bl; };

// AoS (input order) -> SoA (grouped by query id): one 32-byte record per lane as 2 x dwordx4 through the permutation held
// in the low bi bits of the sorted keys; group offsets from the query-id boundaries of the sorted keys.
// skey == nullptr: identity (input already grouped: the per-symbol path).
#ifndef GATHER_ILP
#define GATHER_ILP 2
#endif
// GATHER_ILP slots per thread: two independent request chains in flight per lane (the record fetch is a dependent chain: key -> record)
__global__ __launch_bounds__(256) void k_hit_gather(const ma_hit_t *__restrict__ h, const uint64_t *__restrict__ skey, int bi, size_t n, uint32_t n_seq,
                                                     HitCols c, uint32_t *__restrict__ goff, uint32_t *__restrict__ sidx, const uint32_t *__restrict__ spos = nullptr)
{ // sidx (optional): input position of the record in every slot -- how the order inside a group is re-established (hits_order_rank, push_stable_order)
  // spos (hits sorted as runs): the positions are there already (k_runs_expand), with the group offsets; skey, goff and sidx are then null
	size_t i[GATHER_ILP], j[GATHER_ILP];
	uint4 a[GATHER_ILP], b[GATHER_ILP];
	bool act[GATHER_ILP];
#pragma unroll
	for (int u = 0; u < GATHER_ILP; ++u) { // keys of all slots first
		i[u] = ((size_t)blockIdx.x * GATHER_ILP + u) * 256 + threadIdx.x;
		act[u] = i[u] < n; j[u] = i[u];
		if (act[u] && spos) j[u] = spos[i[u]];
		else if (act[u] && skey) j[u] = (size_t)(skey[i[u]] & ((1ull << bi) - 1));
	}
#pragma unroll
	for (int u = 0; u < GATHER_ILP; ++u)
		if (act[u]) { const uint4 *p = (const uint4*)(h + j[u]); a[u] = p[0]; b[u] = p[1]; } // a = {qs, qid, qe, tn}, b = {ts, te, ml|rev, bl|del}
#pragma unroll
	for (int u = 0; u < GATHER_ILP; ++u) {
		if (i[u] > n) continue;
		uint32_t q = n_seq, qprev = 0;
		const bool first = i[u] == 0;
		if (act[u]) {
			const size_t s = i[u];
			q = a[u].y;
			if (sidx) sidx[s] = (uint32_t)j[u];
			c.qid[s] = a[u].y; c.qs[s] = a[u].x; c.qe[s] = a[u].z; c.tn[s] = a[u].w;
			c.ts[s] = b[u].x; c.te[s] = b[u].y; c.ml[s] = b[u].z; c.bl[s] = b[u].w & ~DEAD;
		}
		if (!first && goff) qprev = skey ? (uint32_t)(skey[i[u] - 1] >> bi) : (uint32_t)(h[i[u] - 1].qns >> 32);
		// reads qprev+1 .. q start at slot i (reads without hits get empty groups)
		const uint32_t r0 = first ? 0 : qprev + 1;
		if (q > n_seq) q = n_seq;
		if (goff) for (uint32_t r = r0; r <= q && r <= n_seq; ++r) goff[r] = (uint32_t)i[u];
	}
}

// group offsets alone, from the sorted keys (the gather itself is left to the first coverage pass: k_hit_sub<false,*,true>)
// On a shard the reads below q_lo and above q_hi have no hits here: their (empty) groups are written by k_goff_outside, all lanes at once -- left to the
// thread that meets the first hit / the sentinel they were one serial loop over up to 7/8 of the reads (found by the round-3 projection: 17 - 29 ms per pass
// on every rank of a sharded run, however small its shard).
__global__ __launch_bounds__(256) void k_goff_outside(uint32_t *__restrict__ goff, uint32_t q_lo, uint32_t q_hi, uint32_t n_seq, uint32_t n)
{
	for (uint32_t r = blockIdx.x * 256u + threadIdx.x; r <= n_seq; r += gridDim.x * 256u)
		if (r < q_lo) goff[r] = 0u; else if (r > q_hi) goff[r] = n;
}
__global__ __launch_bounds__(256) void k_hit_goff(const uint64_t *__restrict__ skey, int bi, size_t n, uint32_t n_seq, uint32_t *__restrict__ goff, uint32_t q_lo, uint32_t q_hi)
{ // two slots per thread (one 16-byte load); slot n is the sentinel that closes the last groups.  Groups of the reads q_lo .. q_hi only (k_goff_outside: the rest)
	for (size_t base = (size_t)blockIdx.x * 512; base <= n; base += (size_t)gridDim.x * 512) {
		const size_t i0 = base + 2 * (size_t)threadIdx.x;
		uint32_t q0 = n_seq, q1 = n_seq;
		if (i0 + 1 < n) { const ulonglong2 kk = *(const ulonglong2*)(skey + i0); q0 = (uint32_t)(kk.x >> bi); q1 = (uint32_t)(kk.y >> bi); }
		else if (i0 < n) q0 = (uint32_t)(skey[i0] >> bi);
		if (q0 > n_seq) q0 = n_seq;
		if (q1 > n_seq) q1 = n_seq;
		uint32_t qprev = __shfl_up(q1, 1, 64); // the predecessor's id from the neighbouring lane; only the first lane of a wave loads it
		if ((threadIdx.x & 63) == 0 && i0 && i0 <= n) qprev = (uint32_t)(skey[i0 - 1] >> bi);
		if (i0 <= n) { // reads qprev+1 .. q start at slot i (reads without hits: empty groups)
			const uint32_t lo = i0 ? qprev + 1 : q_lo, hi = q0 < q_hi ? q0 : q_hi;
			for (uint32_t r = lo > q_lo ? lo : q_lo; r <= hi; ++r) goff[r] = (uint32_t)i0;
		}
		if (i0 + 1 <= n) { const uint32_t hi = q1 < q_hi ? q1 : q_hi; for (uint32_t r = q0 + 1 > q_lo ? q0 + 1 : q_lo; r <= hi; ++r) goff[r] = (uint32_t)(i0 + 1); }
	}
}

// ------------------------------------------------------------------------------------------------ ma_hit_sub
// reference hit.c:109-160: per query read, the first longest interval covered by >= min_dp hits.
//
// Tier R (almost every read): ONE WAVE PER READ, everything in registers.  Each lane loads up to ITEMS/2 hits
// (coalesced column loads), turns them into events (start<<1, end<<1|1), the wave sorts its 64*ITEMS events
// with a bitonic network whose cross-lane steps are wave shuffles (no LDS round trips, no barriers), then the
// depth sweep runs as one prefix scan over lanes plus a short in-register walk; the first-longest run is a
// wave max-reduce on (length, -position).  Reads with more than 512 hits go to tier B.
// Tier B: one 256-thread block per read, events in LDS (<= 8192) or in global scratch (any size).
// grid of the coverage kernels (blocks of 4 waves, one read per wave at a time); env MA_SUB_BLOCKS for experiments
static unsigned sub_blocks() { static unsigned v = 0; if (!v) { const char *e = getenv("MA_SUB_BLOCKS"); v = e ? (unsigned)atoi(e) : 2 * MA_STREAM_BLOCKS; /* 2 x the resident capacity: the dispatcher evens out the tail (measured 2048: 0.51, 4096: 0.46, 8192: 0.45 ms; more blocks = more end-of-block atomics) */ if (v < 1) v = 1; } return v; }
#ifndef MA_SUB_ORDER_DEFAULT
#define MA_SUB_ORDER_DEFAULT "012"
#endif
#define MA_SUB_BLOCKS sub_blocks()
// The three size classes of a coverage pass run side by side on three streams.  Which is QUEUED first matters: the first tier's blocks live for the whole launch (a wave walks
// ~ 120 reads) and fill every slot they are given, so classes queued behind it start when its blocks retire -- a tail (rocprofv3: all three kernels span the same 5 ms although
// the larger classes own a quarter of the hits).  MA_SUB_ORDER: "012" = first tier first (until round 5), "210" = the larger classes first.  MA_SUB_BIG_GRID: blocks of the two
// larger classes (0: as many as the first tier's).
struct SubOrder { int o[3]; };
static const int *sub_order()
{ // (a function-local static with an initialiser: built once, by one thread -- several host threads drive contexts of their own)
	static const SubOrder so = [] {
		SubOrder r;
		const char *e = getenv("MA_SUB_ORDER");
		const char *d = (e && strlen(e) == 3) ? e : MA_SUB_ORDER_DEFAULT;
		int seen = 0;
		for (int k = 0; k < 3; ++k) { r.o[k] = d[k] - '0'; if (r.o[k] < 0 || r.o[k] > 2) r.o[k] = k; seen |= 1 << r.o[k]; }
		if (seen != 7) { r.o[0] = 0; r.o[1] = 1; r.o[2] = 2; }
		return r;
	}();
	return so.o;
}
static unsigned sub_big_grid(unsigned g0) { static long v = -1; if (v < 0) { const char *e = getenv("MA_SUB_BIG_GRID"); v = e ? atol(e) : 0; } return v > 0 && (unsigned)v < g0 ? (unsigned)v : g0; }
// reads per bounds fetch in the larger tiers: up to SUB_CHUNK, fewer when there are not enough reads to give every wave of the grid a chunk
static uint32_t sub_chunk(uint32_t R) { uint64_t waves = 4ull * grid_for(R, 4, MA_SUB_BLOCKS), k = waves ? R / waves : 1; return (uint32_t)(k < 1 ? 1 : k > 16 ? 16 : k); }
#define EV_PAD 0xffffffffu
#define SUB_REG_MAX_HITS 512u
#define SUB_CHUNK 16u // most reads whose bounds a wave of the larger size classes fetches at once
#define SUB_LDS_EVENTS 8192u

// (MA_CE, MA_KEEP, lane_xor, wave_sort_regs: mahip_internal.hpp -- the arc sort of graph.hip uses the same network)

// Optional fusion (resident pipeline): while a hit sits in registers on its way into the coverage events of the
// SECOND pass, first apply ma_hit_cut against the first-pass intervals (hit.c:162-193) and ma_hit_flt
// (hit.c:195-216) -- the two streaming passes the reference runs between its two ma_hit_sub calls.
struct SubFuse {
	const uint2 *cut_sub; // intervals to cut against (and to classify with)
	int min_span, max_hang, min_ovlp;
	uint8_t *r_live;      // query groups that keep a hit after the filter (for the coverage estimate)
};
struct SubAcc { uint32_t n_cut, n_flt; uint64_t dp; }; // per-lane partial counters of the fused passes

// cut + filter of one live hit held in registers; returns 0 if the hit dies (its dead bit is written), else 1
__device__ __forceinline__ int fuse_cut_flt_v(const HitCols &c, uint32_t i, const SubFuse &f, uint32_t q, uint2 rq, uint32_t tn,
                                              uint32_t &qs, uint32_t &qe, uint32_t ml, uint32_t bl, SubAcc &acc, uint2 rt, uint32_t ts, uint32_t te)
{ // rt = the target's interval, ts/te = the hit's target columns: fetched by the caller (ahead of time in the SMALL kernels)
	int keep = 0;
	const uint32_t oqs = qs, oqe = qe, ots = ts, ote = te;
	if (!(rq.x & DEAD) && !(rt.x & DEAD) && mc_cut(&qs, &qe, &ts, &te, ml >> 31, (int32_t)rq.x, rq.y, (int32_t)rt.x, rt.y, f.min_span)) {
		mc_arc_t a;
		uint32_t ql = rq.y - (rq.x & 0x7fffffffu), tl = rt.y - (rt.x & 0x7fffffffu);
		++acc.n_cut;
		int r = mc_hit2arc(q, qs, qe, tn, ts, te, ml >> 31, (int)ql, (int)tl, f.max_hang, .5f, f.min_ovlp, &a);
		if (r >= 0 || r == MC_HT_QCONT || r == MC_HT_TCONT) {
			keep = 1; ++acc.n_flt;
			acc.dp += r >= 0 ? (uint32_t)r : r == MC_HT_QCONT ? ql : tl;
			f.r_live[q] = 1;
			if (qs != oqs) c.qs[i] = qs; // untouched columns are not written back
			if (qe != oqe) c.qe[i] = qe;
			if (ts != ots) c.ts[i] = ts;
			if (te != ote) c.te[i] = te;
		}
	}
	if (!keep) c.bl[i] = bl | DEAD;
	return keep;
}
__device__ __forceinline__ int fuse_cut_flt(const HitCols &c, uint32_t i, const SubFuse &f, uint32_t q, uint2 rq, uint32_t tn,
                                            uint32_t &qs, uint32_t &qe, uint32_t ml, uint32_t bl, SubAcc &acc)
{
	return fuse_cut_flt_v(c, i, f, q, rq, tn, qs, qe, ml, bl, acc, f.cut_sub[tn], c.ts[i], c.te[i]);
}

// the columns of up to 128 hits of one read, two slots per lane (hits lane and lane + 64), loaded ahead of their use: the
// SMALL kernels fetch the next read's hits before they work on the current one (the work is a long dependent chain per read,
// and 8 waves per SIMD do not hide an HBM round trip per read on their own)
struct SubPre { uint32_t beg, end; uint32_t bl[2], ml[2], qs[2], qe[2], tn[2], ts[2], te[2]; uint2 rt[2], rq; }; // ts/te, rt/rq: fused passes only

template <bool FUSE>
__device__ __forceinline__ void sub_preload(const HitCols &c, const uint32_t *__restrict__ goff, uint32_t q, unsigned lane, SubPre &p)
{
	p.beg = goff[q]; p.end = goff[q + 1];
	const uint32_t H = p.end - p.beg;
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		const uint32_t i = p.beg + h * 64 + lane;
		p.bl[h] = DEAD; p.ml[h] = p.qs[h] = p.qe[h] = p.tn[h] = p.ts[h] = p.te[h] = 0;
		if (H <= 128u && i < p.end) {
			p.bl[h] = c.bl[i]; p.ml[h] = c.ml[i]; p.qs[h] = c.qs[i]; p.qe[h] = c.qe[i]; p.tn[h] = c.tn[i];
			if (FUSE) { p.ts[h] = c.ts[i]; p.te[h] = c.te[i]; }
		}
	}
}

// Gather mode (first coverage pass of a context whose hits were just grouped by mahip_hits_sort): the wave that is about to sweep a read
// fetches the read's records itself -- through the permutation in the low bits of the sorted keys -- and writes the SoA columns on the way.
// The gather is a chain of dependent random fetches (memory latency), the sweep is a register sort (VALU): in one kernel the two overlap
// across the waves of a SIMD instead of adding up as two launches.
struct SubGather { const uint32_t *pos; uint32_t pstride, pmask; const ma_hit_t *aos; uint32_t *sidx; uint32_t n, chunk, q_lo; }; // pos[i * pstride] & pmask = input position of the record of slot i: the low words of the sorted keys (pstride 2, the position bits) or sidx itself (pstride 1, all bits: hits sorted as runs, k_runs_expand) // n = slots; chunk = reads per bounds fetch of the larger tiers; q_lo = first read to visit (a shard: its own range only; the kernel's n_seq argument is then the range's end)
struct GBounds { uint32_t beg, end; };                     // a read's slots
struct GKeys { uint32_t beg, end, j[2]; };                 // + low words of the sorted keys of its (up to 128) slots, two slots per lane
struct GRecs { uint32_t beg, end, j[2]; uint4 a[2], b[2]; }; // + input positions and records: a = {qs, qid, qe, tn}, b = {ts, te, ml|rev, bl|del}

__device__ __forceinline__ uint32_t gather_pos(const SubGather &g, size_t i)
{ // (little endian: the low word of key i holds the input position in its low bits)
	return g.pos[i * g.pstride] & g.pmask;
}
// The three stages of the pipelined chain.  All loads are UNCONDITIONAL (indices clamped into range, lanes without a slot fetch
// record 0): straight-line code lets the compiler count the outstanding loads (s_waitcnt vmcnt(N)) instead of draining the queue at
// every join, which is what keeps the next read's fetches in flight during the sweep of the current one.
__device__ __forceinline__ void gather_bounds(const uint32_t *__restrict__ goff, uint64_t q, uint32_t n_seq, GBounds &b)
{
	const uint32_t qc = q < n_seq ? (uint32_t)q : n_seq - 1;
	b.beg = goff[qc]; b.end = goff[qc + 1];
}
__device__ __forceinline__ void gather_keys(const SubGather &g, const GBounds &b, unsigned lane, GKeys &k)
{
	k.beg = b.beg; k.end = b.end;
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		const uint32_t i = b.beg + h * 64 + lane, ic = i < g.n ? i : g.n - 1;
		k.j[h] = g.pos[(size_t)ic * g.pstride];
	}
}
__device__ __forceinline__ void gather_recs(const SubGather &g, const GKeys &k, unsigned lane, GRecs &r)
{
	r.beg = k.beg; r.end = k.end;
	const bool mine = k.end - k.beg <= 128u;
	const uint32_t mask = g.pmask;
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		const uint32_t i = k.beg + h * 64 + lane;
		r.j[h] = (mine && i < k.end) ? (k.j[h] & mask) : 0u;
		const uint4 *p = (const uint4*)(g.aos + r.j[h]);
		r.a[h] = p[0]; r.b[h] = p[1];
	}
}
template <int GATHER>
__device__ __forceinline__ void gather_store(const HitCols &c, const SubGather &g, uint32_t i, uint32_t j, uint4 a, uint4 b)
{ // GATHER 1: positions from the sorted keys, sidx is written here; 2: positions read from sidx (k_runs_expand wrote it)
	c.qid[i] = a.y; c.qs[i] = a.x; c.qe[i] = a.z; c.tn[i] = a.w;
	c.ts[i] = b.x; c.te[i] = b.y; c.ml[i] = b.z; c.bl[i] = b.w & ~DEAD;
	if (GATHER == 1) g.sidx[i] = j;
}

// one read in registers; returns 1 if the read keeps an interval (value identical on all lanes).  pre != nullptr: the
// columns of the (at most 128) hits are already in registers
template <int ITEMS, bool FUSE, int GATHER = 0>
__device__ __forceinline__ uint32_t sub_group_regs(const HitCols &c, uint32_t q, uint32_t beg, uint32_t end, int min_dp, float min_iden,
                                                   int end_clip, uint2 *__restrict__ sub, unsigned lane, const SubFuse &f, SubAcc &acc, const SubPre *pre = nullptr,
                                                   const SubGather *g = nullptr)
{
	uint32_t x[ITEMS];
	int live_any = 0, ev_any = 0;
	uint2 rq = make_uint2(0, 0);
	if (FUSE) rq = pre ? pre->rq : f.cut_sub[q];
	if (GATHER && !pre) { // (the two larger size classes in gather mode) the records through the sorted keys, the columns written on the way.  Round 4: ALL keys of the read
		// first, then ALL records, then the stores -- four slots a lane at a time.  Round 3 went key -> record -> stores slot by slot: with stores between
		// them the compiler keeps the loads in program order, and a read of 256 hits was eight dependent trips to memory instead of two.
		constexpr int HB = 4; // slots per lane per batch: 4 x 9 registers
#pragma unroll
		for (int h0 = 0; h0 < ITEMS / 2; h0 += HB) {
			uint32_t jj[HB];
			uint4 aa[HB], bb[HB];
#pragma unroll
			for (int k = 0; k < HB; ++k) { const uint32_t i = beg + (h0 + k) * 64 + lane; jj[k] = (h0 + k < ITEMS / 2 && i < end) ? gather_pos(*g, i) : 0u; }
#pragma unroll
			for (int k = 0; k < HB; ++k) { const uint4 *p = (const uint4*)(g->aos + jj[k]); aa[k] = p[0]; bb[k] = p[1]; } // (lanes without a slot fetch record 0: no branch between the loads)
#pragma unroll
			for (int k = 0; k < HB; ++k) {
				const int h = h0 + k;
				const uint32_t i = beg + h * 64 + lane;
				if (h < ITEMS / 2) {
					x[2 * h] = x[2 * h + 1] = EV_PAD;
					if (i < end) {
						uint32_t es, ee;
						gather_store<GATHER>(c, *g, i, jj[k], aa[k], bb[k]);
						live_any = 1;
						if (mc_sub_ok(q, aa[k].x, aa[k].z, aa[k].w, (int32_t)(bb[k].z & 0x7fffffffu), (int32_t)(bb[k].w & ~DEAD), min_iden, end_clip, &es, &ee)) x[2 * h] = es, x[2 * h + 1] = ee, ev_any = 1;
					}
				}
			}
		}
	} else
#pragma unroll
	for (int h = 0; h < ITEMS / 2; ++h) {
		uint32_t i = beg + h * 64 + lane;
		x[2 * h] = x[2 * h + 1] = EV_PAD;
		if (i < end) {
			uint32_t bl, ml, qs, qe, tn, es, ee;
			int alive;
			if (pre && h < 2) {
				bl = pre->bl[h]; ml = pre->ml[h]; qs = pre->qs[h]; qe = pre->qe[h]; tn = pre->tn[h];
				alive = !(bl & DEAD) && (!FUSE || fuse_cut_flt_v(c, i, f, q, rq, tn, qs, qe, ml, bl, acc, pre->rt[h], pre->ts[h], pre->te[h]));
			} else if (GATHER) { // the record through the sorted key; the columns are written on the way
				const uint32_t j = gather_pos(*g, i);
				const uint4 *p = (const uint4*)(g->aos + j);
				const uint4 a = p[0], b = p[1];
				gather_store<GATHER>(c, *g, i, j, a, b);
				bl = b.w & ~DEAD; ml = b.z; qs = a.x; qe = a.z; tn = a.w;
				alive = 1;
			} else {
				bl = c.bl[i]; ml = c.ml[i]; qs = c.qs[i]; qe = c.qe[i]; tn = c.tn[i]; // independent loads
				alive = !(bl & DEAD) && (!FUSE || fuse_cut_flt(c, i, f, q, rq, tn, qs, qe, ml, bl, acc));
			}
			if (alive) {
				live_any = 1;
				if (mc_sub_ok(q, qs, qe, tn, (int32_t)(ml & 0x7fffffffu), (int32_t)bl, min_iden, end_clip, &es, &ee)) x[2 * h] = es, x[2 * h + 1] = ee, ev_any = 1;
			}
		}
	}
	if (wv_ballot(live_any) == 0) { if (lane == 0) sub[q] = make_uint2(0, 0); return 0; } // no surviving hit: not a group for the reference
	if (wv_ballot(ev_any) == 0) { if (lane == 0) sub[q] = make_uint2(DEAD, 0); return 0; } // hit.c:152
	wave_sort_regs<ITEMS>(x, lane);
	// depth before this lane's first event
	int net = 0;
#pragma unroll
	for (int r = 0; r < ITEMS; ++r) if (x[r] != EV_PAD) net += (x[r] & 1) ? -1 : 1;
	int dp = wv_scan_incl_i32(net, lane) - net;
	// walk the lane's events: runs opened and closed here are resolved at once; at most one run per lane was
	// opened in an earlier lane (its closing event is the lane's first down before any local up)
	const uint32_t NONE = 0xffffffffu;
	uint32_t cur_up = NONE, last_up = NONE, pend_pos = NONE, pend_idx = 0, best_s = 0;
	uint64_t best = 0;
#pragma unroll
	for (int r = 0; r < ITEMS; ++r) {
		uint32_t e = x[r];
		if (e == EV_PAD) continue;
		int old = dp;
		dp += (e & 1) ? -1 : 1;
		if (old < min_dp && dp >= min_dp) cur_up = last_up = e >> 1;
		else if (old >= min_dp && dp < min_dp) {
			if (cur_up != NONE) {
				uint64_t cand = (uint64_t)((e >> 1) - cur_up) << 32 | (0xffffffffu - (lane * ITEMS + r));
				if (cand > best) best = cand, best_s = cur_up;
				cur_up = NONE;
			} else pend_pos = e >> 1, pend_idx = lane * ITEMS + r;
		}
	}
	const uint32_t carry = wv_scan_last_u32(last_up, NONE, lane); // start of the run that is open at the end of each lane: "last defined" scan
	const uint32_t before = wv_prev_lane_u32(carry, NONE, lane);
	if (pend_pos != NONE && before != NONE) {
		uint64_t cand = (uint64_t)(pend_pos - before) << 32 | (0xffffffffu - pend_idx);
		if (cand > best) best = cand, best_s = before;
	}
	uint64_t gbest = wv_max_u64_full(best);
	uint32_t len = (uint32_t)(gbest >> 32);
	if (len == 0) { if (lane == 0) sub[q] = make_uint2(DEAD, 0); return 0; } // strict '>' against an empty best (hit.c:142,146)
	uint64_t own = wv_ballot(best == gbest);
	uint32_t s = wv_read_lane_u32(best_s, __ffsll((long long)own) - 1);
	if (lane == 0) sub[q] = make_uint2((s - (uint32_t)end_clip) & 0x7fffffffu, s + len + (uint32_t)end_clip);
	return 1;
}

// Three instantiations per fusion mode share the work by read size, so that each runs at the occupancy its register
// need allows: CLS 0 = reads with <= 128 hits (4 events per lane, 8 waves/SIMD, software-pipelined loads), CLS 1 = 129..256
// hits (16 events per lane), CLS 2 = 257..512 hits (32 events per lane); larger reads go to the block kernel (tier B).
// Workgroups go to the 8 XCDs round-robin (workgroup i runs on XCD i mod 8), and every XCD has an L2 of its own.  The first coverage pass fetches
// records through the sorted keys; the record of a line and its mirror share a 64-byte sector and belong to two reads that lie close together in
// id order, so reads that are neighbours should be swept on the SAME XCD: its L2 then serves the second half of the sector instead of fetching it
// again.  The block ids one XCD sees are therefore made consecutive: block (x, k) -> x * G/8 + k.
__device__ __forceinline__ unsigned sub_block_id()
{
	const unsigned b = blockIdx.x, g8 = gridDim.x & ~7u;
	return b < g8 ? (b & 7u) * (g8 >> 3) + (b >> 3) : b;
}
// Waves per SIMD the compiler has to fit (0 = its own choice): the gather tier 0 needs 82 VGPRs of its own accord (5 waves), 80 with the target (6);
// the fused tier 1 98 (4 waves) against 96 (5).  Round 3, all four changes together (these two, the XCD order here and in radix.hip, the size classes
// side by side): 20.18 -> 19.11 ms per pass on the same box, k_hit_sub<gather> 6.67 -> 5.98 ms, the fused pass 3.77 -> 3.42 ms (profiles/r03_experiments.txt).
#if defined(__HIP__) && defined(__clang__) /* (the CPU test build of this file is g++) */
#define SUB_WPE_ATTR(F, C, G) __attribute__((amdgpu_waves_per_eu(sub_wpe(F, C, G))))
#else
#define SUB_WPE_ATTR(F, C, G)
#endif
#ifndef SUB_WPE_G0
#define SUB_WPE_G0 7 // round 4, visit F (one box, base 6.60 / 6.70 ms): 5 waves 6.86, 6 waves (round 3) 6.6 - 6.7, 7 waves 6.42 ms although 18 registers spill
#endif
#ifndef SUB_WPE_F0
#define SUB_WPE_F0 6 // the fused first tier: the compiler's own choice (82 registers, 6 waves by count but no target) 3.26 ms, target 6: 3.11, target 8 (29 spills): 3.15
#endif
constexpr unsigned sub_wpe(bool fuse, int cls, int gather) { return gather && cls == 0 ? SUB_WPE_G0 : fuse && cls == 1 ? 5 : fuse && cls == 0 ? SUB_WPE_F0 : 0; }
template <bool FUSE, int CLS, int GATHER = 0>
__global__ __launch_bounds__(256) SUB_WPE_ATTR(FUSE, CLS, GATHER) void k_hit_sub(HitCols c, const uint32_t *__restrict__ goff, uint32_t n_seq,
                                                  int min_dp, float min_iden, int end_clip, uint2 *__restrict__ sub,
                                                  uint32_t *__restrict__ ovf, unsigned long long *__restrict__ ctr, SubFuse f, SubGather g)
{
	const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t n_kept = 0;
	SubAcc acc = {0, 0, 0};
	const uint32_t chunk = g.chunk ? g.chunk : 1u; // larger tiers: reads per bounds fetch (few reads: one per wave, so that the launch still spreads over the chip)
	if (CLS == 0 && GATHER) { // three-deep software pipeline over the chain bounds -> keys -> records: every load is issued a whole sweep before its
		// first use, and nothing in between waits for it
		const uint64_t stride = (uint64_t)gridDim.x * 4;
		uint64_t q = (uint64_t)g.q_lo + sub_block_id() * 4 + wave;
		GBounds bn;
		GKeys kn;
		GRecs cur, nxt;
		gather_bounds(goff, q, n_seq, bn); gather_keys(g, bn, lane, kn); gather_recs(g, kn, lane, cur);
		gather_bounds(goff, q + stride, n_seq, bn); gather_keys(g, bn, lane, kn);
		gather_bounds(goff, q + 2 * stride, n_seq, bn);
		while (q < n_seq) {
			const uint32_t beg = cur.beg, end = cur.end, H = end - beg;
			// (1) the columns of this read FIRST, and without a branch: the memory counter is in order, so a store issued after the next
			// read's loads would have to be acknowledged before those loads count as complete -- every sweep would start by waiting for
			// its predecessor's write-back.  Lanes without a slot write to the spare slots behind the arrays.  (Masking them with the bounds
			// check of a raw buffer descriptor instead was measured in round 3: 6.60 vs 6.80 ms alone, nothing when combined with the other
			// changes -- within the box's noise, dropped.)
#pragma unroll
			for (int h = 0; h < 2; ++h) {
				const uint32_t i = beg + h * 64 + lane;
				gather_store<GATHER>(c, g, (H <= 128u && i < end) ? i : g.n + lane, cur.j[h], cur.a[h], cur.b[h]);
			}
			__builtin_amdgcn_sched_barrier(0);
			// (2) the fetches of the reads behind it
			gather_recs(g, kn, lane, nxt);
			gather_keys(g, bn, lane, kn);
			gather_bounds(goff, q + 3 * stride, n_seq, bn);
			__builtin_amdgcn_sched_barrier(0);
			// (3) the sweep
			if (H == 0) { if (lane == 0) sub[q] = make_uint2(0, 0); } // never a query: calloc'ed zero (hit.c:115)
			else if (H <= 128) {
				SubPre pre;
				pre.beg = beg; pre.end = end;
#pragma unroll
				for (int h = 0; h < 2; ++h) {
					pre.bl[h] = cur.b[h].w & ~DEAD; pre.ml[h] = cur.b[h].z; pre.qs[h] = cur.a[h].x; pre.qe[h] = cur.a[h].z; pre.tn[h] = cur.a[h].w;
					pre.ts[h] = pre.te[h] = 0;
				}
				if (H <= 64) n_kept += sub_group_regs<2, false>(c, (uint32_t)q, beg, end, min_dp, min_iden, end_clip, sub, lane, f, acc, &pre);
				else n_kept += sub_group_regs<4, false>(c, (uint32_t)q, beg, end, min_dp, min_iden, end_clip, sub, lane, f, acc, &pre);
			}
			cur = nxt; q += stride;
		}
	} else
	if (CLS == 0) { // software pipeline: the hits of the wave's next read are in flight while the current read is processed
		const uint32_t stride = gridDim.x * 4;
		uint32_t q = g.q_lo + sub_block_id() * 4 + wave;
		SubPre cur, nxt;
		if (q < n_seq) sub_preload<FUSE>(c, goff, q, lane, cur);
		while (q < n_seq) {
			const uint32_t qn = q + stride;
			if (FUSE) { // this read's interval gathers first, the next read's columns behind them: the waits below are in issue order
				cur.rq = f.cut_sub[q];
#pragma unroll
				for (int h = 0; h < 2; ++h) cur.rt[h] = (cur.bl[h] & DEAD) ? make_uint2(DEAD, 0) : f.cut_sub[cur.tn[h]];
			}
			if (qn < n_seq) sub_preload<FUSE>(c, goff, qn, lane, nxt);
			const uint32_t beg = cur.beg, end = cur.end, H = end - beg;
			if (H == 0) { if (lane == 0) sub[q] = make_uint2(0, 0); } // never a query: calloc'ed zero (hit.c:115)
			else if (H <= 64) n_kept += sub_group_regs<2, FUSE>(c, q, beg, end, min_dp, min_iden, end_clip, sub, lane, f, acc, &cur);
			else if (H <= 128) n_kept += sub_group_regs<4, FUSE>(c, q, beg, end, min_dp, min_iden, end_clip, sub, lane, f, acc, &cur);
			cur = nxt; q = qn;
		}
	} else
	// a wave takes SUB_CHUNK consecutive reads at a time: their bounds come with one coalesced load, and only the reads of this instantiation's
	// size class are visited (a dependent load per read, most of them somebody else's, is pure latency)
	for (uint64_t qb = (uint64_t)g.q_lo + (uint64_t)(sub_block_id() * 4 + wave) * chunk; qb < n_seq; qb += (uint64_t)gridDim.x * 4 * chunk) {
	const bool in = lane < chunk && qb + lane < n_seq;
	const uint32_t beg_l = in ? goff[qb + lane] : 0, end_l = in ? goff[qb + lane + 1] : 0, H_l = end_l - beg_l;
	unsigned long long todo = wv_ballot(CLS == 1 ? (H_l > 128 && H_l <= 256) : H_l > 256);
	while (todo) {
		const int qbit = __ffsll((long long)todo) - 1;
		todo &= todo - 1;
		const uint32_t q = (uint32_t)qb + (uint32_t)qbit;
		const uint32_t beg = __shfl(beg_l, qbit, 64), end = __shfl(end_l, qbit, 64), H = end - beg;
		if (CLS == 1) {
			if (H > 128 && H <= 256) n_kept += sub_group_regs<8, FUSE, GATHER>(c, q, beg, end, min_dp, min_iden, end_clip, sub, lane, f, acc, nullptr, &g);
		} else {
			if (H > 256 && H <= SUB_REG_MAX_HITS) n_kept += sub_group_regs<16, FUSE, GATHER>(c, q, beg, end, min_dp, min_iden, end_clip, sub, lane, f, acc, nullptr, &g);
			else if (H > SUB_REG_MAX_HITS) { // tier B sweeps it from the columns: in gather mode they are written here
				if (GATHER)
					for (uint32_t i = beg + lane; i < end; i += 64) {
						const uint32_t j = gather_pos(g, i);
						const uint4 *p = (const uint4*)(g.aos + j);
						gather_store<GATHER>(c, g, i, j, p[0], p[1]);
					}
				if (lane == 0) { unsigned long long k = atomicAdd(&ctr[CT_OVF], 1ull); ovf[k] = q; }
			}
		}
	}
	}
	blk_add_u64(&ctr[CT_REMAIN], lane == 0 ? n_kept : 0);
	if (FUSE) { blk_add_u64(&ctr[CT_CUT], acc.n_cut); blk_add_u64(&ctr[CT_LIVE], acc.n_flt); blk_add_u64(&ctr[CT_TOTDP], acc.dp); }
}

// sweep over sorted events held in memory (tier B): ballot prefix counts per 64-event chunk, run starts by rank
__device__ __forceinline__ uint64_t sub_sweep(const uint32_t *ev, uint32_t *up, uint32_t nev, int min_dp, unsigned lane)
{
	const uint64_t lt = wv_lt(lane), le = wv_le(lane);
	int carry = 0;
	uint32_t nup = 0, ndown = 0;
	uint64_t best = 0;
	for (uint32_t base = 0; base < nev; base += 64) {
		uint32_t i = base + lane;
		int valid = i < nev;
		uint32_t e = valid ? ev[i] : 0;
		int is_end = e & 1;
		uint64_t ms = wv_ballot(valid && !is_end), me = wv_ballot(valid && is_end);
		int dp = carry + __popcll(ms & le) - __popcll(me & le);
		int old = dp - (is_end ? -1 : 1);
		int isup = valid && old < min_dp && dp >= min_dp;
		int isdn = valid && old >= min_dp && dp < min_dp;
		uint64_t mu = wv_ballot(isup), md = wv_ballot(isdn);
		uint32_t ur = nup + __popcll(mu & lt), dr = ndown + __popcll(md & lt);
		if (isup) up[ur] = e >> 1;
		wv_sync();
		if (isdn) {
			uint32_t len = (e >> 1) - up[dr];
			uint64_t cand = (uint64_t)len << 32 | (0xffffffffu - dr);
			best = cand > best ? cand : best;
		}
		carry += __popcll(ms) - __popcll(me);
		nup += __popcll(mu), ndown += __popcll(md);
		wv_sync();
	}
	return wv_max_u64(best);
}

// tier B: one 256-thread block per oversized read; events (and run starts) in LDS when they fit, else in global
// scratch (ev at 2*goff[q], up at goff[q]: disjoint per read by construction)
template <bool FUSE>
__global__ __launch_bounds__(256) void k_hit_sub_big(HitCols c, const uint32_t *__restrict__ goff, const uint32_t *__restrict__ ovf, const unsigned long long *__restrict__ n_ovf_dev,
                                                      int min_dp, float min_iden, int end_clip, uint2 *__restrict__ sub,
                                                      uint32_t *__restrict__ gev, uint32_t *__restrict__ gup, unsigned long long *__restrict__ ctr, SubFuse f)
{
	__shared__ uint32_t s_ev[SUB_LDS_EVENTS], s_up[SUB_LDS_EVENTS / 2];
	__shared__ uint32_t s_n, s_live;
	SubAcc acc = {0, 0, 0};
	const uint32_t n_ovf = (uint32_t)*n_ovf_dev; // the list the register tiers just wrote: read on the device, no host round trip in between
	for (uint32_t k = blockIdx.x; k < n_ovf; k += gridDim.x) {
		uint32_t q = ovf[k], beg = goff[q], end = goff[q + 1];
		const bool in_lds = 2 * (end - beg) <= SUB_LDS_EVENTS;
		uint32_t *ev = in_lds ? s_ev : gev + 2 * (size_t)beg, *up = in_lds ? s_up : gup + beg;
		if (threadIdx.x == 0) s_n = 0, s_live = 0;
		__syncthreads();
		uint2 rq = make_uint2(0, 0);
		if (FUSE) rq = f.cut_sub[q];
		for (uint32_t i = beg + threadIdx.x; i < end; i += 256) {
			uint32_t bl = c.bl[i], es, ee;
			if (bl & DEAD) continue;
			uint32_t qs = c.qs[i], qe = c.qe[i], tn = c.tn[i], ml = c.ml[i];
			if (FUSE && !fuse_cut_flt(c, i, f, q, rq, tn, qs, qe, ml, bl, acc)) continue;
			s_live = 1;
			if (mc_sub_ok(q, qs, qe, tn, (int32_t)(ml & 0x7fffffffu), (int32_t)bl, min_iden, end_clip, &es, &ee)) {
				uint32_t p = atomicAdd(&s_n, 2u); // any order: sorted next, equal values are indistinguishable
				ev[p] = es, ev[p + 1] = ee;
			}
		}
		__syncthreads();
		uint32_t nev = s_n, live = s_live;
		if (!live) { if (threadIdx.x == 0) sub[q] = make_uint2(0, 0); __syncthreads(); continue; }
		MA_BITONIC(uint32_t, ev, nev, threadIdx.x, 256, __syncthreads());
		if (threadIdx.x < 64) {
			uint64_t best = sub_sweep(ev, up, nev, min_dp, threadIdx.x);
			if (threadIdx.x == 0) {
				uint32_t len = (uint32_t)(best >> 32);
				if (len > 0) {
					uint32_t s = up[0xffffffffu - (uint32_t)best];
					sub[q] = make_uint2((s - (uint32_t)end_clip) & 0x7fffffffu, s + len + (uint32_t)end_clip);
					atomicAdd(&ctr[CT_REMAIN], 1ull);
				} else sub[q] = make_uint2(DEAD, 0);
			}
		}
		__syncthreads();
	}
	if (FUSE) { blk_add_u64(&ctr[CT_CUT], acc.n_cut); blk_add_u64(&ctr[CT_LIVE], acc.n_flt); blk_add_u64(&ctr[CT_TOTDP], acc.dp); }
}

// ------------------------------------------------------------------------------------------------ ma_hit_cut
// reference hit.c:162-193, one thread per hit; removed hits only get their dead bit
__global__ __launch_bounds__(256) void k_hit_cut(HitCols c, size_t n, const uint2 *__restrict__ sub, int min_span, unsigned long long *__restrict__ ctr)
{
	uint32_t n_keep = 0;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
		uint32_t bl = c.bl[i];
		if (bl & DEAD) continue;
		int keep = 0;
		uint2 rq = sub[c.qid[i]], rt = sub[c.tn[i]];
		if (!(rq.x & DEAD) && !(rt.x & DEAD)) {
			uint32_t qs = c.qs[i], qe = c.qe[i], ts = c.ts[i], te = c.te[i];
			keep = mc_cut(&qs, &qe, &ts, &te, c.ml[i] >> 31, (int32_t)rq.x, rq.y, (int32_t)rt.x, rt.y, min_span);
			if (keep) c.qs[i] = qs, c.qe[i] = qe, c.ts[i] = ts, c.te[i] = te;
		}
		if (!keep) c.bl[i] = bl | DEAD;
		n_keep += keep;
	}
	blk_add_u64(&ctr[CT_LIVE], n_keep);
}

// ------------------------------------------------------------------------------------------------ ma_hit_flt
// reference hit.c:195-216 (int_frac is the literal .5 there); r_live[q] marks query groups that keep a hit
__global__ __launch_bounds__(256) void k_hit_flt(HitCols c, size_t n, const uint2 *__restrict__ sub, int max_hang, int min_ovlp,
                                                  uint8_t *__restrict__ r_live, unsigned long long *__restrict__ ctr)
{
	uint32_t n_keep = 0;
	uint64_t dp = 0;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
		uint32_t bl = c.bl[i];
		if (bl & DEAD) continue;
		int keep = 0;
		uint32_t q = c.qid[i], t = c.tn[i];
		uint2 sq = sub[q], st = sub[t];
		if (!(sq.x & DEAD) && !(st.x & DEAD)) {
			mc_arc_t a;
			uint32_t ql = sq.y - (sq.x & 0x7fffffffu), tl = st.y - (st.x & 0x7fffffffu);
			int r = mc_hit2arc(q, c.qs[i], c.qe[i], t, c.ts[i], c.te[i], c.ml[i] >> 31, (int)ql, (int)tl, max_hang, .5f, min_ovlp, &a);
			if (r >= 0 || r == MC_HT_QCONT || r == MC_HT_TCONT) {
				keep = 1;
				dp += r >= 0 ? (uint32_t)r : r == MC_HT_QCONT ? ql : tl;
				r_live[q] = 1;
			}
		}
		if (!keep) c.bl[i] = bl | DEAD;
		n_keep += keep;
	}
	blk_add_u64(&ctr[CT_LIVE], n_keep);
	blk_add_u64(&ctr[CT_TOTDP], dp);
}

__global__ __launch_bounds__(256) void k_flt_totlen(const uint2 *__restrict__ sub, const uint8_t *__restrict__ r_live, uint32_t n_seq, unsigned long long *__restrict__ ctr)
{
	uint64_t x = 0;
	for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < n_seq; r += gridDim.x * 256)
		if (r_live[r]) { uint2 s = sub[r]; x += (uint32_t)(s.y - (s.x & 0x7fffffffu)); }
	blk_add_u64(&ctr[CT_TOTLEN], x);
}

// ------------------------------------------------------------------------------------------------ ma_sub_merge
// reference hit.c:218-223: a.e = a.s + b.e ; a.s += b.s  (a.del kept, b.del ignored; s is a 31-bit field)
__global__ __launch_bounds__(256) void k_sub_merge(uint2 *__restrict__ a, const uint2 *__restrict__ b, uint32_t n_seq)
{
	uint32_t r = blockIdx.x * 256 + threadIdx.x;
	if (r < n_seq) {
		uint2 x = a[r], y = b[r];
		uint32_t as = x.x & 0x7fffffffu;
		a[r] = make_uint2((x.x & DEAD) | ((as + (y.x & 0x7fffffffu)) & 0x7fffffffu), as + y.y);
	}
}

// ------------------------------------------------------------------------------------------------ ma_hit_contained
// reference hit.c:225-256.  Pass 1 (per hit): classify with the final thresholds, flag contained reads,
// flag reads touched by any hit (hit.c:24-36).  Pass 2 (per read): del = sub.del | contained | seq.del | unused,
// keep flags for the squeeze scan (sdict.c:69-86).  Pass 3 (per hit): drop hits that lost an endpoint.
__global__ __launch_bounds__(256) void k_hit_contained(HitCols c, size_t n, const uint2 *__restrict__ sub, int max_hang, float int_frac, int min_ovlp,
                                                        uint8_t *__restrict__ r_cont, uint8_t *__restrict__ r_used)
{
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
		if (c.bl[i] & DEAD) continue;
		uint32_t q = c.qid[i], t = c.tn[i];
		uint2 sq = sub[q], st = sub[t];
		mc_arc_t a;
		int r = mc_hit2arc(q, c.qs[i], c.qe[i], t, c.ts[i], c.te[i], c.ml[i] >> 31, (int)(sq.y - (sq.x & 0x7fffffffu)),
		                   (int)(st.y - (st.x & 0x7fffffffu)), max_hang, int_frac, min_ovlp, &a);
		if (r == MC_HT_QCONT) r_cont[q] = 1;
		else if (r == MC_HT_TCONT) r_cont[t] = 1;
		r_used[q] = 1; r_used[t] = 1;
	}
}

// resident pipeline: the second ma_hit_cut (against cut_sub) and the flag pass of ma_hit_contained (against the merged
// intervals cls_sub) in one sweep over the hits
// (cut_sub[r], cls_sub[r]) side by side: the kernel below looks both up for the query AND the target of every hit; as one 16-byte entry a look-up is one request instead of two
__global__ __launch_bounds__(256) void k_sub_pair(const uint2 *__restrict__ cut_sub, const uint2 *__restrict__ cls_sub, uint32_t n_seq, uint4 *__restrict__ pair)
{
	const uint32_t r = blockIdx.x * 256 + threadIdx.x;
	if (r < n_seq) { const uint2 a = cut_sub[r], b = cls_sub[r]; pair[r] = make_uint4(a.x, a.y, b.x, b.y); }
}
__global__ __launch_bounds__(256) void k_hit_cut_contained(HitCols c, size_t n, const uint4 *__restrict__ sub2 /* {cut_sub, cls_sub} per read */, int min_span,
                                                            int max_hang, float int_frac, int min_ovlp,
                                                            uint8_t *__restrict__ r_cont, uint8_t *__restrict__ r_used, unsigned long long *__restrict__ ctr)
{
	uint32_t n_keep = 0;
	// CC_UNROLL hits per thread and trip: a hit is two dependent trips to memory (its columns, then the two reads' entries); one hit per thread and trip left the launch
	// latency-bound at a third of the chip's occupancy-bandwidth product.  All columns of the trip's hits are asked for first, then all entries.
	constexpr int CC_UNROLL = 4;
	for (size_t base = (size_t)blockIdx.x * (256 * CC_UNROLL); base < n; base += (size_t)gridDim.x * (256 * CC_UNROLL)) {
		uint32_t bl[CC_UNROLL], q[CC_UNROLL], t[CC_UNROLL], ml[CC_UNROLL], qs[CC_UNROLL], qe[CC_UNROLL], ts[CC_UNROLL], te[CC_UNROLL];
		uint4 pq[CC_UNROLL], pt[CC_UNROLL];
#pragma unroll
		for (int u = 0; u < CC_UNROLL; ++u) {
			const size_t i = base + (size_t)u * 256 + threadIdx.x;
			bl[u] = i < n ? c.bl[i] : DEAD;
			const size_t ic = i < n ? i : 0; // (clamped: the loads below are unconditional)
			q[u] = c.qid[ic]; t[u] = c.tn[ic]; ml[u] = c.ml[ic]; qs[u] = c.qs[ic]; qe[u] = c.qe[ic]; ts[u] = c.ts[ic]; te[u] = c.te[ic];
		}
#pragma unroll
		for (int u = 0; u < CC_UNROLL; ++u) { const bool dead = (bl[u] & DEAD) != 0; pq[u] = sub2[dead ? 0u : q[u]]; pt[u] = sub2[dead ? 0u : t[u]]; } // (a dead slot's entries are not used: entry 0, whatever its columns hold)
#pragma unroll
		for (int u = 0; u < CC_UNROLL; ++u) {
			const size_t i = base + (size_t)u * 256 + threadIdx.x;
			if (bl[u] & DEAD) continue;
			const uint2 rq = make_uint2(pq[u].x, pq[u].y), rt = make_uint2(pt[u].x, pt[u].y);
			uint32_t qs_ = qs[u], qe_ = qe[u], ts_ = ts[u], te_ = te[u];
			if (!(rq.x & DEAD) && !(rt.x & DEAD) && mc_cut(&qs_, &qe_, &ts_, &te_, ml[u] >> 31, (int32_t)rq.x, rq.y, (int32_t)rt.x, rt.y, min_span)) {
				const uint2 sq = make_uint2(pq[u].z, pq[u].w), st = make_uint2(pt[u].z, pt[u].w);
				mc_arc_t a;
				if (qs_ != qs[u]) c.qs[i] = qs_; // most hits lie inside both intervals: untouched columns are not written back
				if (qe_ != qe[u]) c.qe[i] = qe_;
				if (ts_ != ts[u]) c.ts[i] = ts_;
				if (te_ != te[u]) c.te[i] = te_;
				++n_keep;
				int r = mc_hit2arc(q[u], qs_, qe_, t[u], ts_, te_, ml[u] >> 31, (int)(sq.y - (sq.x & 0x7fffffffu)), (int)(st.y - (st.x & 0x7fffffffu)), max_hang, int_frac, min_ovlp, &a);
				if (r == MC_HT_QCONT) r_cont[q[u]] = 1;
				else if (r == MC_HT_TCONT) r_cont[t[u]] = 1;
				r_used[q[u]] = 1; r_used[t[u]] = 1;
			} else c.bl[i] = bl[u] | DEAD;
		}
	}
	blk_add_u64(&ctr[CT_LIVE], n_keep);
}

__global__ __launch_bounds__(256) void k_read_del(const uint2 *__restrict__ sub, const uint8_t *__restrict__ r_cont, const uint8_t *__restrict__ r_used,
                                                   uint8_t *__restrict__ r_del, uint32_t *__restrict__ keep, uint32_t n_seq)
{
	uint32_t r = blockIdx.x * 256 + threadIdx.x;
	if (r < n_seq) {
		int del = (sub[r].x >> 31) | r_cont[r] | r_del[r] | !r_used[r]; // r_del holds the caller's d->seq[].del on entry
		r_del[r] = (uint8_t)del;
		keep[r] = !del;
	}
}

// map[r] = -1 for dropped reads; surv[new id] = old id for the survivors (sdict.c:75-81 seen from both sides)
__global__ __launch_bounds__(256) void k_map_fix(int32_t *__restrict__ map, const uint8_t *__restrict__ r_del, uint32_t n_seq, uint32_t *__restrict__ surv)
{
	uint32_t r = blockIdx.x * 256 + threadIdx.x;
	if (r < n_seq) {
		if (r_del[r]) map[r] = -1;
		else surv[map[r]] = r;
	}
}

__global__ __launch_bounds__(256) void k_hit_squeeze(HitCols c, size_t n, const uint8_t *__restrict__ r_del, unsigned long long *__restrict__ ctr)
{
	uint32_t n_keep = 0;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
		uint32_t bl = c.bl[i];
		if (bl & DEAD) continue;
		int keep = !r_del[c.qid[i]] && !r_del[c.tn[i]];
		if (!keep) c.bl[i] = bl | DEAD;
		n_keep += keep;
	}
	blk_add_u64(&ctr[CT_LIVE], n_keep);
}

// ------------------------------------------------------------------------------------------------ export
// rank (optional): position of slot i in the order the records are exported in (the reference's order of tied hits)
__global__ __launch_bounds__(256) void k_hit_keepflags(const uint32_t *__restrict__ bl, size_t n, uint32_t *__restrict__ keep, const uint32_t *__restrict__ rank)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) keep[rank ? rank[i] : i] = !(bl[i] & DEAD);
}

// live hits -> dense AoS in array order, ids renumbered through map (if any)
__global__ __launch_bounds__(256) void k_hit_export(HitCols c, size_t n, const uint32_t *__restrict__ pos, const int32_t *__restrict__ map,
                                                     ma_hit_t *__restrict__ out, const uint32_t *__restrict__ rank)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	uint32_t bl = c.bl[i];
	if (bl & DEAD) return;
	uint32_t q = c.qid[i], t = c.tn[i];
	if (map) q = (uint32_t)map[q], t = (uint32_t)map[t];
	uint4 *o = (uint4*)(out + pos[rank ? rank[i] : i]);
	o[0] = make_uint4(c.qs[i], q, c.qe[i], t);
	o[1] = make_uint4(c.ts[i], c.te[i], c.ml[i], bl);
}

__global__ __launch_bounds__(256) void k_sub_squeeze(const uint2 *__restrict__ sub, const int32_t *__restrict__ map, uint32_t n_seq, uint2 *__restrict__ out)
{
	uint32_t r = blockIdx.x * 256 + threadIdx.x;
	if (r < n_seq && map[r] >= 0) out[map[r]] = sub[r];
}

// ================================================================================================ host side

static int bitlen(uint64_t x) { int b = 0; while (x) ++b, x >>= 1; return b; }

static HitCols cols_of(mahip_ctx *c)
{
	HitCols h;
	h.qid = P<uint32_t>(c->col[0]); h.qs = P<uint32_t>(c->col[1]); h.qe = P<uint32_t>(c->col[2]); h.tn = P<uint32_t>(c->col[3]);
	h.ts = P<uint32_t>(c->col[4]); h.te = P<uint32_t>(c->col[5]); h.ml = P<uint32_t>(c->col[6]); h.bl = P<uint32_t>(c->col[7]);
	return h;
}

static int reserve_read_arrays(mahip_ctx *c)
{
	size_t R = c->n_seq;
	CHK(dev_reserve(c, c->goff, (R + 2) * 4));
	for (int k = 0; k < 2; ++k) CHK(dev_reserve(c, c->sub[k], (R + 1) * 8));
	CHK(dev_reserve(c, c->r_cont, R + 16)); CHK(dev_reserve(c, c->r_used, R + 16));
	CHK(dev_reserve(c, c->r_del, R + 16)); CHK(dev_reserve(c, c->r_live, R + 16));
	CHK(dev_reserve(c, c->map, (R + 1) * 4));
	CHK(dev_reserve(c, c->surv, (R + 1) * 4));
	return 0;
}

static int hits_common_setup(mahip_ctx *c, size_t n, uint32_t n_seq)
{
	c->n_hits = n; c->n_in = n; c->n_live = n; c->n_seq = n_seq; c->n_seq_new = n_seq;
	c->soa_ready = false; c->has_map = false; c->surv_ready = false; c->graph_ready = false; c->gather_pending = false;
	c->hint_max_qs = 0; c->run_stride = 0; c->gk_runs = false; c->n_runs = 0; // hints describe one upload: set them again after every upload/adopt
	c->lazy_squeeze = false;
	c->sorted_here = false; c->hrank_ready = false; c->orank_ready = false;
	c->n_total = 0; // positions describe one upload, like the hints
	c->shard_bounds.clear(); // read ranges describe one upload, like the hints and the positions: a caller that shards by its own table says so again (mahip_set_shard_bounds / mahip_hits_balance)
	                         // after every upload / adopt.  (Until round 4 a table survived a new input with the same number of reads: a second input of equal read count silently got the stale ranges.)
	memset(&c->tie, 0, sizeof(c->tie));
	CHK(reserve_read_arrays(c));
	for (int k = 0; k < 8; ++k) CHK(dev_reserve(c, c->col[k], (n + 128) * 4)); // + spare slots (k_hit_sub gather mode: lanes without a slot)
	return 0;
}

extern "C" int mahip_hits_upload(mahip_ctx_t *c, const ma_hit_t *h, size_t n, uint32_t n_seq)
{
	HIPCHK(hipSetDevice(c->dev));
	CHK(hits_common_setup(c, n, n_seq));
	CHK(dev_reserve(c, c->aos_own, (n + 1) * sizeof(ma_hit_t)));
	CHK(xfer_copy(c, c->aos_own.p, (void*)h, n * sizeof(ma_hit_t), 1));
	c->d_aos = (const ma_hit_t*)c->aos_own.p;
	return 0;
}

extern "C" int mahip_hits_adopt(mahip_ctx_t *c, const void *d_hits, size_t n, uint32_t n_seq)
{
	HIPCHK(hipSetDevice(c->dev));
	CHK(hits_common_setup(c, n, n_seq));
	c->d_aos = (const ma_hit_t*)d_hits;
	return 0;
}

// the unsorted records as they sit in the context (after an upload / adopt / device-side parse), for tests
extern "C" int mahip_hits_raw_download(mahip_ctx_t *c, ma_hit_t *out)
{
	HIPCHK(hipSetDevice(c->dev));
	if (c->n_hits == 0) return 0;
	if (!c->d_aos) { mahip_set_error("mahip_hits_raw_download: no records"); return -1; }
	return xfer_copy(c, (void*)c->d_aos, out, c->n_hits * sizeof(ma_hit_t), 0);
}

__global__ __launch_bounds__(256) void k_rec_keep(const ma_hit_t *__restrict__ h, size_t n, uint32_t q_beg, uint32_t q_end, uint32_t *__restrict__ keep)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) { uint32_t q = (uint32_t)(h[i].qns >> 32); keep[i] = q >= q_beg && q < q_end; }
}
__global__ __launch_bounds__(256) void k_rec_compact(const ma_hit_t *__restrict__ h, size_t n, const uint32_t *__restrict__ keep, const uint32_t *__restrict__ pos, ma_hit_t *__restrict__ out)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n && keep[i]) { const uint4 *p = (const uint4*)(h + i); uint4 *o = (uint4*)(out + pos[i]); o[0] = p[0]; o[1] = p[1]; }
}

__global__ __launch_bounds__(256) void k_rec_positions(size_t n, const uint32_t *__restrict__ keep, const uint32_t *__restrict__ pos, uint32_t *__restrict__ out)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n && keep[i]) out[pos[i]] = (uint32_t)i;
}

extern "C" int mahip_hits_raw_extract_pos(mahip_ctx_t *c, uint32_t q_beg, uint32_t q_end, void *d_dst, uint32_t *d_pos, size_t *n_out);
extern "C" int mahip_hits_raw_extract(mahip_ctx_t *c, uint32_t q_beg, uint32_t q_end, void *d_dst, size_t *n_out) { return mahip_hits_raw_extract_pos(c, q_beg, q_end, d_dst, nullptr, n_out); }
// d_pos (optional, device): where each extracted record stood in this context's input -- what mahip_hits_set_positions of the shard's own context wants
extern "C" int mahip_hits_raw_extract_pos(mahip_ctx_t *c, uint32_t q_beg, uint32_t q_end, void *d_dst, uint32_t *d_pos, size_t *n_out)
{
	HIPCHK(hipSetDevice(c->dev));
	const size_t n = c->n_hits;
	if (n_out) *n_out = 0;
	if (n == 0) return 0;
	if (!c->d_aos) { mahip_set_error("mahip_hits_raw_extract: no records"); return -1; }
	CHK(dev_reserve(c, c->keep, (n + 16) * 4)); CHK(dev_reserve(c, c->pos, (n + 16) * 4));
	uint32_t *d_tot = (uint32_t*)(P<unsigned long long>(c->ctr) + CT_TOTAL);
	hipLaunchKernelGGL(k_rec_keep, dim3(grid_for(n, 256)), dim3(256), 0, c->st, c->d_aos, n, q_beg, q_end, P<uint32_t>(c->keep));
	CHK(scan_exclusive_u32(c, P<uint32_t>(c->keep), P<uint32_t>(c->pos), n, d_tot));
	if (d_dst) hipLaunchKernelGGL(k_rec_compact, dim3(grid_for(n, 256)), dim3(256), 0, c->st, c->d_aos, n, (const uint32_t*)P<uint32_t>(c->keep), (const uint32_t*)P<uint32_t>(c->pos), (ma_hit_t*)d_dst);
	if (d_pos) hipLaunchKernelGGL(k_rec_positions, dim3(grid_for(n, 256)), dim3(256), 0, c->st, n, (const uint32_t*)P<uint32_t>(c->keep), (const uint32_t*)P<uint32_t>(c->pos), d_pos);
	CHK(ctr_fetch(c));
	HIPCHK(hipGetLastError());
	if (n_out) *n_out = (size_t)(c->h_ctr[CT_TOTAL] & 0xffffffffu);
	return 0;
}

// ---- read ranges that hold equally many HITS (the sharded mode's unit of work), instead of equally many reads ----
__global__ __launch_bounds__(256) void k_qid_count(const ma_hit_t *__restrict__ h, size_t n, uint32_t n_seq, uint32_t *__restrict__ cnt)
{
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
		const uint32_t q = (uint32_t)(h[i].qns >> 32);
		if (q < n_seq) atomicAdd(&cnt[q], 1u);
	}
}
// bounds[0..world]: rank r owns the reads [bounds[r], bounds[r+1]); computed from the unsorted records in the context (every rank that holds the whole
// input computes the same table).  The table is kept in the context (mahip_shard_bounds) for the orchestrator.
extern "C" int mahip_hits_balance(mahip_ctx_t *c, int world, uint32_t *bounds)
{
	HIPCHK(hipSetDevice(c->dev));
	const uint32_t R = c->n_seq;
	const size_t n = c->n_hits;
	if (world < 1 || world > 1024) { mahip_set_error("mahip_hits_balance: bad world size"); return -1; }
	std::vector<uint32_t> b((size_t)world + 1, R);
	b[0] = 0;
	if (R && n && c->d_aos && world > 1) {
		CHK(dev_reserve(c, c->keep, ((size_t)R + 16) * 4));
		HIPCHK(hipMemsetAsync(c->keep.p, 0, (size_t)R * 4, c->st));
		hipLaunchKernelGGL(k_qid_count, dim3(grid_for(n, 256, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, c->d_aos, n, R, P<uint32_t>(c->keep));
		std::vector<uint32_t> cnt(R);
		HIPCHK(hipMemcpyAsync(cnt.data(), c->keep.p, (size_t)R * 4, hipMemcpyDeviceToHost, c->st));
		HIPCHK(hipStreamSynchronize(c->st));
		unsigned long long run = 0;
		int r = 1;
		for (uint32_t q = 0; q < R && r < world; ++q) { // rank r starts at the first read behind which r/world of the hits lie
			run += cnt[q];
			while (r < world && run * (unsigned long long)world >= (unsigned long long)n * (unsigned long long)r) b[r++] = q + 1;
		}
	} else if (world > 1) { // nothing to weigh: equal read counts
		const uint32_t per = (uint32_t)(((uint64_t)R + world - 1) / world);
		for (int r = 1; r < world; ++r) b[r] = (uint64_t)r * per < R ? (uint32_t)((uint64_t)r * per) : R;
	}
	c->shard_bounds = b;
	if (bounds) memcpy(bounds, b.data(), b.size() * 4);
	return 0;
}
// ---- sharded ingest: the records of a rank's own LINES to the ranks that own their QUERY reads ("one all-to-all of 32-byte hits") ----
__global__ __launch_bounds__(256) void k_add_u32(uint32_t *__restrict__ x, size_t n, uint32_t a)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) x[i] += a;
}
// The context holds the records its own text range yielded (mahip_paf_parse_sharded: ids of the merged dictionary).  Read ranges with equally many hits are
// made from the ranks' summed per-read counts, every record travels to the owner of its query read -- with its position in the record sequence of the WHOLE
// input, which the tie repair of a rank that holds only its own records needs (mahip_hits_set_positions) -- and the context ends as a rank of the sharded
// head wants it: its own records in input order (source ranks hold consecutive ranges: pieces in source order ARE input order), bounds in shard_bounds.
extern "C" int mahip_hits_route(mahip_ctx_t *c, uint64_t *n_total_out, uint64_t *bytes_sent)
{
	HIPCHK(hipSetDevice(c->dev));
	const int W = mahip_comm_world(c), me = mahip_comm_rank(c);
	const uint32_t R = c->n_seq;
	const size_t n = c->n_hits;
	if (W > 32) { mahip_set_error("mahip_hits_route: at most 32 ranks"); return -1; }
	if (n_total_out) *n_total_out = n;
	if (bytes_sent) *bytes_sent = 0;
	if (!mahip_comm_active(c)) { std::vector<uint32_t> b2(2); b2[0] = 0; b2[1] = R; c->shard_bounds = b2; return 0; }
	// (1) hits per read over all ranks -> the same hit-balanced read ranges everywhere
	uint64_t mine = n, every[32], total = 0, base = 0;
	CHK(mahip_comm_all_gather_u64(c, &mine, 1, every));
	for (int r = 0; r < W; ++r) { if (r < me) base += every[r]; total += every[r]; }
	if (total >= 0xffffffffull) { mahip_set_error("mahip_hits_route: too many hits"); return -1; }
	std::vector<uint32_t> b((size_t)W + 1, R);
	b[0] = 0;
	int lrc_pre = 0; // this rank's own failures in front of the first exchange: they travel as the FAILED marker below, so that all ranks leave together
	if (R) {
		// every step of this rank that can fail on its own is agreed on before the collective behind it: a rank that returned here alone would leave the others in the all-reduce
		lrc_pre = dev_reserve(c, c->keep, ((size_t)R + 16) * 4);
		if (lrc_pre == 0 && hipMemsetAsync(c->keep.p, 0, (size_t)R * 4, c->st) != hipSuccess) { mahip_set_error("mahip_hits_route: hipMemsetAsync failed"); lrc_pre = -1; }
		if (lrc_pre == 0 && n) hipLaunchKernelGGL(k_qid_count, dim3(grid_for(n, 256, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, c->d_aos, n, R, P<uint32_t>(c->keep));
		{
			uint64_t bad = lrc_pre ? 1 : 0;
			CHK(mahip_comm_all_reduce_sum_u64(c, &bad, 1));
			if (bad) { if (!lrc_pre) mahip_set_error("mahip_hits_route: another rank could not count its hits"); return -1; }
		}
		CHK(mahip_comm_all_reduce_sum_u32(c, c->keep.p, R));
		std::vector<uint32_t> cnt(R);
		if (hipMemcpyAsync(cnt.data(), c->keep.p, (size_t)R * 4, hipMemcpyDeviceToHost, c->st) != hipSuccess || hipStreamSynchronize(c->st) != hipSuccess) {
			mahip_set_error("mahip_hits_route: the summed hit counts did not come down"); lrc_pre = -1; // (the marker in the row gathered below tells the others)
		}
		unsigned long long run = 0;
		int r = 1;
		for (uint32_t q = 0; q < R && r < W && lrc_pre == 0; ++q) { // as mahip_hits_balance: rank r starts behind the read that completes r/W of the hits
			run += cnt[q];
			while (r < W && run * (unsigned long long)W >= (unsigned long long)total * (unsigned long long)r) b[r++] = q + 1;
		}
	}
	// (2) my records by destination, each with its position in the whole input
	DevBuf send_rec, send_pos, recv_rec, recv_pos;
	uint64_t row[32], mat[32 * 32], by[32 * 32];
	int rc = 0;
	// A rank that fails on its own must not leave the others alone in the next collective: a local failure becomes a marker in the row every rank gathers
	// (a count no rank can hold), every rank sees it and all leave together; the receive buffers' reservation is agreed on the same way before the exchange.
	const uint64_t FAILED = ~0ull;
	do {
		int lrc = lrc_pre; // this rank's own verdict so far
		if (lrc == 0 && (lrc = dev_reserve(c, send_rec, (n + 1) * sizeof(ma_hit_t))) == 0) lrc = dev_reserve(c, send_pos, (n + 1) * 4);
		size_t off = 0;
		for (int h = 0; h < W; ++h) {
			size_t k = 0;
			if (lrc == 0) lrc = mahip_hits_raw_extract_pos(c, b[h], b[h + 1], (char*)send_rec.p + off * sizeof(ma_hit_t), (uint32_t*)send_pos.p + off, &k);
			row[h] = k; off += k;
		}
		if (lrc == 0 && off != n) { mahip_set_error("mahip_hits_route: %zu of %zu records have a query id inside the dictionary", off, n); lrc = -1; }
		if (lrc == 0 && n && base) hipLaunchKernelGGL(k_add_u32, dim3(grid_for(n, 256)), dim3(256), 0, c->st, (uint32_t*)send_pos.p, n, (uint32_t)base);
		if (lrc) for (int h = 0; h < W; ++h) row[h] = FAILED;
		if ((rc = mahip_comm_all_gather_u64(c, row, (size_t)W, mat)) != 0) break; // mat[i * W + j]: records rank i holds for rank j
		bool any_failed = false;
		for (int k = 0; k < W * W; ++k) any_failed |= mat[k] == FAILED;
		if (any_failed) { if (!lrc) mahip_set_error("mahip_hits_route: another rank failed"); rc = -1; break; }
		size_t n_my = 0;
		for (int r = 0; r < W; ++r) n_my += mat[(size_t)r * W + me];
		lrc = dev_reserve(c, recv_rec, (n_my + 1) * sizeof(ma_hit_t));
		if (lrc == 0) lrc = dev_reserve(c, recv_pos, (n_my + 1) * 4);
		{ uint64_t ok = lrc ? 1 : 0; if ((rc = mahip_comm_all_reduce_sum_u64(c, &ok, 1)) != 0) break; if (ok) { if (!lrc) mahip_set_error("mahip_hits_route: another rank could not reserve its receive buffers"); rc = -1; break; } }
		for (int k = 0; k < W * W; ++k) by[k] = mat[k] * sizeof(ma_hit_t);
		if ((rc = mahip_comm_all_to_all_v(c, send_rec.p, recv_rec.p, by)) != 0) break;
		for (int k = 0; k < W * W; ++k) by[k] = mat[k] * 4;
		if ((rc = mahip_comm_all_to_all_v(c, send_pos.p, recv_pos.p, by)) != 0) break;
		if (hipStreamSynchronize(c->st) != hipSuccess) { mahip_set_error("mahip_hits_route: stream error behind the exchange"); rc = -1; break; } // (no HIPCHK inside the block: it would return past the clean-up)
		if (bytes_sent) *bytes_sent = (uint64_t)(n - row[me]) * (sizeof(ma_hit_t) + 4);
		// (3) the context as a rank of the sharded head wants it
		const uint32_t max_qs = c->paf_max_qs;
		dev_free(c, c->aos_own);
		c->aos_own = recv_rec; recv_rec = DevBuf();
		if ((rc = mahip_hits_adopt(c, c->aos_own.p, n_my, R)) != 0) break;
		c->hint_max_qs = max_qs;
		if ((rc = mahip_hits_set_positions(c, (const uint32_t*)recv_pos.p, 1, total)) != 0) break;
		if (hipStreamSynchronize(c->st) != hipSuccess) { mahip_set_error("mahip_hits_route: stream error"); rc = -1; break; }
		c->shard_bounds = b;
		if (n_total_out) *n_total_out = total;
	} while (0);
	dev_free(c, send_rec); dev_free(c, send_pos); dev_free(c, recv_rec); dev_free(c, recv_pos);
	return rc;
}

extern "C" int mahip_set_shard_bounds(mahip_ctx_t *c, const uint32_t *bounds, int world)
{
	if (!bounds || world < 1) { c->shard_bounds.clear(); return 0; }
	c->shard_bounds.assign(bounds, bounds + world + 1);
	return 0;
}
extern "C" const uint32_t *mahip_shard_bounds(mahip_ctx_t *c, int *world)
{
	if (world) *world = c->shard_bounds.empty() ? 0 : (int)c->shard_bounds.size() - 1;
	return c->shard_bounds.empty() ? nullptr : c->shard_bounds.data();
}

extern "C" int mahip_set_shard(mahip_ctx_t *c, uint32_t q_beg, uint32_t q_end)
{
	c->q_beg = q_beg; c->q_end = q_end;
	return 0;
}

extern "C" int mahip_set_full_input(mahip_ctx_t *c, int full)
{
	c->full_input = full != 0;
	return 0;
}


// Own-records shards (mahip_set_full_input(c, 0)): where this context's records stood in the whole input.  The reference's hit order is a function of
// the order of ALL records (hit.c:19-22 sorts them in place with an unstable sort), so a rank that only holds its read range can take part in the tie
// repair only if it knows these positions: the ranks then put the keys of the whole input together (hits_reference_rank).  pos[i] < n_total, all distinct
// over the ranks, increasing on a rank (the records keep their input order).  Copied; set again after every upload/adopt.
extern "C" int mahip_hits_set_positions(mahip_ctx_t *c, const uint32_t *pos, int on_device, uint64_t n_total)
{
	HIPCHK(hipSetDevice(c->dev));
	if (pos == nullptr || n_total == 0) { c->n_total = 0; return 0; }
	if (n_total >= 0xffffffffull || n_total < c->n_in) { mahip_set_error("mahip_hits_set_positions: n_total out of range"); return -1; }
	CHK(dev_reserve(c, c->gpos, (c->n_in + 1) * 4));
	if (c->n_in) HIPCHK(hipMemcpyAsync(c->gpos.p, pos, c->n_in * 4, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->st));
	if (!on_device) HIPCHK(hipStreamSynchronize(c->st)); // the caller's array may go away
	c->n_total = n_total;
	c->hrank_ready = false;
	return 0;
}
extern "C" int mahip_hits_have_positions(mahip_ctx_t *c) { return c->n_total != 0; }

extern "C" int mahip_set_hints(mahip_ctx_t *c, uint32_t max_qs)
{
	c->hint_max_qs = max_qs;
	return 0;
}

extern "C" int mahip_set_run_stride(mahip_ctx_t *c, int stride)
{ // how the records of one query's own PAF lines stand in the input: 2 = record + mirror side by side (ma_hit_read with bi_dir, hit.c:87-98), 1 = no mirrors, 0 = unknown
	c->run_stride = stride == 1 || stride == 2 ? stride : 0;
	return 0;
}

// elements of the last mahip_hits_sort when it sorted RUNS of records (0: it sorted records)
extern "C" uint64_t mahip_hits_sorted_runs(mahip_ctx_t *c) { return c->gk_runs ? (uint64_t)c->n_runs : 0; }

extern "C" int mahip_set_exact_ties(mahip_ctx_t *c, int mode)
{
	c->tie_mode = mode < 0 || mode > 2 ? 2 : mode;
	return 0;
}

extern "C" int mahip_tie_stats(mahip_ctx_t *c, mahip_tie_info_t *out)
{
	if (out) *out = c->tie;
	return 0;
}

static inline bool ctx_sharded(const mahip_ctx *c) { return c->q_beg > 0 || (c->n_seq && c->q_end < c->n_seq); }
static inline uint32_t shard_lo(const mahip_ctx *c) { return c->q_beg < c->n_seq ? c->q_beg : c->n_seq; }
static inline uint32_t shard_hi(const mahip_ctx *c) { return c->q_end < c->n_seq ? c->q_end : c->n_seq; }

// ---- the order ma_hit_sort leaves the hits in (hit.c:19-22): by the ORIGINAL qns of each slot's record, ties as the reference has them ----
// A slot's original key is the qns of its input record (cuts rewrite the qs column, never the input records).
__global__ __launch_bounds__(256) void k_slot_keys(const ma_hit_t *__restrict__ h, const uint32_t *__restrict__ sidx, size_t n, uint64_t *__restrict__ key, uint32_t *__restrict__ val)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) { key[i] = h[sidx[i]].qns; val[i] = (uint32_t)i; }
}
__global__ __launch_bounds__(256) void k_key_tie_count(const uint64_t *__restrict__ skey, size_t n, unsigned long long *__restrict__ ctr, int slot)
{ // adjacent equal keys of a sorted sequence
	uint32_t cnt = 0;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x + 1; i < n; i += (size_t)gridDim.x * 256) cnt += skey[i] == skey[i - 1];
	blk_add_u64(&ctr[slot], cnt);
}
__global__ __launch_bounds__(256) void k_perm_invert(const uint32_t *__restrict__ perm, size_t n, uint32_t *__restrict__ inv)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) inv[perm[i]] = (uint32_t)i;
}
__global__ __launch_bounds__(256) void k_hit_rank(const uint32_t *__restrict__ sidx, const uint32_t *__restrict__ inv, size_t n, uint32_t *__restrict__ hrank)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) hrank[i] = inv[sidx[i]];
}

__global__ __launch_bounds__(256) void k_gkey_scatter(const uint64_t *__restrict__ key, const uint32_t *__restrict__ pos, size_t n, size_t n_total, uint64_t *__restrict__ gkey,
                                                       unsigned long long *__restrict__ ctr)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) {
		const uint32_t p = pos[i];
		if (p < n_total) gkey[p] = key[i]; else atomicAdd(&ctr[CT_OVF2], 1ull);
	}
}
__global__ __launch_bounds__(256) void k_hit_rank_pos(const uint32_t *__restrict__ sidx, const uint32_t *__restrict__ pos, const uint32_t *__restrict__ inv, size_t n, uint32_t *__restrict__ hrank)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) hrank[i] = inv[pos[sidx[i]]];
}

// Own-records shards: the keys of the WHOLE input, in input order, put together from every rank's (key, position) pairs (two all-gathers padded to the
// longest rank); then the same walk on every rank, and hrank[slot] = place of the slot's record among all records of the input.
static int hits_reference_rank_global(mahip_ctx *c)
{
	const size_t n = c->n_hits, N = c->n_in, T = (size_t)c->n_total;
	const int world = mahip_comm_world(c), rank = mahip_comm_rank(c);
	if (world > 32) { mahip_set_error("hits_reference_rank: at most 32 ranks"); return -1; }
	uint64_t cnt[32];
	memset(cnt, 0, sizeof(cnt));
	cnt[rank] = N;
	CHK(mahip_comm_all_reduce_sum_u64(c, cnt, (size_t)world));
	size_t stride = 1, tot = 0;
	for (int r = 0; r < world; ++r) { if (cnt[r] > stride) stride = cnt[r]; tot += cnt[r]; }
	if (tot != T) { mahip_set_error("hits_reference_rank: the ranks hold %zu records, mahip_hits_set_positions said %zu", tot, T); return -1; }
	CHK(dev_reserve(c, c->key[0], (stride + 1) * 8));
	CHK(dev_reserve(c, c->key[1], (T + 1) * 8));
	for (int k = 0; k < 2; ++k) CHK(dev_reserve(c, c->val[k], (T + 1) * 4));
	CHK(dev_reserve(c, c->xb[0], stride * 8 * world + 256));
	CHK(dev_reserve(c, c->xb[1], stride * 4 * world + 256));
	if (N) hipLaunchKernelGGL(k_hit_keys, dim3(grid_for(N, 256, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, c->d_aos, N, P<uint64_t>(c->key[0]), P<uint32_t>(c->val[0]),
	                          (uint32_t*)nullptr, P<unsigned long long>(c->ctr), 0u, 0xffffffffu, 0, 1); // key = qid<<32 | qs, input order
	CHK(mahip_comm_all_gather(c, c->key[0].p, c->xb[0].p, stride * 8));
	if (N) HIPCHK(hipMemcpyAsync(c->val[1].p, c->gpos.p, N * 4, hipMemcpyDeviceToDevice, c->st)); // (a send slot as long as the longest rank's)
	CHK(mahip_comm_all_gather(c, c->val[1].p, c->xb[1].p, stride * 4));
	CHK(ctr_zero(c));
	for (int r = 0; r < world; ++r)
		if (cnt[r]) hipLaunchKernelGGL(k_gkey_scatter, dim3(grid_for(cnt[r], 256)), dim3(256), 0, c->st, (const uint64_t*)c->xb[0].p + (size_t)r * stride,
		                               (const uint32_t*)c->xb[1].p + (size_t)r * stride, (size_t)cnt[r], T, P<uint64_t>(c->key[1]), P<unsigned long long>(c->ctr));
	CHK(ctr_fetch(c));
	if (c->h_ctr[CT_OVF2]) { mahip_set_error("hits_reference_rank: %llu record positions are not below the number of records", (unsigned long long)c->h_ctr[CT_OVF2]); return -1; }
	CHK(reference_order(c, P<uint64_t>(c->key[1]), T, P<uint32_t>(c->val[1])));
	hipLaunchKernelGGL(k_perm_invert, dim3(grid_for(T, 256)), dim3(256), 0, c->st, (const uint32_t*)P<uint32_t>(c->val[1]), T, P<uint32_t>(c->val[0]));
	if (n) hipLaunchKernelGGL(k_hit_rank_pos, dim3(grid_for(n, 256)), dim3(256), 0, c->st, (const uint32_t*)P<uint32_t>(c->sidx), (const uint32_t*)P<uint32_t>(c->gpos),
	                          (const uint32_t*)P<uint32_t>(c->val[0]), n, P<uint32_t>(c->hrank));
	HIPCHK(hipGetLastError());
	return 0;
}

// c->hrank[slot] = position of the slot's record in the order the reference's ma_hit_sort (hit.c:19-22) leaves the input in.
// Both orders are sorted by key, so hrank is the identity outside runs of equal keys.
// stretch of each listed read's hits in the stably sorted keys (ids ascending): seg[2 s] = first position, seg[2 s + 1] = number
__global__ __launch_bounds__(256) void k_read_stretches(const uint64_t *__restrict__ skey, uint32_t n, const uint32_t *__restrict__ ids, uint32_t n_ids, uint32_t *__restrict__ seg)
{
	const uint32_t s = blockIdx.x * 256 + threadIdx.x;
	if (s >= n_ids) return;
	uint32_t b[2];
	for (int k = 0; k < 2; ++k) {
		const uint64_t x = (uint64_t)(ids[s] + (uint32_t)k) << 32; // (id + 1 does not wrap: ids are below 2^32 - 1)
		uint32_t lo = 0, hi = n;
		while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (skey[mid] < x) lo = mid + 1; else hi = mid; }
		b[k] = lo;
	}
	seg[2 * s] = b[0]; seg[2 * s + 1] = b[1] - b[0];
}

int hits_reference_rank(mahip_ctx *c, bool collective_ok, bool wanted_only)
{ // collective_ok: the caller is the orchestrated tie repair (mahip_sg_push_fix, which host/sharded.c enters on ALL ranks after an all-reduce);
	// any other way in must not start a collective one rank alone would wait in for ever
	// wanted_only: c->wantb marks the reads inside whose hit groups the order is needed (graph.hip: push_order); the other slots get their rank in the STABLE order, and
	// c->hrank is then not "the reference's order" for anybody else: hrank_ready stays false
	if (c->hrank_ready) return 0;
	const size_t n = c->n_hits, N = c->n_in; // slots of this context / records of the input
	if (!c->sorted_here || !c->sidx.p || !c->d_aos) { mahip_set_error("hits_reference_rank: the hits were not sorted by this context"); return -1; }
	if (ctx_sharded(c) && !c->full_input) {
		if (c->n_total == 0 || !c->comm || !collective_ok) { mahip_set_error("hits_reference_rank: the reference's tie order is a function of the whole input; this context only holds a shard of it (%s)", c->n_total == 0 || !c->comm ? "and no positions: mahip_hits_set_positions" : "the ranks put their keys together only inside the sharded head's tie repair"); return -1; }
		CHK(dev_reserve(c, c->hrank, (n + 1) * 4));
		CHK(hits_reference_rank_global(c)); // collective: every rank gets here together (host/sharded.c decides on all-reduced counters)
		c->hrank_ready = true;
		c->tie.hit_walk = 1;
		return 0;
	}
	CHK(dev_reserve(c, c->hrank, (n + 1) * 4));
	if (n == 0) { c->hrank_ready = true; return 0; }
	// the walk runs over ALL input records (on a shard: every rank repeats it and keeps the ranks of its own slots; sidx holds global positions)
	TieLaps tl0(c);
	for (int k = 0; k < 2; ++k) { CHK(dev_reserve(c, c->key[k], (N + 1) * 8)); CHK(dev_reserve(c, c->val[k], (N + 1) * 4)); }
	hipLaunchKernelGGL(k_hit_keys, dim3(grid_for(N, 256, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, c->d_aos, N, P<uint64_t>(c->key[0]), P<uint32_t>(c->val[0]),
	                   (uint32_t*)nullptr, P<unsigned long long>(c->ctr), 0u, 0xffffffffu, 0, 1); // key = qid<<32 | qs, input order
	tl0.lap("hit keys (+ buffers)");
	bool restricted = false;
	c->tie.hit_walk_reads = 0;
	if (wanted_only && c->wantb.p && c->n_seq && N < 0xffffffffull) do {
		// the wanted reads (a bit per read id) -> host: their list and the running count the walk asks ("any wanted read among ids lo .. hi?")
		const uint32_t R = c->n_seq, nw = (R + 31) / 32;
		std::vector<uint32_t> bits(nw), wcum((size_t)R + 1), ids;
		HIPCHK(hipMemcpyAsync(bits.data(), c->wantb.p, (size_t)nw * 4, hipMemcpyDeviceToHost, c->st));
		HIPCHK(hipStreamSynchronize(c->st));
		wcum[0] = 0;
		for (uint32_t r = 0; r < R; ++r) { const uint32_t wnt = bits[r >> 5] >> (r & 31) & 1u; wcum[r + 1] = wcum[r] + wnt; if (wnt) ids.push_back(r); }
		const size_t W = ids.size();
		if (W == 0 || W > R / 4) break; // (nobody: cannot be, the caller saw conflicts; a quarter of the reads: the restriction buys little -- the whole walk)
		// the stable order (what every other read keeps) and where the wanted reads' hits stand in it
		const int bs = hits_qs_bits(c), bq = bitlen(R - 1);
		int g = 0;
		CHK(radix_sort_pairs(c, N, 0, bs, 32, 32 + (bq ? bq : 1), &g));
		CHK(dev_reserve(c, c->wseg, W * 12 + 64));
		uint32_t *d_ids = (uint32_t*)c->wseg.p, *d_seg = d_ids + W;
		HIPCHK(hipMemcpyAsync(d_ids, ids.data(), W * 4, hipMemcpyHostToDevice, c->st));
		hipLaunchKernelGGL(k_read_stretches, dim3(grid_for(W, 256)), dim3(256), 0, c->st, (const uint64_t*)P<uint64_t>(c->key[g]), (uint32_t)N, (const uint32_t*)d_ids, (uint32_t)W, d_seg);
		std::vector<uint32_t> seg(2 * W), pos(W), len(W);
		HIPCHK(hipMemcpyAsync(seg.data(), d_seg, W * 8, hipMemcpyDeviceToHost, c->st));
		if (g != 1) HIPCHK(hipMemcpyAsync(c->val[1].p, c->val[0].p, N * 4, hipMemcpyDeviceToDevice, c->st)); // the stable order where reference_order patches it
		HIPCHK(hipStreamSynchronize(c->st));
		for (size_t s = 0; s < W; ++s) pos[s] = seg[2 * s], len[s] = seg[2 * s + 1];
		hipLaunchKernelGGL(k_hit_keys, dim3(grid_for(N, 256, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, c->d_aos, N, P<uint64_t>(c->key[0]), P<uint32_t>(c->val[0]),
		                   (uint32_t*)nullptr, P<unsigned long long>(c->ctr), 0u, 0xffffffffu, 0, 1); // (the sort consumed them) the keys in input order again: what the walk walks
		tl0.lap("stable order + the wanted reads' stretches");
		WalkWanted w = { wcum.data(), R, pos.data(), len.data(), W };
		CHK(reference_order(c, P<uint64_t>(c->key[0]), N, P<uint32_t>(c->val[1]), &w));
		c->tie.hit_walk_reads = W;
		restricted = true;
	} while (0);
	if (!restricted) CHK(reference_order(c, P<uint64_t>(c->key[0]), N, P<uint32_t>(c->val[1])));
	TieLaps tl(c);
	hipLaunchKernelGGL(k_perm_invert, dim3(grid_for(N, 256)), dim3(256), 0, c->st, (const uint32_t*)P<uint32_t>(c->val[1]), N, P<uint32_t>(c->val[0]));
	hipLaunchKernelGGL(k_hit_rank, dim3(grid_for(n, 256)), dim3(256), 0, c->st, (const uint32_t*)P<uint32_t>(c->sidx), (const uint32_t*)P<uint32_t>(c->val[0]), n, P<uint32_t>(c->hrank));
	HIPCHK(hipGetLastError());
	tl.lap("hit ranks on the device");
	c->hrank_ready = !restricted;
	c->tie.hit_walk = 1;
	return 0;
}

// bits of the largest query start (the caller's hint, else one sweep over the records)
int hits_qs_bits(mahip_ctx *c)
{
	if (c->hint_max_qs) return bitlen(c->hint_max_qs);
	if (c->n_in == 0 || !c->d_aos) return 32;
	if (ctr_zero(c) != 0) return 32;
	hipLaunchKernelGGL(k_hit_bounds, dim3(grid_for(c->n_in, 256, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, c->d_aos, c->n_in, P<unsigned long long>(c->ctr));
	if (ctr_fetch(c) != 0) return 32;
	int b = bitlen(c->h_ctr[CT_MAXQS]);
	return b ? b : 1;
}

// *rank = device array: position of every slot in the order the reference's ma_hit_sort leaves the hits in (nullptr: the slots are
// in that order already).  Without equal (qid,qs) keys that is a stable device sort of the slots' original keys; with them it is
// the walk of hits_reference_rank() (whole input on this context), or -- on a shard in the automatic mode -- the stable order, reported
// as unrepaired.
static int hits_order_rank(mahip_ctx *c, const uint32_t **rank)
{
	*rank = nullptr;
	if (!c->sorted_here || !c->sidx.p || !c->d_aos) return 0;
	const size_t n = c->n_hits;
	if (n < 2) return 0;
	if (c->hrank_ready && !ctx_sharded(c)) { *rank = P<uint32_t>(c->hrank); return 0; }
	if (!c->orank_ready) {
		unsigned long long *ctr = P<unsigned long long>(c->ctr);
		const int bs = hits_qs_bits(c), bq = c->n_seq ? bitlen(c->n_seq - 1) : 32;
		for (int k = 0; k < 2; ++k) { CHK(dev_reserve(c, c->key[k], (n + 1) * 8)); CHK(dev_reserve(c, c->val[k], (n + 1) * 4)); }
		CHK(dev_reserve(c, c->orank, (n + 1) * 4));
		hipLaunchKernelGGL(k_slot_keys, dim3(grid_for(n, 256)), dim3(256), 0, c->st, c->d_aos, (const uint32_t*)P<uint32_t>(c->sidx), n, P<uint64_t>(c->key[0]), P<uint32_t>(c->val[0]));
		int g = 0;
		CHK(radix_sort_pairs(c, n, 0, bs, 32, 32 + (bq ? bq : 1), &g)); // stable: equal keys stay in slot order = input order
		HIPCHK(hipMemsetAsync(ctr + ST_HIT_TIES, 0, 8, c->st));
		hipLaunchKernelGGL(k_key_tie_count, dim3(grid_for(n, 256, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, (const uint64_t*)P<uint64_t>(c->key[g]), n, ctr, (int)ST_HIT_TIES);
		hipLaunchKernelGGL(k_perm_invert, dim3(grid_for(n, 256)), dim3(256), 0, c->st, (const uint32_t*)P<uint32_t>(c->val[g]), n, P<uint32_t>(c->orank));
		CHK(ctr_fetch(c));
		HIPCHK(hipGetLastError());
		c->tie.hit_ties = c->h_ctr[ST_HIT_TIES];
		c->orank_ready = true;
	}
	if (c->tie.hit_ties && c->tie_mode != 0) {
		if (ctx_sharded(c)) { // hrank counts positions in the whole input: not an order of this shard's slots alone
			if (c->tie_mode == 1) { mahip_set_error("mahip_hits_download: exact tie order is not available on a shard"); return -1; }
			c->tie.unrepaired = 1;
		} else {
			CHK(hits_reference_rank(c, false));
			*rank = P<uint32_t>(c->hrank);
			return 0;
		}
	}
	*rank = P<uint32_t>(c->orank);
	return 0;
}

extern "C" int mahip_hits_sort(mahip_ctx_t *c)
{ // groups the hits by query id (input order inside a group); see the note at the top of the sort section
	HIPCHK(hipSetDevice(c->dev));
	size_t n = c->n_hits;
	if (n == 0) {
		HIPCHK(hipMemsetAsync(c->goff.p, 0, ((size_t)c->n_seq + 1) * 4, c->st));
		c->soa_ready = true; c->n_live = 0;
		return 0;
	}
	if (n >= 0xffffffffull) { mahip_set_error("mahip_hits_sort: too many hits"); return -1; }
	for (int k = 0; k < 2; ++k) CHK(dev_reserve(c, c->key[k], (n + 1) * 8));
	unsigned long long *ctr = P<unsigned long long>(c->ctr);
	CHK(dev_reserve(c, c->sidx, (n + 128) * 4));
	c->sorted_here = true; c->hrank_ready = false; c->orank_ready = false;
	// digit plan: bits of the query id above the bits of the record index
	int bq, bi = bitlen(n - 1);
	if (c->n_seq) bq = bitlen(c->n_seq - 1); // ids are < n_seq by contract (sdict.c:45-57 hands them out densely)
	else {
		CHK(ctr_zero(c));
		hipLaunchKernelGGL(k_hit_bounds, dim3(grid_for(n, 256, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, c->d_aos, n, ctr);
		CHK(ctr_fetch(c));
		bq = bitlen(c->h_ctr[CT_MAXQID]);
	}
	if (bi == 0) bi = 1;
	if (bq == 0) bq = 1;
	const bool sharded = c->q_beg > 0 || (c->n_seq && c->q_end < c->n_seq);
	int gen = 0;
	if (sharded) { CHK(dev_reserve(c, c->keep, (n + 16) * 4)); CHK(dev_reserve(c, c->pos, (n + 16) * 4)); }
	CHK(ctr_zero(c));
	bool first_hist = false, runs_done = false;
	// ---- RUNS of records as the sort's elements (the kernels' comment): when the caller said how the records of one query's own lines stand in the input
	// (mahip_set_run_stride: 2 with mirrored records, 1 without), the ids are dense and the three fields fit a word.  Falls back to sorting records when the
	// input has few runs, or two runs of one read interleave.
	static const int runs_on = getenv("MA_SORT_RUNS") ? atoi(getenv("MA_SORT_RUNS")) : 1;
	const int bl_runs = 64 - bq - bi;
	if (!sharded && runs_on && c->run_stride && c->n_seq && bl_runs >= RUN_MIN_LEN_BITS && bi <= 32) {
		const int bl = bl_runs > 16 ? 16 : bl_runs;
		const size_t nb1 = (n + RUN_TILE - 1) / RUN_TILE;
		uint32_t *ticket; unsigned long long *state; uint32_t ticket_base, epoch;
		CHK(scan_chain_begin(c, nb1, &state, &ticket, &ticket_base, &epoch));
		{
			ProfScope ps(c, "k_hit_keys", 16.0 * (double)n);
			if (c->run_stride == 2) hipLaunchKernelGGL(k_hit_keys_runs<2>, dim3((unsigned)nb1), dim3(256), 0, c->st, c->d_aos, n, P<uint64_t>(c->key[0]), bi, bl, ctr + CT_TOTAL, state, ticket, ticket_base, epoch, c->n_seq, ctr + CT_OVF);
			else hipLaunchKernelGGL(k_hit_keys_runs<1>, dim3((unsigned)nb1), dim3(256), 0, c->st, c->d_aos, n, P<uint64_t>(c->key[0]), bi, bl, ctr + CT_TOTAL, state, ticket, ticket_base, epoch, c->n_seq, ctr + CT_OVF);
		}
		CHK(ctr_fetch(c));
		const size_t n_runs = (size_t)c->h_ctr[CT_TOTAL];
		if (n_runs && n_runs * 4 <= n * 3 && c->h_ctr[CT_OVF] == 0) { // worth it (else: the keys of all records below, as if nothing had happened)
			int g2 = 0;
			CHK(radix_sort_keys(c, n_runs, bi + bl, bi + bl + bq, &g2, false));
			CHK(radix_group_starts_begin(c, P<uint32_t>(c->goff), c->n_seq, (uint32_t)n));
			const size_t ng2 = (n_runs + (size_t)RX_GROUP * RX_TILE - 1) / ((size_t)RX_GROUP * RX_TILE);
			CHK(dev_reserve(c, c->keep, (ng2 + 16) * 4)); CHK(dev_reserve(c, c->pos, (ng2 + 16) * 4));
			HIPCHK(hipMemsetAsync(ctr + CT_OVF2, 0, 8, c->st));
			{
				ProfScope ps(c, "k_runs_expand", 8.0 * (double)n_runs + 4.0 * (double)n);
				hipLaunchKernelGGL(k_runs_count, dim3((unsigned)ng2), dim3(256), 0, c->st, (const uint64_t*)P<uint64_t>(c->key[g2]), (uint32_t)n_runs, bl, P<uint32_t>(c->keep));
				CHK(scan_exclusive_u32(c, P<uint32_t>(c->keep), P<uint32_t>(c->pos), ng2, nullptr));
				if (c->run_stride == 2) hipLaunchKernelGGL(k_runs_expand<2>, dim3((unsigned)ng2), dim3(256), 0, c->st, (const uint64_t*)P<uint64_t>(c->key[g2]), (uint32_t)n_runs, bi, bl, c->n_seq, (uint32_t)n,
				                                           (const uint32_t*)P<uint32_t>(c->pos), P<uint32_t>(c->sidx), P<uint32_t>(c->goff), ctr);
				else hipLaunchKernelGGL(k_runs_expand<1>, dim3((unsigned)ng2), dim3(256), 0, c->st, (const uint64_t*)P<uint64_t>(c->key[g2]), (uint32_t)n_runs, bi, bl, c->n_seq, (uint32_t)n,
				                        (const uint32_t*)P<uint32_t>(c->pos), P<uint32_t>(c->sidx), P<uint32_t>(c->goff), ctr);
			}
			CHK(radix_group_starts_finish(c, P<uint32_t>(c->goff), c->n_seq));
			CHK(ctr_fetch(c));
			runs_done = c->h_ctr[CT_OVF2] == 0; // interleaved runs of one read (or an id outside the dictionary): sort the records instead
			c->n_runs = runs_done ? n_runs : 0;
		}
		CHK(ctr_zero(c));
	}
	if (!sharded && !runs_done) { // keys + the first pass's per-tile histogram in one sweep
		int sh0, bt0; unsigned tile;
		radix_first_digit(bi, bi + bq, &sh0, &bt0, &tile);
		if (bt0 > 0 && bt0 <= RS_MAXBITS) {
			const unsigned nb = (unsigned)((n + tile - 1) / tile);
			CHK(radix_reserve_hist(c, n));
			ProfScope ps(c, "k_hit_keys", 16.0 * (double)n);
			hipLaunchKernelGGL(k_hit_keys_tiled, dim3(nb), dim3(256), 0, c->st, c->d_aos, n, P<uint64_t>(c->key[0]), bi, P<uint32_t>(c->hist), nb, tile, sh0, (1u << bt0) - 1);
			first_hist = true;
		}
	}
	if (!first_hist && !runs_done) {
		ProfScope ps(c, "k_hit_keys", 16.0 * (double)n); // reads qns (8 B), writes the key
		hipLaunchKernelGGL(k_hit_keys, dim3(grid_for(n, 256, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, c->d_aos, n, P<uint64_t>(c->key[0]), (uint32_t*)nullptr,
		                   sharded ? P<uint32_t>(c->keep) : (uint32_t*)nullptr, ctr, c->q_beg, c->q_end, bi, 0);
	}
	if (sharded) { // this context only keeps the hits whose query read lies in its range
		CHK(ctr_fetch(c));
		size_t n_in = (size_t)c->h_ctr[CT_LIVE];
		if (n_in < n && c->n_total) { // a rank that was handed "its own records" (mahip_hits_set_positions) holds records of somebody else's reads: the read
			// range is not the one the records were selected by -- e.g. a table of bounds left in the context by an earlier input.  Dropping them
			// silently would lose hits on every rank.
			mahip_set_error("mahip_hits_sort: %zu of the %zu records handed to this rank as its own lie outside its read range [%u, %u): stale shard bounds?", n - n_in, n, c->q_beg, c->q_end);
			return -1;
		}
		if (n_in < n) {
			CHK(scan_exclusive_u32(c, P<uint32_t>(c->keep), P<uint32_t>(c->pos), n, nullptr));
			hipLaunchKernelGGL(k_key_compact, dim3(grid_for(n, 256)), dim3(256), 0, c->st, (const uint64_t*)P<uint64_t>(c->key[0]), (const uint32_t*)nullptr,
			                   n, (const uint32_t*)P<uint32_t>(c->keep), (const uint32_t*)P<uint32_t>(c->pos), P<uint64_t>(c->key[1]), (uint32_t*)nullptr);
			gen = 1;
			c->n_hits = n = n_in;
		}
	}
	c->n_live = n;
	if (n == 0) {
		HIPCHK(hipMemsetAsync(c->goff.p, 0, ((size_t)c->n_seq + 1) * 4, c->st));
		c->soa_ready = true;
		return 0;
	}
	if (runs_done) first_hist = true; // (nothing below needs keys)
	static const int fuse_goff = getenv("MA_GOFF_FUSE") ? atoi(getenv("MA_GOFF_FUSE")) : 1;
	if (runs_done) { /* sorted as runs above: sidx and goff are made */ }
	else if (!sharded && fuse_goff && c->n_seq) { // the group offsets come out of the sort's last pass (ids are < n_seq by contract, checked by the kernel)
		const RadixGroups grp = {P<uint32_t>(c->goff), bi, c->n_seq};
		CHK(radix_sort_keys(c, n, bi, bi + bq, &gen, first_hist, &grp));
	} else {
	CHK(radix_sort_keys(c, n, bi, bi + bq, &gen, first_hist));
	{ // the records stay where they are for now: the first consumer moves them (hits_need_cols), ma_hit_sub while it sweeps them
		ProfScope ps(c, "k_hit_goff", 8.0 * (double)n);
		const uint32_t q_lo = sharded ? c->q_beg : 0u, q_hi = sharded && c->q_end < c->n_seq ? c->q_end : c->n_seq;
		if (sharded) hipLaunchKernelGGL(k_goff_outside, dim3(grid_for((size_t)c->n_seq + 1, 256, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, P<uint32_t>(c->goff), q_lo, q_hi, c->n_seq, (uint32_t)n);
		hipLaunchKernelGGL(k_hit_goff, dim3(grid_for(n + 1, 512, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, (const uint64_t*)P<uint64_t>(c->key[gen]), bi, n, c->n_seq, P<uint32_t>(c->goff), q_lo, q_hi);
	}
	}
	HIPCHK(hipGetLastError());
	c->gather_pending = true; c->gk_gen = gen; c->gk_bi = bi; c->gk_runs = runs_done;
	c->soa_ready = true;
	return 0;
}

// the SoA columns of a context whose gather is still pending (mahip_hits_sort leaves it to the first consumer)
int hits_need_cols(mahip_ctx *c, const char *who)
{
	if (!c->soa_ready) { mahip_set_error("%s: hits not indexed", who); return -1; }
	if (!c->gather_pending) return 0;
	const size_t n = c->n_hits;
	ProfScope ps(c, "k_hit_gather", 76.0 * (double)n); // key 8 + record 32 + columns 32 + input position 4
	if (c->gk_runs) hipLaunchKernelGGL(k_hit_gather, dim3(grid_for(n + 1, 256 * GATHER_ILP)), dim3(256), 0, c->st, c->d_aos, (const uint64_t*)nullptr, 0, n, c->n_seq, cols_of(c),
	                                   (uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)P<uint32_t>(c->sidx)); // sorted as runs: positions and group offsets are there (k_runs_expand)
	else
	hipLaunchKernelGGL(k_hit_gather, dim3(grid_for(n + 1, 256 * GATHER_ILP)), dim3(256), 0, c->st, c->d_aos, (const uint64_t*)P<uint64_t>(c->key[c->gk_gen]),
	                   c->gk_bi, n, c->n_seq, cols_of(c), (uint32_t*)nullptr /* the group offsets are there already (k_hit_goff) */, P<uint32_t>(c->sidx));
	HIPCHK(hipGetLastError());
	c->gather_pending = false;
	return 0;
}

extern "C" int mahip_hits_index(mahip_ctx_t *c)
{
	HIPCHK(hipSetDevice(c->dev));
	size_t n = c->n_hits;
	HitCols h = cols_of(c);
	ProfScope ps(c, "k_hit_gather", 64.0 * (double)n);
	hipLaunchKernelGGL(k_hit_gather, dim3(grid_for(n + 1, 256 * GATHER_ILP)), dim3(256), 0, c->st, c->d_aos, (const uint64_t*)nullptr, 0, n, c->n_seq, h, P<uint32_t>(c->goff), (uint32_t*)nullptr);
	HIPCHK(hipGetLastError());
	c->soa_ready = true; c->gather_pending = false;
	c->sorted_here = false; c->hrank_ready = false; c->orank_ready = false; // the caller's order is final (per-symbol path: already the reference's)
	return 0;
}


// The three size-class launches of a coverage pass run side by side on three streams instead of back to back -- each visits all reads and works
// on its own class only, so together they are one pass; side by side the later classes fill the first one's tail (round 3: the fused pass 3.77 -> 3.42 ms).
struct SubFork {
	mahip_ctx *c; bool on;
	hipStream_t side[2]; hipEvent_t start, done[2];
	SubFork(mahip_ctx *c_) : c(c_), on(true) {
		if (!on) return;
		if (!c->sub_side[0] && !c->sub_fork_failed) { // streams and events are made once per context; if any of them cannot be had, all three size classes run on the context's stream
			bool ok = true;
			// (the side streams at the device's highest priority -- the larger classes finish last -- were measured in round 4, visit G: no effect)
			for (int k = 0; k < 2 && ok; ++k) ok = hipStreamCreateWithFlags(&c->sub_side[k], hipStreamNonBlocking) == hipSuccess;
			for (int k = 0; k < 3 && ok; ++k) ok = hipEventCreateWithFlags(&c->sub_ev[k], hipEventDisableTiming) == hipSuccess;
			if (!ok) {
				(void)hipGetLastError();
				for (int k = 0; k < 2; ++k) if (c->sub_side[k]) { (void)hipStreamDestroy(c->sub_side[k]); c->sub_side[k] = nullptr; }
				for (int k = 0; k < 3; ++k) if (c->sub_ev[k]) { (void)hipEventDestroy(c->sub_ev[k]); c->sub_ev[k] = nullptr; }
				c->sub_fork_failed = true;
			}
		}
		if (c->sub_fork_failed) { on = false; return; }
		side[0] = c->sub_side[0]; side[1] = c->sub_side[1]; start = c->sub_ev[0]; done[0] = c->sub_ev[1]; done[1] = c->sub_ev[2];
		(void)hipEventRecord(start, c->st);
		for (int k = 0; k < 2; ++k) (void)hipStreamWaitEvent(side[k], start, 0);
	}
	hipStream_t st(int cls) const { return on && cls > 0 ? side[cls - 1] : c->st; }
	void join() { if (!on) return; for (int k = 0; k < 2; ++k) { (void)hipEventRecord(done[k], side[k]); (void)hipStreamWaitEvent(c->st, done[k], 0); } }
};

extern "C" int mahip_hits_sub(mahip_ctx_t *c, int min_dp, float min_iden, int end_clip, int slot, size_t *n_remained)
{
	HIPCHK(hipSetDevice(c->dev));
	if (!c->soa_ready) { mahip_set_error("mahip_hits_sub: hits not indexed"); return -1; }
	HitCols h = cols_of(c);
	uint32_t R = c->n_seq;
	const uint32_t q_lo = shard_lo(c), q_hi = shard_hi(c), Rr = q_hi > q_lo ? q_hi - q_lo : 1; // a shard sweeps its own reads only (everybody else's have no hits here)
	CHK(ctr_zero(c));
	CHK(dev_reserve(c, c->ovf, ((size_t)R + 1) * 4));
	unsigned long long *ctr = P<unsigned long long>(c->ctr);
	uint2 *sub = P<uint2>(c->sub[slot]);
	SubFuse nofuse = {nullptr, 0, 0, 0, nullptr};
	SubGather nog = {nullptr, 0, 0, nullptr, nullptr, 0, sub_chunk(Rr), q_lo};
	const bool fuse_gather = c->gather_pending && R && c->n_hits && !getenv("MA_NO_GATHER_FUSE");
	if (c->gather_pending && !fuse_gather) CHK(hits_need_cols(c, "mahip_hits_sub"));
	if (fuse_gather) { // the sweep fetches the records itself and writes the columns on the way
		const uint32_t pmask = c->gk_bi >= 32 ? 0xffffffffu : (1u << c->gk_bi) - 1u;
		const SubGather g = c->gk_runs ? SubGather{(const uint32_t*)P<uint32_t>(c->sidx), 1u, 0xffffffffu, c->d_aos, nullptr, (uint32_t)c->n_hits, sub_chunk(Rr), q_lo}
		                               : SubGather{(const uint32_t*)P<uint64_t>(c->key[c->gk_gen]), 2u, pmask, c->d_aos, P<uint32_t>(c->sidx), (uint32_t)c->n_hits, sub_chunk(Rr), q_lo};
		ProfScope ps(c, "k_hit_sub<gather>", (64.0 + 48.0) * (double)c->n_hits); // hit sort 64 (32 r + 32 w, counted once whatever the digit passes) + ma_hit_sub 48 B per stored hit
		SubFork fk(c);
		const dim3 grd(grid_for(Rr, 4, MA_SUB_BLOCKS)), blk(256);
		const uint32_t *gf = (const uint32_t*)P<uint32_t>(c->goff);
		auto launch = [&](int cls) {
			const dim3 gr = cls == 0 ? grd : dim3(sub_big_grid(grd.x));
			if (c->gk_runs) { // positions from sidx (k_runs_expand), which is not written again
				if (cls == 0) hipLaunchKernelGGL((k_hit_sub<false, 0, 2>), gr, blk, 0, fk.st(0), h, gf, q_hi, min_dp, min_iden, end_clip, sub, P<uint32_t>(c->ovf), ctr, nofuse, g);
				else if (cls == 1) hipLaunchKernelGGL((k_hit_sub<false, 1, 2>), gr, blk, 0, fk.st(1), h, gf, q_hi, min_dp, min_iden, end_clip, sub, P<uint32_t>(c->ovf), ctr, nofuse, g);
				else hipLaunchKernelGGL((k_hit_sub<false, 2, 2>), gr, blk, 0, fk.st(2), h, gf, q_hi, min_dp, min_iden, end_clip, sub, P<uint32_t>(c->ovf), ctr, nofuse, g);
			} else {
				if (cls == 0) hipLaunchKernelGGL((k_hit_sub<false, 0, 1>), gr, blk, 0, fk.st(0), h, gf, q_hi, min_dp, min_iden, end_clip, sub, P<uint32_t>(c->ovf), ctr, nofuse, g);
				else if (cls == 1) hipLaunchKernelGGL((k_hit_sub<false, 1, 1>), gr, blk, 0, fk.st(1), h, gf, q_hi, min_dp, min_iden, end_clip, sub, P<uint32_t>(c->ovf), ctr, nofuse, g);
				else hipLaunchKernelGGL((k_hit_sub<false, 2, 1>), gr, blk, 0, fk.st(2), h, gf, q_hi, min_dp, min_iden, end_clip, sub, P<uint32_t>(c->ovf), ctr, nofuse, g);
			}
		};
		for (int k = 0; k < 3; ++k) launch(sub_order()[k]);
		fk.join();
		c->gather_pending = false;
	} else if (R) {
		ProfScope ps(c, "k_hit_sub", 48.0 * (double)c->n_hits + 8.0 * R); // 32 r + 8 w events + 8 r events per stored hit
		hipLaunchKernelGGL((k_hit_sub<false, 0>), dim3(grid_for(Rr, 4, MA_SUB_BLOCKS)), dim3(256), 0, c->st, h, (const uint32_t*)P<uint32_t>(c->goff), q_hi, min_dp, min_iden, end_clip,
		                   sub, P<uint32_t>(c->ovf), ctr, nofuse, nog);
		hipLaunchKernelGGL((k_hit_sub<false, 1>), dim3(grid_for(Rr, 4, MA_SUB_BLOCKS)), dim3(256), 0, c->st, h, (const uint32_t*)P<uint32_t>(c->goff), q_hi, min_dp, min_iden, end_clip,
		                   sub, P<uint32_t>(c->ovf), ctr, nofuse, nog);
		hipLaunchKernelGGL((k_hit_sub<false, 2>), dim3(grid_for(Rr, 4, MA_SUB_BLOCKS)), dim3(256), 0, c->st, h, (const uint32_t*)P<uint32_t>(c->goff), q_hi, min_dp, min_iden, end_clip,
		                   sub, P<uint32_t>(c->ovf), ctr, nofuse, nog);
	}
	if (R) { // tier B always runs behind the register tiers on a small grid: it finds its work list (usually empty) in the device counter
		CHK(dev_reserve(c, c->big0, (2 * c->n_hits + 8) * 4));
		CHK(dev_reserve(c, c->big1, (c->n_hits + 8) * 4));
		ProfScope ps(c, "k_hit_sub_big", 0);
		hipLaunchKernelGGL(k_hit_sub_big<false>, dim3(256), dim3(256), 0, c->st, h, (const uint32_t*)P<uint32_t>(c->goff), (const uint32_t*)P<uint32_t>(c->ovf), (const unsigned long long*)(ctr + CT_OVF),
		                   min_dp, min_iden, end_clip, sub, P<uint32_t>(c->big0), P<uint32_t>(c->big1), ctr, nofuse);
	}
	CHK(ctr_fetch(c));
	HIPCHK(hipGetLastError());
	if (n_remained) *n_remained = (size_t)c->h_ctr[CT_REMAIN];
	return 0;
}

// resident pipeline: hit.c:162-216 (cut against cut_slot, then flt) folded into the coverage pass that writes out_slot
extern "C" int mahip_hits_cutflt_sub(mahip_ctx_t *c, int cut_slot, int min_span, int max_hang, int min_ovlp, int min_dp, float min_iden, int end_clip,
                                     int out_slot, size_t *n_cut, size_t *n_flt, float *cov, size_t *n_remained)
{
	HIPCHK(hipSetDevice(c->dev));
	CHK(hits_need_cols(c, "mahip_hits_cutflt_sub"));
	HitCols h = cols_of(c);
	uint32_t R = c->n_seq;
	const uint32_t q_lo = shard_lo(c), q_hi = shard_hi(c), Rr = q_hi > q_lo ? q_hi - q_lo : 1;
	CHK(ctr_zero(c));
	CHK(dev_reserve(c, c->ovf, ((size_t)R + 1) * 4));
	HIPCHK(hipMemsetAsync(c->r_live.p, 0, R, c->st));
	unsigned long long *ctr = P<unsigned long long>(c->ctr);
	uint2 *sub = P<uint2>(c->sub[out_slot]);
	SubFuse f = {(const uint2*)P<uint2>(c->sub[cut_slot]), min_span, max_hang, min_ovlp, P<uint8_t>(c->r_live)};
	if (R) {
		ProfScope ps(c, "k_hit_sub<cut+flt>", (80.0 + 80.0 + 48.0) * (double)c->n_hits + 8.0 * R); // cut 80 + flt 80 + sub 48 B per hit
		SubFork fk(c);
		const dim3 grd(grid_for(Rr, 4, MA_SUB_BLOCKS)), blk(256);
		const uint32_t *gf = (const uint32_t*)P<uint32_t>(c->goff);
		const SubGather nog2 = SubGather{nullptr, 0, 0, nullptr, nullptr, 0, sub_chunk(Rr), q_lo};
		auto launch = [&](int cls) {
			const dim3 gr = cls == 0 ? grd : dim3(sub_big_grid(grd.x));
			if (cls == 0) hipLaunchKernelGGL((k_hit_sub<true, 0>), gr, blk, 0, fk.st(0), h, gf, q_hi, min_dp, min_iden, end_clip, sub, P<uint32_t>(c->ovf), ctr, f, nog2);
			else if (cls == 1) hipLaunchKernelGGL((k_hit_sub<true, 1>), gr, blk, 0, fk.st(1), h, gf, q_hi, min_dp, min_iden, end_clip, sub, P<uint32_t>(c->ovf), ctr, f, nog2);
			else hipLaunchKernelGGL((k_hit_sub<true, 2>), gr, blk, 0, fk.st(2), h, gf, q_hi, min_dp, min_iden, end_clip, sub, P<uint32_t>(c->ovf), ctr, f, nog2);
		};
		for (int k = 0; k < 3; ++k) launch(sub_order()[k]);
		fk.join();
	}
	if (R) {
		CHK(dev_reserve(c, c->big0, (2 * c->n_hits + 8) * 4));
		CHK(dev_reserve(c, c->big1, (c->n_hits + 8) * 4));
		ProfScope ps(c, "k_hit_sub_big", 0);
		hipLaunchKernelGGL(k_hit_sub_big<true>, dim3(256), dim3(256), 0, c->st, h, (const uint32_t*)P<uint32_t>(c->goff), (const uint32_t*)P<uint32_t>(c->ovf), (const unsigned long long*)(ctr + CT_OVF),
		                   min_dp, min_iden, end_clip, sub, P<uint32_t>(c->big0), P<uint32_t>(c->big1), ctr, f);
	}
	if (R) hipLaunchKernelGGL(k_flt_totlen, dim3(grid_for(R, 256, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, f.cut_sub, (const uint8_t*)P<uint8_t>(c->r_live), R, ctr);
	CHK(ctr_fetch(c));
	HIPCHK(hipGetLastError());
	c->n_live = (size_t)c->h_ctr[CT_LIVE];
	if (n_cut) *n_cut = (size_t)c->h_ctr[CT_CUT];
	if (n_flt) *n_flt = c->n_live;
	if (cov) *cov = (float)((double)c->h_ctr[CT_TOTDP] / (double)c->h_ctr[CT_TOTLEN]);
	if (n_remained) *n_remained = (size_t)c->h_ctr[CT_REMAIN];
	return 0;
}

// resident pipeline: second ma_hit_cut (against cut_slot) + the flag pass of ma_hit_contained (against slot 0) in one sweep;
// the squeeze of the hits is left to ma_sg_gen's pass (lazy), which is the next reader of the hits
// sharded mode runs it in two halves with the flag exchange (max-all-reduce of r_cont / r_used) in between
extern "C" int mahip_hits_cut_contained_flags(mahip_ctx_t *c, int cut_slot, int min_span, const ma_opt_t *opt)
{
	HIPCHK(hipSetDevice(c->dev));
	CHK(hits_need_cols(c, "mahip_hits_cut_contained"));
	size_t n = c->n_hits;
	uint32_t R = c->n_seq;
	CHK(ctr_zero(c));
	HIPCHK(hipMemsetAsync(c->r_cont.p, 0, R, c->st));
	HIPCHK(hipMemsetAsync(c->r_used.p, 0, R, c->st));
	unsigned long long *ctr = P<unsigned long long>(c->ctr);
	if (n) {
		CHK(dev_reserve(c, c->big1, ((size_t)R + 4) * 16)); // (the coverage passes' scratch: free here)
		ProfScope ps(c, "k_hit_cut_contained", (80.0 + 48.0) * (double)c->n_live);
		hipLaunchKernelGGL(k_sub_pair, dim3(grid_for(R, 256)), dim3(256), 0, c->st, (const uint2*)P<uint2>(c->sub[cut_slot]), (const uint2*)P<uint2>(c->sub[0]), R, (uint4*)c->big1.p);
		hipLaunchKernelGGL(k_hit_cut_contained, dim3(grid_for(n, 256, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, cols_of(c), n, (const uint4*)c->big1.p, min_span,
		                   opt->max_hang, opt->int_frac, opt->min_ovlp, P<uint8_t>(c->r_cont), P<uint8_t>(c->r_used), ctr);
	}
	HIPCHK(hipGetLastError());
	return 0;
}

extern "C" int mahip_hits_cut_contained_finish(mahip_ctx_t *c, size_t *n_cut, uint32_t *n_seq_new)
{
	HIPCHK(hipSetDevice(c->dev));
	uint32_t R = c->n_seq;
	CHK(dev_reserve(c, c->keep, ((size_t)R + 16) * 4));
	HIPCHK(hipMemsetAsync(c->r_del.p, 0, R, c->st));
	unsigned long long *ctr = P<unsigned long long>(c->ctr);
	uint32_t *d_tot = (uint32_t*)(ctr + CT_TOTAL);
	if (R) {
		hipLaunchKernelGGL(k_read_del, dim3(grid_for(R, 256)), dim3(256), 0, c->st, (const uint2*)P<uint2>(c->sub[0]), (const uint8_t*)P<uint8_t>(c->r_cont),
		                   (const uint8_t*)P<uint8_t>(c->r_used), P<uint8_t>(c->r_del), P<uint32_t>(c->keep), R);
		CHK(scan_exclusive_u32(c, P<uint32_t>(c->keep), (uint32_t*)c->map.p, R, d_tot));
		hipLaunchKernelGGL(k_map_fix, dim3(grid_for(R, 256)), dim3(256), 0, c->st, P<int32_t>(c->map), (const uint8_t*)P<uint8_t>(c->r_del), R, P<uint32_t>(c->surv));
	}
	CHK(ctr_fetch(c));
	HIPCHK(hipGetLastError());
	c->n_live = (size_t)c->h_ctr[CT_LIVE];
	c->n_seq_new = R ? (uint32_t)(c->h_ctr[CT_TOTAL] & 0xffffffffu) : 0;
	c->has_map = true;
	c->lazy_squeeze = true; // hits of dropped reads still carry no dead bit: readers must consult r_del
	if (n_cut) *n_cut = c->n_live;
	if (n_seq_new) *n_seq_new = c->n_seq_new;
	return 0;
}

extern "C" int mahip_hits_cut_contained(mahip_ctx_t *c, int cut_slot, int min_span, const ma_opt_t *opt, size_t *n_cut, uint32_t *n_seq_new)
{
	CHK(mahip_hits_cut_contained_flags(c, cut_slot, min_span, opt));
	return mahip_hits_cut_contained_finish(c, n_cut, n_seq_new);
}

extern "C" int mahip_hits_cut(mahip_ctx_t *c, int slot, int min_span, size_t *n_live)
{
	HIPCHK(hipSetDevice(c->dev));
	CHK(hits_need_cols(c, "mahip_hits_cut"));
	size_t n = c->n_hits;
	CHK(ctr_zero(c));
	if (n) {
		ProfScope ps(c, "k_hit_cut", 80.0 * (double)c->n_live); // 32 r + 2x8 sub look-ups + 32 w per hit
		hipLaunchKernelGGL(k_hit_cut, dim3(grid_for(n, 256, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, cols_of(c), n, (const uint2*)P<uint2>(c->sub[slot]), min_span, P<unsigned long long>(c->ctr));
	}
	CHK(ctr_fetch(c));
	c->n_live = (size_t)c->h_ctr[CT_LIVE];
	if (n_live) *n_live = c->n_live;
	return 0;
}

extern "C" int mahip_hits_flt(mahip_ctx_t *c, int slot, int max_hang, int min_ovlp, size_t *n_live, float *cov)
{
	HIPCHK(hipSetDevice(c->dev));
	CHK(hits_need_cols(c, "mahip_hits_flt"));
	size_t n = c->n_hits;
	uint32_t R = c->n_seq;
	CHK(ctr_zero(c));
	HIPCHK(hipMemsetAsync(c->r_live.p, 0, R, c->st));
	if (n) {
		ProfScope ps(c, "k_hit_flt", 80.0 * (double)c->n_live);
		hipLaunchKernelGGL(k_hit_flt, dim3(grid_for(n, 256, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, cols_of(c), n, (const uint2*)P<uint2>(c->sub[slot]), max_hang, min_ovlp,
		                   P<uint8_t>(c->r_live), P<unsigned long long>(c->ctr));
	}
	if (R) hipLaunchKernelGGL(k_flt_totlen, dim3(grid_for(R, 256, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, (const uint2*)P<uint2>(c->sub[slot]), (const uint8_t*)P<uint8_t>(c->r_live), R, P<unsigned long long>(c->ctr));
	CHK(ctr_fetch(c));
	c->n_live = (size_t)c->h_ctr[CT_LIVE];
	if (n_live) *n_live = c->n_live;
	if (cov) *cov = (float)((double)c->h_ctr[CT_TOTDP] / (double)c->h_ctr[CT_TOTLEN]); // hit.c:212
	return 0;
}

extern "C" int mahip_sub_merge(mahip_ctx_t *c)
{
	HIPCHK(hipSetDevice(c->dev));
	uint32_t R = c->n_seq;
	if (R) hipLaunchKernelGGL(k_sub_merge, dim3(grid_for(R, 256)), dim3(256), 0, c->st, P<uint2>(c->sub[0]), (const uint2*)P<uint2>(c->sub[1]), R);
	HIPCHK(hipGetLastError());
	return 0;
}

// Pass 1 of ma_hit_contained on the local hits: r_cont / r_used flags (any read id).  In the sharded mode the two
// flag arrays are OR-reduced across ranks between _flags and _finish.
extern "C" int mahip_hits_contained_flags(mahip_ctx_t *c, const ma_opt_t *opt)
{
	HIPCHK(hipSetDevice(c->dev));
	CHK(hits_need_cols(c, "mahip_hits_contained"));
	size_t n = c->n_hits;
	uint32_t R = c->n_seq;
	HIPCHK(hipMemsetAsync(c->r_cont.p, 0, R, c->st));
	HIPCHK(hipMemsetAsync(c->r_used.p, 0, R, c->st));
	if (n) {
		ProfScope ps(c, "k_hit_contained", 48.0 * (double)c->n_live);
		hipLaunchKernelGGL(k_hit_contained, dim3(grid_for(n, 256, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, cols_of(c), n, (const uint2*)P<uint2>(c->sub[0]),
		                   opt->max_hang, opt->int_frac, opt->min_ovlp, P<uint8_t>(c->r_cont), P<uint8_t>(c->r_used));
	}
	HIPCHK(hipGetLastError());
	return 0;
}

// Passes 2+3: per-read delete flags, squeeze map (identical on every rank once the flags are complete), local hits dropped
extern "C" int mahip_hits_contained_finish(mahip_ctx_t *c, const uint8_t *seq_del, uint32_t *n_seq_new, size_t *n_live)
{
	HIPCHK(hipSetDevice(c->dev));
	size_t n = c->n_hits;
	uint32_t R = c->n_seq;
	CHK(ctr_zero(c));
	CHK(dev_reserve(c, c->keep, ((size_t)R + 16) * 4));
	if (seq_del) HIPCHK(hipMemcpyAsync(c->r_del.p, seq_del, R, hipMemcpyHostToDevice, c->st));
	else HIPCHK(hipMemsetAsync(c->r_del.p, 0, R, c->st));
	uint32_t *d_tot = (uint32_t*)(P<unsigned long long>(c->ctr) + CT_TOTAL);
	if (R) {
		hipLaunchKernelGGL(k_read_del, dim3(grid_for(R, 256)), dim3(256), 0, c->st, (const uint2*)P<uint2>(c->sub[0]), (const uint8_t*)P<uint8_t>(c->r_cont),
		                   (const uint8_t*)P<uint8_t>(c->r_used), P<uint8_t>(c->r_del), P<uint32_t>(c->keep), R);
		CHK(scan_exclusive_u32(c, P<uint32_t>(c->keep), (uint32_t*)c->map.p, R, d_tot));
		hipLaunchKernelGGL(k_map_fix, dim3(grid_for(R, 256)), dim3(256), 0, c->st, P<int32_t>(c->map), (const uint8_t*)P<uint8_t>(c->r_del), R, P<uint32_t>(c->surv));
	}
	if (n) {
		ProfScope ps(c, "k_hit_squeeze", 72.0 * (double)c->n_live);
		hipLaunchKernelGGL(k_hit_squeeze, dim3(grid_for(n, 256, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, cols_of(c), n, (const uint8_t*)P<uint8_t>(c->r_del), P<unsigned long long>(c->ctr));
	}
	CHK(ctr_fetch(c));
	c->n_live = (size_t)c->h_ctr[CT_LIVE];
	c->n_seq_new = R ? (uint32_t)(c->h_ctr[CT_TOTAL] & 0xffffffffu) : 0;
	c->has_map = true;
	c->lazy_squeeze = false;
	if (n_seq_new) *n_seq_new = c->n_seq_new;
	if (n_live) *n_live = c->n_live;
	return 0;
}

extern "C" int mahip_hits_contained(mahip_ctx_t *c, const ma_opt_t *opt, const uint8_t *seq_del, uint32_t *n_seq_new, size_t *n_live)
{
	CHK(mahip_hits_contained_flags(c, opt));
	return mahip_hits_contained_finish(c, seq_del, n_seq_new, n_live);
}

// device-to-device access to the read-indexed arrays for the exchanges of the sharded mode (done by the caller over RCCL)
static DevBuf *xbuf(mahip_ctx *c, int which, size_t *elem)
{
	switch (which) {
	case MAHIP_BUF_SUB0: *elem = 8; return &c->sub[0];
	case MAHIP_BUF_SUB1: *elem = 8; return &c->sub[1];
	case MAHIP_BUF_RCONT: *elem = 1; return &c->r_cont;
	case MAHIP_BUF_RUSED: *elem = 1; return &c->r_used;
	case MAHIP_BUF_SDEL: *elem = 1; c->arcs_clean = false; return &c->sdel; // (handed out for an exchange: assume it changes)
	default: return nullptr;
	}
}

extern "C" int mahip_copy_out(mahip_ctx_t *c, int which, void *d_dst, size_t first, size_t count)
{
	HIPCHK(hipSetDevice(c->dev));
	size_t es;
	DevBuf *b = xbuf(c, which, &es);
	if (!b || (first + count) * es > b->cap) { mahip_set_error("mahip_copy_out: bad buffer/range"); return -1; }
	if (count) HIPCHK(hipMemcpyAsync(d_dst, (char*)b->p + first * es, count * es, hipMemcpyDeviceToDevice, c->st));
	if (xchg_needs_sync(c)) HIPCHK(hipStreamSynchronize(c->st)); // the exchange runs on somebody else's stream (mahip_internal.hpp)
	return 0;
}

extern "C" int mahip_copy_in(mahip_ctx_t *c, int which, const void *d_src, size_t first, size_t count)
{
	HIPCHK(hipSetDevice(c->dev));
	size_t es;
	DevBuf *b = xbuf(c, which, &es);
	if (!b || (first + count) * es > b->cap) { mahip_set_error("mahip_copy_in: bad buffer/range"); return -1; }
	if (count) HIPCHK(hipMemcpyAsync((char*)b->p + first * es, d_src, count * es, hipMemcpyDeviceToDevice, c->st));
	if (xchg_needs_sync(c)) HIPCHK(hipStreamSynchronize(c->st)); // the exchange runs on somebody else's stream (mahip_internal.hpp)
	return 0;
}

extern "C" int mahip_sub_upload(mahip_ctx_t *c, int slot, const ma_sub_t *sub, size_t n_sub)
{
	HIPCHK(hipSetDevice(c->dev));
	if (n_sub > c->n_seq) n_sub = c->n_seq;
	HIPCHK(hipMemsetAsync(c->sub[slot].p, 0, (size_t)c->n_seq * 8, c->st));
	if (n_sub) HIPCHK(hipMemcpyAsync(c->sub[slot].p, sub, n_sub * 8, hipMemcpyHostToDevice, c->st));
	return 0;
}

extern "C" int mahip_sub_download(mahip_ctx_t *c, int slot, ma_sub_t *sub, int squeezed)
{
	HIPCHK(hipSetDevice(c->dev));
	uint32_t R = c->n_seq;
	if (R == 0) return 0;
	if (squeezed && c->has_map) {
		CHK(dev_reserve(c, c->pos, ((size_t)R + 1) * 8));
		hipLaunchKernelGGL(k_sub_squeeze, dim3(grid_for(R, 256)), dim3(256), 0, c->st, (const uint2*)P<uint2>(c->sub[slot]), (const int32_t*)P<int32_t>(c->map), R, P<uint2>(c->pos));
		if (c->n_seq_new) HIPCHK(hipMemcpyAsync(sub, c->pos.p, (size_t)c->n_seq_new * 8, hipMemcpyDeviceToHost, c->st));
	} else HIPCHK(hipMemcpyAsync(sub, c->sub[slot].p, (size_t)R * 8, hipMemcpyDeviceToHost, c->st));
	HIPCHK(hipStreamSynchronize(c->st));
	return 0;
}

extern "C" int mahip_seqdel_download(mahip_ctx_t *c, uint8_t *del)
{
	HIPCHK(hipSetDevice(c->dev));
	if (c->n_seq) HIPCHK(hipMemcpyAsync(del, c->r_del.p, c->n_seq, hipMemcpyDeviceToHost, c->st));
	HIPCHK(hipStreamSynchronize(c->st));
	return 0;
}

extern "C" int mahip_map_download(mahip_ctx_t *c, int32_t *map)
{
	HIPCHK(hipSetDevice(c->dev));
	if (!c->has_map) { mahip_set_error("mahip_map_download: no squeeze map"); return -1; }
	if (c->n_seq) HIPCHK(hipMemcpyAsync(map, c->map.p, (size_t)c->n_seq * 4, hipMemcpyDeviceToHost, c->st));
	HIPCHK(hipStreamSynchronize(c->st));
	return 0;
}

extern "C" size_t mahip_hits_live(mahip_ctx_t *c) { return c->n_live; }

extern "C" uint32_t mahip_n_seq_new(mahip_ctx_t *c) { return c->has_map ? c->n_seq_new : c->n_seq; }

extern "C" int mahip_survivors_download(mahip_ctx_t *c, uint32_t *old_ids)
{
	HIPCHK(hipSetDevice(c->dev));
	if (!c->has_map && !c->surv_ready) { mahip_set_error("mahip_survivors_download: no squeeze map"); return -1; }
	if (c->n_seq_new) HIPCHK(hipMemcpyAsync(old_ids, c->surv.p, (size_t)c->n_seq_new * 4, hipMemcpyDeviceToHost, c->st));
	HIPCHK(hipStreamSynchronize(c->st));
	return 0;
}

extern "C" int mahip_hits_download(mahip_ctx_t *c, ma_hit_t *out, size_t *n_out)
{
	HIPCHK(hipSetDevice(c->dev));
	CHK(hits_need_cols(c, "mahip_hits_download"));
	size_t n = c->n_hits;
	if (c->lazy_squeeze && n) { // the resident pipeline postponed the squeeze of the hits: do it now
		CHK(ctr_zero(c));
		hipLaunchKernelGGL(k_hit_squeeze, dim3(grid_for(n, 256, MA_STREAM_BLOCKS)), dim3(256), 0, c->st, cols_of(c), n, (const uint8_t*)P<uint8_t>(c->r_del), P<unsigned long long>(c->ctr));
		CHK(ctr_fetch(c));
		c->n_live = (size_t)c->h_ctr[CT_LIVE];
		c->lazy_squeeze = false;
	}
	if (n_out) *n_out = c->n_live;
	if (n == 0 || c->n_live == 0) return 0;
	HitCols h = cols_of(c);
	// a dump shows the order of ma_hit_sort: the slots are only grouped by read, the rank of each slot in that order is made on demand
	const uint32_t *rank = nullptr;
	CHK(hits_order_rank(c, &rank));
	CHK(dev_reserve(c, c->keep, (n + 16) * 4));
	CHK(dev_reserve(c, c->pos, (n + 16) * 4));
	CHK(dev_reserve(c, c->key[0], (c->n_live + 1) * sizeof(ma_hit_t))); // staging for the dense AoS
	hipLaunchKernelGGL(k_hit_keepflags, dim3(grid_for(n, 256)), dim3(256), 0, c->st, (const uint32_t*)h.bl, n, P<uint32_t>(c->keep), rank);
	CHK(scan_exclusive_u32(c, P<uint32_t>(c->keep), P<uint32_t>(c->pos), n, nullptr));
	hipLaunchKernelGGL(k_hit_export, dim3(grid_for(n, 256)), dim3(256), 0, c->st, h, n, (const uint32_t*)P<uint32_t>(c->pos),
	                   c->has_map ? (const int32_t*)P<int32_t>(c->map) : (const int32_t*)nullptr, (ma_hit_t*)c->key[0].p, rank);
	CHK(xfer_copy(c, c->key[0].p, out, c->n_live * sizeof(ma_hit_t), 0));
	walk_scratch_release(c);
	return 0;
}
