// mahip_internal.hpp -- context, device buffers, wave64 helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <vector>
#include <mutex>
#include <string>
#include "mahip.h"
#include "ma_core.h"

void mahip_set_error(const char *fmt, ...);

#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
	mahip_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); return -1; } } while (0)
#define CHK(expr) do { int r_ = (expr); if (r_ != 0) return r_; } while (0)

// ---- grow-only device buffer ----
struct DevBuf {
	void *p = nullptr;
	size_t cap = 0;
	bool ext = false; // its address was handed to the caller (mahip_xbuf / mahip_devptr): streams this context does not know may be using it
};

struct ProfEvent { const char *name; hipEvent_t a, b; double alg_bytes; };
struct ProfAcc { const char *name; uint64_t launches; double ms; double alg_bytes; };

// device memory that DevBufs gave back, kept for the next one that asks (mahip_api.hip: dev_reserve / dev_free)
struct DevPool {
	struct Base { char *p; size_t bytes; };             // what hipMalloc returned
	struct Piece { char *p; size_t cap; char *base; };  // a free stretch of one of them
	std::vector<Base> bases;
	std::vector<Piece> free_pieces;
	std::mutex mu; // a sibling context of the process that runs out of memory trims this pool from ITS thread
};

// big pageable host arrays of the tie walk (radix.hip)
struct BigHost { void *p = nullptr; size_t bytes = 0; bool mapped = false; bool reserve(size_t n); void drop(); };

struct mahip_ctx {
	int dev = 0;
	hipStream_t st = nullptr;
	bool own_stream = false;
	size_t mem_bytes = 0;
	DevPool pool;

	// ---- hits ----
	size_t n_hits = 0;        // slots (live + dead)
	size_t n_in = 0;          // records at d_aos (= n_hits unless a shard range dropped some at the sort)
	bool full_input = true;   // d_aos holds the WHOLE input (false: the caller handed over this rank's records only)
	DevBuf gpos;              // u32 [n_in] own-records shards: position of each of this context's records in the whole input (mahip_hits_set_positions)
	uint64_t n_total = 0;     //   records of the whole input; 0 = no positions known
	size_t n_live = 0;
	uint32_t n_seq = 0;       // reads, original numbering
	uint32_t q_beg = 0, q_end = 0xffffffffu; // shard
	std::vector<uint32_t> shard_bounds;      // read ranges of all ranks (mahip_hits_balance / mahip_set_shard_bounds); empty: equal read counts
	uint32_t hint_max_qs = 0; // upper bound of the query starts (max read length), 0 = unknown
	uint32_t paf_max_qs = 0;  // the same as found by the last mahip_paf_parse (hints are reset by every upload/adopt)
	const ma_hit_t *d_aos = nullptr; // input records (adopted or aos_own)
	DevBuf aos_own;
	DevBuf col[8];            // qid qs qe tn ts te mlrev bl(dead<<31)
	DevBuf goff;              // [n_seq+1]
	DevBuf sub[2];            // uint2 [n_seq]   word0 = s | del<<31, word1 = e
	DevBuf r_cont, r_used, r_del, r_live; // u8 [n_seq]
	DevBuf map;               // int32 [n_seq]  old -> new id, -1 dropped
	DevBuf surv;              // u32 [n_seq_new] new -> old id
	bool soa_ready = false, has_map = false, lazy_squeeze = false;
	bool surv_ready = false;  // surv holds the old ids of this context's reads although no map is pending (mahip_tail_handoff)
	int sg_max_hang = 0, sg_min_ovlp = 0; float sg_int_frac = 0; // classifier options of the last mahip_sg_flags (pass B/C recompute the arcs)
	// ---- tie order.  The device sorts are stable; the reference's are not.  tie_mode 2 (default): after the arc
	// sort a census counts (u,len) tie groups; none => the stable order IS the reference's result; some => the reference's order is
	// reproduced (host walk over the keys) for the arcs and, when two arcs were pushed from hits with equal (qid,qs), for the hits.
	// tie_mode 1: always reproduce it.  tie_mode 0: never (documented total order).
	int tie_mode = 2;
	DevBuf sidx;              // u32 [n_hits] input position of the record in each slot (slots: grouped by query id, input order inside a group)
	DevBuf hrank;             // u32 [n_hits] position of each slot's record in the order the reference's ma_hit_sort gives the input (host walk)
	DevBuf orank;             // u32 [n_hits] position of each slot in the stable (qid, qs, input position) order (device sort, on demand)
	DevBuf aslot;             // u32 [n_arc]  hit slot every pushed arc came from
	DevBuf apos;              // u32 [n_hits / 64] position in the push sequence of the first arc of every 64-slot word (graph.hip: k_sg_emit -> k_arc_group_sort)
	DevBuf pushrows[2];       // sharded mode: this rank's arcs in push order as packed rows (the arc arrays are overwritten by the exchange)
	uint32_t n_push = 0;
	bool sorted_here = false, hrank_ready = false, orank_ready = false; // hits grouped by mahip_hits_sort (d_aos = the unsorted input) / hrank valid / orank valid
	bool gather_pending = false; int gk_gen = 0, gk_bi = 0; // mahip_hits_sort left the records in place: sorted keys in key[gk_gen], position in their low gk_bi bits
	bool gk_runs = false; size_t n_runs = 0; int run_stride = 0; // sorted as RUNS of records (hits.hip: k_hit_keys_runs / k_runs_expand): the positions are in sidx already; run_stride: the caller's hint (mahip_set_run_stride)
	bool push_ordered = false; // sharded mode: pushrows[1] holds this rank's arcs in push order
	mahip_tie_info_t tie = {0, 0, 0, 0, 0, 0, 0, 0, 0};
	uint32_t n_seq_new = 0;

	// ---- arcs (dense SoA, two generations for compaction) ----
	DevBuf au[2], av[2], alen[2], aol[2]; // u32 [n_arc]   aol MSB = del
	int ag = 0;               // current generation
	uint32_t n_arc = 0;
	DevBuf idx;               // u64 [2*n_seq]  start<<32 | count
	DevBuf sdel;              // u8 [n_seq] seq.del
	DevBuf slen;              // u32 [n_seq] seq.len
	bool graph_ready = false;
	bool gsq = false;         // the graph has been renumbered to the squeezed read ids (mahip_asg_squeeze): it has n_seq_new reads, no map applies
	void *comm = nullptr;     // communicator of the sharded mode (comm.hip)
	DevBuf xb[2];             // exchange buffers of the sharded mode
	void *clean = nullptr;    // scratch of the graph cleaners (clean.hip)
	void *ug = nullptr;       // unitig arrays (ug.hip)
	void *useq = nullptr;     // unitig sequence arena (useq.hip)

	// ---- scratch ----
	DevBuf keep, pos;         // u32 flags / scanned positions
	BigHost hwalk, hdig;      // host side of the tie walk: the packed elements and, when they do not hold it, the top level's digits; dropped by walk_scratch_release()
	DevBuf tdig;              // u8 [n] tie walk: the keys' top digit when it does not fit into the packed element (radix.hip: reference_order)
	DevBuf wantb;             // u32 [n_seq / 32] tie walk: the reads whose hit order the arc sort can see (graph.hip: k_arc_push_conflicts_seen)
	DevBuf wseg;              // tie walk: the wanted reads' ids / stretches / order on their way between host and device
	DevBuf key[2], val[2];    // radix sort ping-pong
	DevBuf hist;              // radix block histograms
	DevBuf scan_tmp[3];       // scan levels
	DevBuf gs_tmp;            // radix.hip: tile minima of the group starts
	DevBuf ctr;               // u64 [64] device counters
	DevBuf ovf;               // overflow lists
	DevBuf big0, big1;        // lazily allocated global scratch for oversized groups
	DevBuf marks;             // trans-reduce tier-2 mark arrays
	DevBuf sgmask;            // ma_sg_gen: one candidate bit per hit slot
	uint64_t *h_ctr = nullptr; // pinned host mirror of ctr (host-coherent; word 64 = sequence number of the last publish, see ctr_fetch)
	unsigned long long ctr_seq = 0;
	uint32_t scan_ticket = 0, scan_epoch = 0; // scan.hip: tickets handed out so far, launch number
	void *xfer = nullptr;      // staged-copy worker pool (xfer.hip)
	void *paf = nullptr;       // text-ingest buffers (paf.hip)

	hipEvent_t mark_ev[64] = {}; // phase marks (mahip_mark)
	hipStream_t sub_side[2] = {}; hipEvent_t sub_ev[3] = {}; // side streams of the coverage passes' size classes (hits.hip: SubFork)
	uint64_t tr_inner = 0;                                  // iterations of asg.c:169's loop in the last reduction (mahip_asg_trans_inner)
	bool arcs_clean = false;                                // no arc touches a read with seq.del set (checked when the arcs were made / last cleaned, no read deleted since): asg_arc_rm need not look
	bool radix_arcs = false;                                // the radix passes running now sort arcs (profile names)
	bool sub_fork_failed = false;                           // they could not be created: every size class on the context's stream
	unsigned long long mark_set = 0;
	// ---- profiling ----
	bool prof = false;
	std::vector<ProfEvent> pev;
	std::vector<size_t> pstack;
	std::vector<ProfAcc> pacc;
};

int dev_reserve(mahip_ctx *c, DevBuf &b, size_t bytes);
void dev_free(mahip_ctx *c, DevBuf &b);
void prof_begin(mahip_ctx *c, const char *name, double alg_bytes);
void prof_end(mahip_ctx *c);
int prof_collect(mahip_ctx *c);
void prof_patch_last(mahip_ctx *c, const char *name, double alg_bytes);

struct ProfScope {
	mahip_ctx *c;
	ProfScope(mahip_ctx *c_, const char *name, double alg_bytes) : c(c_) { if (c->prof) prof_begin(c, name, alg_bytes); }
	~ProfScope() { if (c->prof) prof_end(c); }
};

// MA_PIPE_TIMING >= 2: wall time of the steps of the tie repair (a stream sync per lap), `[T::ties]   <step> <ms>` on stderr
struct TieLaps {
	mahip_ctx *c; bool on; double t0 = 0;
	static double now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
	explicit TieLaps(mahip_ctx *c_) : c(c_) { const char *s = getenv("MA_PIPE_TIMING"); on = s && atoi(s) >= 2; if (on) { (void)hipStreamSynchronize(c->st); t0 = now(); } }
	void lap(const char *what) { if (!on) return; (void)hipStreamSynchronize(c->st); const double t1 = now(); fprintf(stderr, "[T::ties]   %-34s %9.3f ms\n", what, (t1 - t0) * 1e3); t0 = t1; }
};

template <typename T> static inline T *P(DevBuf &b) { return (T*)b.p; }

// The building blocks of the sharded mode hand buffers to an exchange.  When the collectives are this context's own (mahip_comm_*: queued on c->st, the C
// orchestration of host/sharded.c) stream order is all that is needed.  When somebody else runs them on another stream (the torch-driven specification,
// miniasm_amd/sharded.py) and the context has a private stream, the producer has to finish first.  A caller-owned stream orders the exchange itself.
static inline bool xchg_needs_sync(const mahip_ctx *c) { return c->own_stream && !c->comm; }

// counters (indices into ctx->ctr)
enum { CT_LIVE = 0, CT_REMAIN, CT_TOTDP, CT_TOTLEN, CT_OVF, CT_NRED, CT_NMULTI, CT_NASYMM, CT_NSHORT, CT_MAXQID, CT_MAXQS, CT_TOTAL, CT_OVF2, CT_MAXLEN, CT_CUT, CT_TRINNER /* iterations of asg.c:169's loop */, CT_PROBED /* list entries asg.c:131 looked at */, CT_N };
static_assert(CT_N <= 48, "a named counter would overlap the sticky counters");
// slots [CT_STICKY, 64) are not touched by ctr_zero: the tie census keeps its results there until they are read
enum { CT_STICKY = 48, ST_ARC_TIE_GROUPS = 48, ST_ARC_TIE_ARCS, ST_PUSH_CONFLICTS, ST_HIT_TIES, ST_PUSH_SEEN };
// words [CT_XCHG, CT_XCHG + CT_XCHG_WORDS): scratch of the collectives (comm.hip: all-reduce of counters, all-gather of u64) -- behind the 64 words the mailbox publishes, so that no
// counter, named or sticky, can ever share a word with it (it used to be `ctr + 16`, which CT_PROBED had reached)
enum { CT_WORDS = 64, CT_XCHG = 64, CT_XCHG_WORDS = 32, CT_ALLOC_WORDS = CT_XCHG + CT_XCHG_WORDS };
int ctr_zero(mahip_ctx *c);
int ctr_fetch(mahip_ctx *c); // D2H + sync into c->h_ctr

// ---- primitives (scan.hip / radix.hip) ----
// exclusive prefix sum of n u32; if d_total != nullptr the grand total is written there (may alias out+n)
int scan_exclusive_u32(mahip_ctx *c, const uint32_t *in, uint32_t *out, size_t n, uint32_t *d_total);
// stable LSD radix sort of (u64 key, u32 val) pairs on key bits [lo0,hi0) and [lo1,hi1); result in key[*gen], val[*gen]
#ifndef RS_MAXBITS
#define RS_MAXBITS 9 // widest radix digit
#endif
int radix_sort_pairs(mahip_ctx *c, size_t n, int lo0, int hi0, int lo1, int hi1, int *gen);
// same for bare u64 keys (a payload such as the record index may ride in the bits below lo): key bits [lo,hi)
// groups (optional): the sorted bits [lo, hi) are a group id in [0, n_id) whose lowest bit is key bit groups->lo (= lo): the last pass also leaves the CSR offsets of
// the groups in groups->start[0 .. n_id] (start[id] = first slot of the id's keys, empty groups closed, start[n_id] = n) -- no sweep over the sorted keys for them
struct RadixGroups { uint32_t *start; int lo; uint32_t n_id; };
int radix_sort_keys(mahip_ctx *c, size_t n, int lo, int hi, int *gen, bool first_hist_ready = false, const RadixGroups *groups = nullptr);
void radix_first_digit(int lo, int hi, int *shift, int *bits, unsigned *tile);
int radix_reserve_hist(mahip_ctx *c, size_t n);
int radix_group_starts_begin(mahip_ctx *c, uint32_t *start, uint32_t n_id, uint32_t n);  // start[id] = ~0, start[n_id] = n; a pass then notes the first slot of every id that has keys ...
int radix_group_starts_finish(mahip_ctx *c, uint32_t *start, uint32_t n_id);           // ... and ids without keys are closed (a suffix minimum): CSR offsets
int scan_chain_begin(mahip_ctx *c, size_t nb, unsigned long long **state, uint32_t **ticket, uint32_t *ticket_base, uint32_t *epoch, size_t n_tickets = 0);
// the permutation the reference's (unstable) sort applies to d_keys[0..n) (input order), written to d_perm
// The restricted tie walk: the reference's order is wanted inside the hit groups of some reads only (graph.hip: the reads whose push conflicts the arc sort can see).
// wcum[id] = wanted ids below id (host, n_ids + 1 words); seg_pos / seg_len: where each wanted read's hits stand in any order sorted by key (host, n_seg entries).
struct WalkWanted { const uint32_t *wcum; uint64_t n_ids; const uint32_t *seg_pos, *seg_len; size_t n_seg; };
// w == nullptr: d_perm <- the whole order.  w: d_perm holds the STABLE order on entry and only the wanted reads' stretches are replaced
int reference_order(mahip_ctx *c, uint64_t *d_keys /* overwritten */, size_t n, uint32_t *d_perm, const WalkWanted *w = nullptr);
void walk_scratch_release(mahip_ctx *c); // the host arrays of the walks go away (on a thread of their own when they are big)
// position of every hit slot in the reference's order -> c->hrank (hits.hip)
int hits_reference_rank(mahip_ctx *c, bool collective_ok, bool wanted_only = false);
// bits of the largest query start of the input records
int hits_qs_bits(mahip_ctx *c);
// make sure the SoA columns exist (the gather after mahip_hits_sort is lazy)
int hits_need_cols(mahip_ctx *c, const char *who);
// bulk pageable<->device copy through per-thread pinned slots (xfer.hip); returns after the copy is complete
int xfer_copy(mahip_ctx *c, void *dev_ptr, void *host_ptr, size_t bytes, int to_device);
void xfer_pool_free(mahip_ctx *c);
int xfer_from_fd(mahip_ctx *c, void *dev_ptr, int fd, size_t bytes);
int xfer_from_fd_at(mahip_ctx *c, void *dev_ptr, int fd, size_t off, size_t bytes);
void paf_free(mahip_ctx *c);
void clean_free(mahip_ctx *c);
void ug_free(mahip_ctx *c);
void useq_free(mahip_ctx *c);

static inline unsigned grid_for(size_t n, unsigned per_block, unsigned cap = 0x7fffffffu)
{
	size_t g = (n + per_block - 1) / per_block;
	if (g < 1) g = 1;
	if (g > cap) g = cap;
	return (unsigned)g;
}

// ---- wave64 device helpers ----
#ifdef __HIPCC__
__device__ __forceinline__ unsigned wv_lane() { return __lane_id(); }
__device__ __forceinline__ uint64_t wv_lt(unsigned lane) { return (1ull << lane) - 1ull; }
__device__ __forceinline__ uint64_t wv_le(unsigned lane) { return lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull); }
__device__ __forceinline__ uint64_t wv_ballot(int p) { return __ballot(p); }
// make LDS/global writes of this wave visible to its other lanes, and stop the compiler from reordering
__device__ __forceinline__ void wv_sync()
{
	// workgroup scope: also drains this wave's outstanding LDS/global stores (s_waitcnt), which the
	// global-scratch second tiers rely on; the L1 is per CU and write-through, so no invalidate is needed
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ uint32_t wv_bcast(uint32_t x, int src) { return __shfl(x, src, 64); }
__device__ __forceinline__ uint64_t wv_max_u64(uint64_t x)
{
	for (int o = 32; o > 0; o >>= 1) {
		uint32_t lo = __shfl_xor((uint32_t)x, o, 64), hi = __shfl_xor((uint32_t)(x >> 32), o, 64);
		uint64_t y = (uint64_t)hi << 32 | lo;
		x = y > x ? y : x;
	}
	return x;
}
__device__ __forceinline__ uint32_t wv_sum_u32(uint32_t x)
{
	for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
	return x;
}
__device__ __forceinline__ uint64_t wv_sum_u64(uint64_t x)
{
	for (int o = 32; o > 0; o >>= 1) {
		uint32_t lo = __shfl_xor((uint32_t)x, o, 64), hi = __shfl_xor((uint32_t)(x >> 32), o, 64);
		x += (uint64_t)hi << 32 | lo;
	}
	return x;
}
// ---- wave64 scans for a FULL wave (all 64 lanes active), as the coverage sweep needs them.
// DPP row shifts + the gfx9 row broadcasts (row_bcast:15 / :31, the sequence LLVM's atomic optimizer emits for wave64): one VALU
// instruction per step, nothing through the LDS crossbar (round 3: -6 % on the fused coverage pass against the __shfl_up form, whose 18
// chained steps per read were ds_bpermute round trips).
#define WV_DPP(old, x, ctrl, rm) __builtin_amdgcn_update_dpp((int)(old), (int)(x), (ctrl), (rm), 0xf, false)
__device__ __forceinline__ int wv_scan_incl_i32(int x, unsigned lane)
{
	(void)lane;
	x += WV_DPP(0, x, 0x111, 0xf); x += WV_DPP(0, x, 0x112, 0xf); x += WV_DPP(0, x, 0x114, 0xf); x += WV_DPP(0, x, 0x118, 0xf); // inside the rows
	x += WV_DPP(0, x, 0x142, 0xa); // row_bcast:15 -> rows 1 and 3
	x += WV_DPP(0, x, 0x143, 0xc); // row_bcast:31 -> rows 2 and 3
	return x;
}
// inclusive scan of "the last value that is not `none`"
__device__ __forceinline__ uint32_t wv_scan_last_u32(uint32_t x, uint32_t none, unsigned lane)
{
	(void)lane;
	uint32_t t;
	t = (uint32_t)WV_DPP(none, x, 0x111, 0xf); x = x == none ? t : x;
	t = (uint32_t)WV_DPP(none, x, 0x112, 0xf); x = x == none ? t : x;
	t = (uint32_t)WV_DPP(none, x, 0x114, 0xf); x = x == none ? t : x;
	t = (uint32_t)WV_DPP(none, x, 0x118, 0xf); x = x == none ? t : x;
	t = (uint32_t)WV_DPP(none, x, 0x142, 0xa); x = x == none ? t : x;
	t = (uint32_t)WV_DPP(none, x, 0x143, 0xc); x = x == none ? t : x;
	return x;
}
// value of the lane below (lane 0: fill)
__device__ __forceinline__ uint32_t wv_prev_lane_u32(uint32_t x, uint32_t fill, unsigned lane) { (void)lane; return (uint32_t)WV_DPP(fill, x, 0x138, 0xf); } // wave_shr:1
// maximum over the wave, the same value in every lane: an inclusive max-scan, then lane 63 read back through the scalar unit
#define WV_MAX64_STEP(ctrl, rm) do { \
		const uint32_t lo_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)x, (ctrl), (rm), 0xf, false); \
		const uint32_t hi_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(x >> 32), (ctrl), (rm), 0xf, false); \
		const uint64_t y_ = (uint64_t)hi_ << 32 | lo_; \
		x = y_ > x ? y_ : x; } while (0)
__device__ __forceinline__ uint64_t wv_max_u64_full(uint64_t x)
{
	WV_MAX64_STEP(0x111, 0xf); WV_MAX64_STEP(0x112, 0xf); WV_MAX64_STEP(0x114, 0xf); WV_MAX64_STEP(0x118, 0xf);
	WV_MAX64_STEP(0x142, 0xa); WV_MAX64_STEP(0x143, 0xc);
	return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), 63) << 32 | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, 63);
}
#undef WV_MAX64_STEP
// the value one lane holds, for all (src is the same in every lane)
__device__ __forceinline__ uint32_t wv_read_lane_u32(uint32_t x, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)x, src); }
#undef WV_DPP
// one atomicAdd per wave of the number of lanes with p set
__device__ __forceinline__ void wv_count_add(unsigned long long *ctr, int p)
{
	uint64_t m = __ballot(p);
	if (m && wv_lane() == (unsigned)(__ffsll((long long)m) - 1)) atomicAdd(ctr, (unsigned long long)__popcll(m));
}
// Block-level reductions into a global counter: ONE atomic per block.  A single counter word sustains only
// ~88 atomics/us on MI355X, so per-wave atomics from a 20M-element pass cost milliseconds; kernels therefore run
// as grid-stride loops over <= 2048 blocks, keep private partial results and call these once at the end.
// Every thread of the block must call (contains __syncthreads); blocks have <= 512 threads.
__device__ __forceinline__ void blk_add_u64(unsigned long long *ctr, uint64_t x)
{
	__shared__ unsigned long long s_part[8];
	x = wv_sum_u64(x);
	if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = x;
	__syncthreads();
	if (threadIdx.x == 0) {
		unsigned long long t = 0;
		for (unsigned w = 0; w < (blockDim.x + 63) / 64; ++w) t += s_part[w];
		if (t) atomicAdd(ctr, t);
	}
	__syncthreads();
}
__device__ __forceinline__ void blk_max_u64(unsigned long long *ctr, uint64_t x)
{
	__shared__ unsigned long long s_part[8];
	x = wv_max_u64(x);
	if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = x;
	__syncthreads();
	if (threadIdx.x == 0) {
		unsigned long long t = 0;
		for (unsigned w = 0; w < (blockDim.x + 63) / 64; ++w) t = s_part[w] > t ? s_part[w] : t;
		if (t) atomicMax(ctr, t);
	}
	__syncthreads();
}
// block-wide exclusive scan of one u32 per thread for 256-thread blocks; s_wave: 4 words of LDS scratch
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t x, uint32_t *s_wave, uint32_t *total)
{
	unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t incl = (uint32_t)wv_scan_incl_i32((int)x, lane); // DPP row shifts + row broadcasts (every thread of the block is here: full waves); until round 5 six ds_bpermute round trips
	if (lane == 63) s_wave[wave] = incl;
	__syncthreads();
	uint32_t base = 0, tot = 0;
	for (unsigned w = 0; w < 4; ++w) {
		uint32_t v = s_wave[w];
		if (w < wave) base += v;
		tot += v;
	}
	__syncthreads();
	*total = tot;
	return base + incl - x;
}
// ---- chained tiles (scan.hip: k_scan_chain; graph.hip: k_arc_rm_chain) ----
#define SC_AGG 1ull
#define SC_INCL 2ull
__device__ __forceinline__ unsigned long long sc_pack(uint32_t epoch, unsigned long long st, uint32_t v) { return (unsigned long long)epoch << 34 | st << 32 | v; }
// A published word carries its own flag, so nothing has to be ordered against it: RELAXED atomics at AGENT scope (served where the eight XCDs' L2s meet).
// The first version used __atomic_store_n(RELEASE) / __atomic_load_n(ACQUIRE), i.e. SYSTEM scope: on this chip a release writes the XCD's L2 back and an
// acquire invalidates it -- per tile, and per spin of a waiting lane.  A 100 M-element scan took 11.4 ms instead of 0.4 (round 3, visit E).
#define SC_PUBLISH(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define SC_PEEK(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// exclusive prefix of tile `tile` (> 0) from the words its predecessors published: one wave, 64 predecessors per step (one per lane), until one of them
// knows its inclusive prefix.  Tiles take their numbers from an atomic ticket, so every predecessor has started and publishes without waiting for anybody.
__device__ __forceinline__ uint32_t sc_look_back(const unsigned long long *state, uint32_t tile, uint32_t epoch, unsigned lane)
{
	uint32_t prefix = 0;
	for (long look = (long)tile - 1;; look -= 64) {
		const long idx = look - (long)lane;
		unsigned long long w = sc_pack(epoch, SC_INCL, 0); // in front of tile 0: an inclusive prefix of zero
		if (idx >= 0) do { w = SC_PEEK(&state[idx]); } while ((uint32_t)(w >> 34) != epoch || ((w >> 32) & 3ull) == 0);
		const unsigned long long incl = wv_ballot(((w >> 32) & 3ull) == SC_INCL);
		const int first = incl ? __ffsll((long long)incl) - 1 : 63; // the nearest predecessor that knows its inclusive prefix
		prefix += wv_sum_u32(lane <= (unsigned)first ? (uint32_t)w : 0u);
		if (incl) break;
	}
	return prefix;
}
// ---- the wave's register sorting network (coverage sweeps of hits.hip, arc groups of graph.hip) ----
#define MA_CE(a, b) do { uint32_t lo_ = (a) < (b) ? (a) : (b), hi_ = (a) < (b) ? (b) : (a); (a) = lo_; (b) = hi_; } while (0)

// what a lane keeps of a compare-exchange with its partner's value y: the smaller one in the lower lane, the larger one in the upper;
// min / max take the DPP operand themselves and the select reads the lane mask from scalar registers (round 3: -5 % on the fused pass
// against mov_dpp + compare + xor on vcc + select)
// (Round 3: min + max with the DPP operand folded in, then a select -- 3 VALU.)  Round 4 (measured on one box: the fused coverage pass 3.37 -> 3.26 ms at BASELINE configs[3], 2.75 -> 2.50 ms on the graph-heavy input; profiles/r04_experiments.txt):
// ONE instruction instead of min + max + select: the median of (x, y, 0) is min(x, y), the median of (x, y, ~0) is max(x, y) -- v_med3_u32 with the
// lane's role (0 in the lower lane of a pair, ~0 in the upper) as its third operand.  gfx9 has no DPP form of a three-operand instruction, so the partner's
// value arrives through a v_mov_b32 dpp (or ds_bpermute across rows): 2 VALU per compare-exchange instead of 3.
__device__ __forceinline__ uint32_t ma_med3_u32(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
	uint32_t r;
	asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
	return r;
#else
	const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
	return c < lo ? lo : c > hi ? hi : c;
#endif
}
#define MA_KEEP(x, y, lower) ma_med3_u32((x), (y), (lower) ? 0u : 0xffffffffu)
// Value of lane (lane ^ M) for a compile-time M.  Exchanges inside a row of 16 lanes are DPP modifiers of a VALU move
// (quad_perm, row_half_mirror, row_mirror, row_ror, banked row_shl/shr): no LDS crossbar round trip, no s_waitcnt.
// Only the exchanges across rows (16, 31, 63) go through ds_bpermute.
__device__ __forceinline__ uint32_t lane_xor(uint32_t x, int m)
{
	const int v = (int)x;
	switch (m) {
	case 1: return (uint32_t)__builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
	case 2: return (uint32_t)__builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
	case 3: return (uint32_t)__builtin_amdgcn_mov_dpp(v, 0x1B, 0xf, 0xf, true);  // quad_perm [3,2,1,0]
	case 7: return (uint32_t)__builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true); // row_half_mirror
	case 15: return (uint32_t)__builtin_amdgcn_mov_dpp(v, 0x140, 0xf, 0xf, true); // row_mirror
	case 8: return (uint32_t)__builtin_amdgcn_mov_dpp(v, 0x128, 0xf, 0xf, true);  // row_ror:8
	case 4: { // banks with lane bit 2 clear read lane+4 (row_shl:4), the others lane-4 (row_shr:4)
		int t = __builtin_amdgcn_mov_dpp(v, 0x104, 0xf, 0x5, true);
		return (uint32_t)__builtin_amdgcn_update_dpp(t, v, 0x114, 0xf, 0xa, false);
	}
	default: return __shfl_xor(x, m, 64);
	}
}

// ascending sort of the 64*ITEMS values held blocked (element lane*ITEMS + r) across one wave
template <int ITEMS>
__device__ __forceinline__ void wave_sort_regs(uint32_t (&x)[ITEMS], unsigned lane)
{
#pragma unroll
	for (int k = 2; k <= ITEMS; k <<= 1) { // inside a lane
#pragma unroll
		for (int r = 0; r < ITEMS; ++r) { int p = r ^ (k - 1); if (p > r) MA_CE(x[r], x[p]); }
#pragma unroll
		for (int j = k >> 2; j > 0; j >>= 1)
#pragma unroll
			for (int r = 0; r < ITEMS; ++r) if ((r & j) == 0) MA_CE(x[r], x[r | j]);
	}
#pragma unroll
	for (int L = 2; L <= 64; L <<= 1) { // merge blocks of L lanes
		{ // flip step: element e pairs with e ^ (L*ITEMS - 1) = (lane ^ (L-1), ITEMS-1-r)
			const bool lower = (lane & (L >> 1)) == 0;
			uint32_t y[ITEMS];
#pragma unroll
			for (int r = 0; r < ITEMS; ++r) y[r] = lane_xor(x[ITEMS - 1 - r], L - 1);
#pragma unroll
			for (int r = 0; r < ITEMS; ++r) x[r] = MA_KEEP(x[r], y[r], lower); // lower half keeps the min, upper the max
		}
#pragma unroll
		for (int m = L >> 2; m > 0; m >>= 1) { // half cleaners across lanes
			const bool lower = (lane & m) == 0;
#pragma unroll
			for (int r = 0; r < ITEMS; ++r) {
				uint32_t y = lane_xor(x[r], m);
				x[r] = MA_KEEP(x[r], y, lower);
			}
		}
#pragma unroll
		for (int j = ITEMS >> 1; j > 0; j >>= 1) // half cleaners inside a lane
#pragma unroll
			for (int r = 0; r < ITEMS; ++r) if ((r & j) == 0) MA_CE(x[r], x[r | j]);
	}
}

#define MA_STREAM_BLOCKS 2048u // 256 CUs x 8 blocks of 256 threads: full occupancy for streaming passes

// All-ascending bitonic network on a[0..n) (n need not be a power of two: indices >= n act as +inf).
// STRIDE threads cooperate (tid in [0,STRIDE)); SYNC() separates the stages.
#define MA_BITONIC(T, a, n, tid, STRIDE, SYNC) do { \
	uint32_t P_ = 1; while (P_ < (uint32_t)(n)) P_ <<= 1; \
	for (uint32_t k_ = 2; k_ <= P_; k_ <<= 1) { \
		{ uint32_t h_ = k_ >> 1; \
		  for (uint32_t t_ = (tid); t_ < (P_ >> 1); t_ += (STRIDE)) { \
			uint32_t lo_ = (t_ / h_) * k_ + (t_ % h_), hi_ = lo_ ^ (k_ - 1); \
			if (hi_ < (uint32_t)(n)) { T x_ = (a)[lo_], y_ = (a)[hi_]; if (y_ < x_) (a)[lo_] = y_, (a)[hi_] = x_; } \
		  } SYNC; } \
		for (uint32_t j_ = k_ >> 2; j_ > 0; j_ >>= 1) { \
		  for (uint32_t t_ = (tid); t_ < (P_ >> 1); t_ += (STRIDE)) { \
			uint32_t lo_ = ((t_ & ~(j_ - 1)) << 1) | (t_ & (j_ - 1)), hi_ = lo_ | j_; \
			if (hi_ < (uint32_t)(n)) { T x_ = (a)[lo_], y_ = (a)[hi_]; if (y_ < x_) (a)[lo_] = y_, (a)[hi_] = x_; } \
		  } SYNC; } \
	} } while (0)
#endif
