/* mahip.h -- thin C ABI between the host C code and the hand-written HIP kernels (gfx950).
 *
 * One mahip_ctx_t per GPU (one process per GPU).  Hits live in HBM as SoA columns between calls; every
 * pass is an in-place "flag + rewrite" pass (no compaction until something is exported), arcs are a dense
 * SoA edge list + CSR index.  Each entry cites the reference function whose result it reproduces
 * (file:line under the reference tree).  All functions return 0 on success, non-zero on failure with the
 * message available from mahip_strerror(); there is NO CPU fallback: without a usable GPU mahip_create fails.
 */
#ifndef MAHIP_H
#define MAHIP_H

#include <stddef.h>
#include <stdint.h>
#include "miniasm_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mahip_ctx mahip_ctx_t;

int mahip_device_count(void);
/* stream: a hipStream_t to launch on (e.g. torch's current stream), or NULL to create a private one */
mahip_ctx_t *mahip_create(int device, void *stream);
void mahip_destroy(mahip_ctx_t *c);
const char *mahip_strerror(void);
int mahip_sync(mahip_ctx_t *c);
/* one trivial kernel launch + wait (start-up timing: the first launch of a process loads the code object) */
int mahip_first_launch(mahip_ctx_t *c);

/* ---- hits ------------------------------------------------------------------------------------------ */
/* n unsorted (or sorted) 32-byte ma_hit_t records, reads numbered [0,n_seq).  upload = H2D copy of a host
 * array; adopt = the records already sit in HBM at d_hits (not owned, never written). */
int mahip_hits_upload(mahip_ctx_t *c, const ma_hit_t *h, size_t n, uint32_t n_seq);
int mahip_hits_adopt(mahip_ctx_t *c, const void *d_hits, size_t n, uint32_t n_seq);
/* Multi-GPU: this context owns the hits whose query id lies in [q_beg,q_end); call before sort/index. */
int mahip_set_shard(mahip_ctx_t *c, uint32_t q_beg, uint32_t q_end);
/* Read ranges that hold equally many hits: bounds[0..world], rank r owns [bounds[r], bounds[r+1]).  mahip_hits_balance computes the table from the unsorted
 * records in the context (identical on every rank that holds the whole input) and keeps it; mahip_set_shard_bounds installs a table made elsewhere;
 * the orchestrator (host/sharded.c) uses the table of its world size when there is one, equal read counts otherwise.  A table describes ONE upload, like the
 * hints and the positions: every mahip_hits_upload / mahip_hits_adopt forgets it -- install it (again) after the records. */
int mahip_hits_balance(mahip_ctx_t *c, int world, uint32_t *bounds);
int mahip_set_shard_bounds(mahip_ctx_t *c, const uint32_t *bounds, int world);
const uint32_t *mahip_shard_bounds(mahip_ctx_t *c, int *world);

/* ---- PAF text ingest on the device (replaces paf.c:34-67 paf_parse/paf_read + sdict.c:27-45 sd_put + hit.c:70-101, the part of
 * ma_hit_read before the sort).  Load the whole (decompressed) text, parse: afterwards the context holds the unsorted hit
 * records exactly as the reference has them before ma_hit_sort (as after mahip_hits_upload) and the read-name dictionary
 * with the reference's ids (dense, in order of first appearance, first length wins). */
typedef struct {
	uint64_t n_lines;         /* text lines */
	uint64_t n_records;       /* lines with >= 10 columns = what the reference logs as "read %ld hits" (hit.c:102) */
	uint64_t n_stored_lines;  /* lines passing the span/match filter (hit.c:85) */
	uint64_t n_hits;          /* records stored (mirrored hits included) */
	uint64_t name_bytes;      /* bytes mahip_paf_names() writes (names NUL-terminated, back to back, in id order) */
	uint32_t n_seq, max_qs;   /* reads in the dictionary; upper bound of the stored query starts */
	uint32_t n_excl;          /* -R: reads excluded as clearly contained (what the reference logs as "dropped %d contained reads", hit.c:66) */
} mahip_paf_info_t;
int mahip_paf_load_mem(mahip_ctx_t *c, const void *text, size_t nbytes);   /* text in host memory */
int mahip_paf_load_fd(mahip_ctx_t *c, int fd, size_t nbytes);              /* bytes [0,nbytes) of an open plain file, read straight into pinned staging */
int mahip_paf_parse(mahip_ctx_t *c, int min_span, int min_match, int bi_dir, mahip_paf_info_t *info);
/* the same with the -R pre-filter folded in (hit.c:38-68 ma_hit_no_cont + the exclusion test of hit.c:86): names of reads that some line shows
 * to be clearly contained are excluded before ids are given out; lines touching them are dropped */
int mahip_paf_parse_excl(mahip_ctx_t *c, int min_span, int min_match, int bi_dir, int no_cont, int max_hang, float int_frac, mahip_paf_info_t *info);
/* Sharded ingest ("ingest routing option B"; host/ingest_sharded.c): every rank loads and parses ITS byte range of the text (ranges in rank order,
 * cut at line starts) and the ranks exchange what the reference's sequential reader carries across a range border: line counts (occurrence numbers count lines
 * of the whole file), the `bl` a 10-column line inherits (paf.c:54), and the distinct names of every range with their first appearances, merged into ONE
 * dictionary with the reference's ids on every rank (sdict.c:27-45).  Afterwards the context holds the records of its own lines (global ids); info: line /
 * record counts are totals over the ranks, n_hits is this rank's.  Collective: needs a communicator (mahip_comm_init*); -R is not served in this mode. */
int mahip_paf_load_fd_range(mahip_ctx_t *c, int fd, size_t off, size_t nbytes);
int mahip_paf_parse_sharded(mahip_ctx_t *c, int min_span, int min_match, int bi_dir, mahip_paf_info_t *info);
/* after mahip_paf_parse_sharded: every record to the rank that owns its query read (read ranges with equally many hits, from the ranks' summed per-read
 * counts: kept as the context's shard bounds), with its position in the record sequence of the whole input.  The context then holds its own records in input
 * order, as ma_pipeline_head_sharded(full_input = 0) wants them.  *n_total = records of all ranks; *bytes_sent = what this rank sent away. */
int mahip_hits_route(mahip_ctx_t *c, uint64_t *n_total, uint64_t *bytes_sent);
int mahip_paf_names(mahip_ctx_t *c, char *names, uint32_t *lens);
/* the same as ready-made sd_seq_t records (sdict.h:6-10) whose name pointers point into `names` (a host block of name_bytes): seqs16[n_seq * 16 bytes] */
int mahip_paf_seqs(mahip_ctx_t *c, char *names, void *seqs16, uint64_t *tot_len);          /* names[name_bytes], lens[n_seq] = first-seen read lengths */
int mahip_paf_release(mahip_ctx_t *c);                                      /* free the text and the per-line columns */
uint32_t mahip_paf_max_qs(mahip_ctx_t *c);                                  /* info.max_qs of the last parse (a sort hint for later mahip_hits_adopt calls) */
int mahip_hits_raw_download(mahip_ctx_t *c, ma_hit_t *out);                 /* the unsorted records held by the context (n_hits of them) */
/* device-to-device: the unsorted records whose query id lies in [q_beg,q_end), in input order, into d_dst (NULL: only count
 * them); *n_out = their number.  Lets a caller keep the records of one read-range shard (bench.py, multi-GPU set-up). */
int mahip_hits_raw_extract(mahip_ctx_t *c, uint32_t q_beg, uint32_t q_end, void *d_dst, size_t *n_out);
/* the same, and (d_pos != NULL, device) the position each extracted record had in this context's input */
int mahip_hits_raw_extract_pos(mahip_ctx_t *c, uint32_t q_beg, uint32_t q_end, void *d_dst, uint32_t *d_pos, size_t *n_out);
/* A context that holds only the records of its read range (mahip_set_full_input(c, 0)) can take part in the repair of the reference's HIT order
 * (hit.c:19-22: an unstable in-place sort -- the order of tied records is a function of the order of ALL records) only if it knows where its records
 * stood in the whole input: pos[i] < n_total (the records of all ranks), distinct over the ranks, increasing on a rank.  The ranks then put the keys of
 * the whole input together (two all-gathers, only when the tie census asks for it) and every rank runs the walk.  Copied (host or device array); like
 * the hints it describes one upload/adopt.  Without positions such a context leaves tied hits in the stable order and reports `unrepaired`. */
int mahip_hits_set_positions(mahip_ctx_t *c, const uint32_t *pos, int on_device, uint64_t n_total);
int mahip_hits_have_positions(mahip_ctx_t *c);

/* optional: an upper bound of the query starts (e.g. the longest read) lets the on-demand (qid,qs) sorts (hit dumps, push order
 * of the arcs) plan their digits without a sweep over the records; 0 = unknown */
int mahip_set_hints(mahip_ctx_t *c, uint32_t max_qs);
/* optional: how the records of one query's own PAF lines stand in the array: 2 = record and mirror side by side (what ma_hit_read stores with bi_dir, hit.c:87-98),
 * 1 = no mirrored records, 0 = unknown (default).  With 1 or 2 mahip_hits_sort sorts RUNS of records (about half as many keys at stride 2) and expands them
 * afterwards; the result is the same grouping in input order, and it falls back to sorting records by itself when the hint does not fit the data.  Describes one
 * upload/adopt, like the other hints. */
int mahip_set_run_stride(mahip_ctx_t *c, int stride);
uint64_t mahip_hits_sorted_runs(mahip_ctx_t *c); /* elements of the last mahip_hits_sort if it sorted runs, else 0 */
/* Bulk copies between pageable host memory and device memory at PCIe speed: worker threads stage slices through
 * pinned slots while their DMAs run (a plain hipMemcpy of pageable memory is staged by one runtime thread).
 * Synchronous; ordered after the work already queued on the context's stream.  MA_XFER_THREADS sets the workers. */
int mahip_memcpy_h2d(mahip_ctx_t *c, void *d_dst, const void *h_src, size_t bytes);
int mahip_memcpy_d2h(mahip_ctx_t *c, void *h_dst, const void *d_src, size_t bytes);
/* Tie order.  The reference's two sorts (ksort.h:134-183, used at hit.c:21 and asg.c:24) are an in-place MSD radix sort that
 * leaves records with equal keys in an order that is a sequential function of the whole input; the device sorts are stable.
 * The difference is only observable through arcs with equal (u,len) keys, so:
 *   mode 2 (default, "auto"): after the arc sort a census counts (u,len) tie groups.  None: the stable order is provably the
 *           reference's result (whatever the order of tied hits was).  Some: the reference's order is reproduced -- a host walk
 *           over the arc keys (squeezed ids, as the reference sorts them) and, only when two arcs were pushed from hits with
 *           equal (qid,qs), over the hit keys too; the device re-gathers.  Hit dumps (mahip_hits_download) are put in the
 *           reference's order when tied hits exist.
 *   mode 1: the same repair, unconditionally.      mode 0: never (stable total order (key, input position)).
 * Initial value: MA_EXACT_TIES in the environment (unset = 2).  On a shard (mahip_set_shard) the census and the repair run after the
 * arc exchange, driven by the orchestrator (host/sharded.c; the split entry points are listed with the sharded building blocks). */
int mahip_set_exact_ties(mahip_ctx_t *c, int mode);
typedef struct {
	uint64_t arc_tie_groups;   /* groups of >= 2 arcs with equal (u,len) after the last ma_sg_gen */
	uint64_t arc_tie_arcs;     /* arcs in such groups */
	uint64_t push_conflicts;   /* consecutive pushed arcs whose hits had equal (qid,qs): the hit order matters */
	uint64_t hit_ties;         /* hits with the same (qid,qs) as their predecessor (only counted when needed, else 0) */
	int arc_walk, hit_walk;    /* 1 if the reference's arc / hit order was computed on the host for the last graph */
	int unrepaired;            /* 1 if tie groups exist and the stable order was kept (mode 0, or a shard) */
	uint64_t push_conflicts_seen; /* of push_conflicts: those the reference's arc sort can turn into a difference (the two arcs part in a bucket of its radix passes that
	                                 * is walked and holds a tie group, or in an insertion-sorted one where one of them has a twin, csrc/graph.hip); 0 of them => the hit
	                                 * walk is not needed and not done */
	uint64_t hit_walk_reads;      /* hit_walk: reads inside whose hit groups the walk's order was taken (the reads with a conflict in sight; every other read keeps the
	                                 * stable order); 0 = the whole order was taken */
} mahip_tie_info_t;
int mahip_tie_stats(mahip_ctx_t *c, mahip_tie_info_t *out);
/* hit.c:19-22 ma_hit_sort.  The resident layout is GROUPED by query id (stable LSD radix sort on the id bits -> SoA + group offsets),
 * input order inside a group: ma_hit_sub / cut / flt / contained do not look at the order inside a group (hit.c:109-160 sorts its own
 * events).  The (qid, qs) order of ma_hit_sort -- with the reference's order of equal keys, see "Tie order" above -- is produced where
 * it can be observed: mahip_hits_download sorts the slots' original keys on demand, mahip_sg_finish orders the pushed arcs. */
int mahip_hits_sort(mahip_ctx_t *c);
/* same layout change without sorting (input already grouped by query id: the per-symbol ABI path) */
int mahip_hits_index(mahip_ctx_t *c);

/* hit.c:109-160 ma_hit_sub into sub slot 0 or 1 */
int mahip_hits_sub(mahip_ctx_t *c, int min_dp, float min_iden, int end_clip, int slot, size_t *n_remained);
/* hit.c:162-193 ma_hit_cut against sub slot */
int mahip_hits_cut(mahip_ctx_t *c, int slot, int min_span, size_t *n_live);
/* hit.c:195-216 ma_hit_flt (int_frac fixed at .5 there) */
int mahip_hits_flt(mahip_ctx_t *c, int slot, int max_hang, int min_ovlp, size_t *n_live, float *cov);
/* hit.c:218-223 ma_sub_merge: slot0 <- compose(slot0, slot1) */
int mahip_sub_merge(mahip_ctx_t *c);
/* hit.c:225-256 ma_hit_contained (+ hit.c:24-36 mark_unused, sdict.c:69-86 squeeze map).
 * seq_del: optional host array [n_seq] of reads already flagged d->seq[i].del. */
int mahip_hits_contained(mahip_ctx_t *c, const ma_opt_t *opt, const uint8_t *seq_del, uint32_t *n_seq_new, size_t *n_live);

/* Fused forms used by the resident pipeline (same results, fewer sweeps over the hits):
 *   cutflt_sub    : hit.c:162-216 (ma_hit_cut against cut_slot, then ma_hit_flt) applied inside the coverage pass that
 *                   computes out_slot (hit.c:109-160) -- the hits are already in registers there;
 *   cut_contained : the second ma_hit_cut (against cut_slot) + the flag pass and the squeeze MAP of ma_hit_contained
 *                   (against slot 0); the squeeze of the hit array itself is postponed to ma_sg_gen / hits_download. */
int mahip_hits_cutflt_sub(mahip_ctx_t *c, int cut_slot, int min_span, int max_hang, int min_ovlp, int min_dp, float min_iden, int end_clip,
                          int out_slot, size_t *n_cut, size_t *n_flt, float *cov, size_t *n_remained);
int mahip_hits_cut_contained(mahip_ctx_t *c, int cut_slot, int min_span, const ma_opt_t *opt, size_t *n_cut, uint32_t *n_seq_new);
/* its two halves for the sharded mode: the r_cont / r_used flags are max-all-reduced across ranks in between */
int mahip_hits_cut_contained_flags(mahip_ctx_t *c, int cut_slot, int min_span, const ma_opt_t *opt);
int mahip_hits_cut_contained_finish(mahip_ctx_t *c, size_t *n_cut, uint32_t *n_seq_new);

int mahip_sub_upload(mahip_ctx_t *c, int slot, const ma_sub_t *sub, size_t n_sub);
int mahip_sub_download(mahip_ctx_t *c, int slot, ma_sub_t *sub, int squeezed); /* squeezed: compacted by the contained map */
int mahip_seqdel_download(mahip_ctx_t *c, uint8_t *del);                       /* per old read id, after contained */
int mahip_map_download(mahip_ctx_t *c, int32_t *map);                          /* old -> new id (-1 dropped) */
uint32_t mahip_n_seq_new(mahip_ctx_t *c);                                      /* reads left after contained (else n_seq) */
int mahip_survivors_download(mahip_ctx_t *c, uint32_t *old_ids);               /* [n_seq_new] new id -> old id */
size_t mahip_hits_live(mahip_ctx_t *c);
/* live hits, compacted, in the order ma_hit_sort + the order-preserving filters leave them in (hit.c:19-22, 162-258), ids renumbered
 * if contained has run; out must hold mahip_hits_live() */
int mahip_hits_download(mahip_ctx_t *c, ma_hit_t *out, size_t *n);

/* ---- string graph ---------------------------------------------------------------------------------- */
/* asm.c:9-39 ma_sg_gen (+ asg.c:72-80 asg_cleanup with the reference's arc order) from the resident hits.
 * use_sub: lengths from sub slot 0 (else seq_len).  seq_len/seq_del: optional host arrays indexed by the
 * CURRENT read numbering (needed when use_sub == 0; seq_del may be NULL). */
int mahip_sg_gen(mahip_ctx_t *c, const ma_opt_t *opt, int use_sub, const uint32_t *seq_len, const uint8_t *seq_del, uint32_t *n_arc);
/* per-symbol path: take a host graph (dense, sorted, indexed) */
int mahip_asg_upload(mahip_ctx_t *c, const asg_t *g);
/* asg.c:148-193 asg_arc_del_trans marking + asg_cleanup (symm is a separate call, as in the reference) */
int mahip_asg_del_trans(mahip_ctx_t *c, int fuzz, uint32_t *n_reduced);
/* asg.c:104-145 asg_arc_del_multi / asg_arc_del_asymm, each followed by asg_cleanup when it removed arcs */
int mahip_asg_symm(mahip_ctx_t *c, uint32_t *n_multi, uint32_t *n_asymm);
int mahip_asg_del_multi(mahip_ctx_t *c, uint32_t *n_multi);   /* asg.c:104-121 */
int mahip_asg_del_asymm(mahip_ctx_t *c, uint32_t *n_asymm);   /* asg.c:124-138 */
/* asg.c:83-101 asg_arc_del_short marking + cleanup (symm separate) */
int mahip_asg_del_short(mahip_ctx_t *c, float drop_ratio, uint32_t *n_short);
/* Renumber the device graph to the squeezed read ids (sdict.c:69-86 applied to the graph, as the reference has it from ma_sg_gen
 * on): a relabelling -- arc order and CSR positions are unchanged -- after which the cleaners and the unitig pass sweep the
 * surviving reads only.  No-op when no squeeze map is pending. */
int mahip_asg_squeeze(mahip_ctx_t *c);
/* Streams of inputs: hand the reduced graph to a SECOND context on the same device, so that the latency-bound rest of a batch (cleaners,
 * unitigs, downloads: ma_pipeline_tail_fetch on `to`, from another host thread) runs beside the next batch's hit passes on `from`.
 * `from` must hold a graph (after mahip_asg_del_trans / symm); it is squeezed first.  `to` receives the squeezed graph, the surviving reads'
 * intervals (sub slot 0) and their old ids -- O(survivors + arcs) bytes, device to device -- and answers the same queries a context answers
 * after ma_hit_contained + ma_sg_gen (mahip_n_seq_new, mahip_survivors_download, mahip_sub_download, the graph passes, mahip_ug_gen).
 * Returns when the copies are complete: `from` may be reused at once. */
int mahip_tail_handoff(mahip_ctx_t *from, mahip_ctx_t *to);
/* The order-dependent cleaners (asg.c:238-433) as a device fixpoint over versioned state (csrc/clean_core.h); each includes the
 * asg_cleanup the reference runs when something was cut.  Results equal the reference's sequential sweeps exactly. */
int mahip_asg_cut_tip(mahip_ctx_t *c, int max_ext, uint32_t *n_cut);                       /* asg.c:238-254 */
int mahip_asg_cut_internal(mahip_ctx_t *c, int max_ext, uint32_t *n_cut);                  /* asg.c:256-272 */
int mahip_asg_cut_biloop(mahip_ctx_t *c, int max_ext, uint32_t *n_cut);                    /* asg.c:274-306 */
int mahip_asg_pop_bubble(mahip_ctx_t *c, int max_dist, uint32_t *n_pop, uint32_t *n_tips); /* asg.c:412-433 (the graph must be symmetric) */
/* how many asg_pop_bubble calls of this context were run as the reference's sequential sweep on one lane (graphs that are not symmetric or not clean:
 * csrc/clean_core.h, ASSUMPTION; MA_BUBBLE_SEQ=1 forces it) */
uint32_t mahip_bubble_seq_sweeps(mahip_ctx_t *c);
/* asm.c:121-210 ma_ug_gen on the device (csrc/ug.hip): unitigs of the current graph.  Counts: unitigs, reads on them, arcs
 * between unitig ends.  mahip_ug_download: per unitig {reads, length, start, end} (start == end == 0xffffffff: circular) and the
 * offset of its members; members = vertex << 32 | length to the next read; uarcs = the unitig arcs in push order (the
 * reference sorts them with its own sort afterwards, asm.c:208). */
int mahip_ug_gen(mahip_ctx_t *c, uint32_t *n_utg, uint32_t *n_members, uint32_t *n_uarc);
int mahip_ug_download(mahip_ctx_t *c, uint32_t *u_n, uint32_t *u_len, uint32_t *u_start, uint32_t *u_end, uint32_t *u_off, uint64_t *members, asg_arc_t *uarcs);
/* asm.c:216-290 ma_ug_seq as a device byte gather (csrc/useq.hip).  The host reads the sequence file and hands the bases of the
 * reads that sit on a unitig over in batches; a job copies `len` bases from the batch buffer to the unitig arena -- the first `len`
 * bases at src_off (forward) or the reverse complement of the last `len` of the src_len bases there (reverse). */
typedef struct { uint64_t src_off, dst_off; uint32_t src_len, len; uint32_t rev, pad; } mahip_useq_job_t;
int mahip_useq_begin(mahip_ctx_t *c, size_t arena_bytes);          /* arena of all unitig strings, filled with 'N' */
int mahip_useq_batch(mahip_ctx_t *c, const char *h_seq, size_t seq_bytes, const mahip_useq_job_t *h_jobs, size_t n_jobs);
int mahip_useq_end(mahip_ctx_t *c, char *h_arena);                 /* the arena back to the host */
uint32_t mahip_asg_n_arc(mahip_ctx_t *c);
/* iterations of the inner loop of asg_arc_del_trans (asg.c:169) in the last reduction this context ran, counted on the device: SURVEY 8(d) prices the
 * reduction at 16 (A + I) bytes (bench.py: roofline.reduce_group) */
uint64_t mahip_asg_trans_inner(mahip_ctx_t *c);
/* fills g (arc/seq/idx malloc'ed, is_srt=1) in the squeezed numbering */
int mahip_asg_download(mahip_ctx_t *c, asg_t *g);

/* ---- building blocks of the sharded multi-GPU mode ---------------------------------------------------
 * One context per GPU owns the hits whose query id lies in its read range (mahip_set_shard).  Passes that read
 * another read's sub/flags need the complete read-indexed arrays, so the caller exchanges them between the
 * split halves below (host/sharded.c: RCCL all-gather / max all-reduce on the context's stream);
 * the arcs are exchanged once, before the transitive reduction. */
#define MAHIP_BUF_SUB0  0   /* uint2 [n_seq] */
#define MAHIP_BUF_SUB1  1
#define MAHIP_BUF_RCONT 2   /* u8 [n_seq] contained flags   (hit.c:234-235) */
#define MAHIP_BUF_RUSED 3   /* u8 [n_seq] touched-by-a-hit  (hit.c:24-36) */
#define MAHIP_BUF_SDEL  4   /* u8 [n_seq] seq.del           (asm.c:27-34) */
/* device-to-device copies of elements [first, first+count) out of / into one of the arrays above */
int mahip_copy_out(mahip_ctx_t *c, int which, void *d_dst, size_t first, size_t count);
int mahip_copy_in(mahip_ctx_t *c, int which, const void *d_src, size_t first, size_t count);
/* hit.c:225-256 split at the point where the flags must be complete */
int mahip_hits_contained_flags(mahip_ctx_t *c, const ma_opt_t *opt);
int mahip_hits_contained_finish(mahip_ctx_t *c, const uint8_t *seq_del, uint32_t *n_seq_new, size_t *n_live);
/* asm.c:9-39 split at the point where seq.del must be complete */
int mahip_sg_flags(mahip_ctx_t *c, const ma_opt_t *opt, int use_sub, const uint32_t *seq_len, const uint8_t *seq_del);
int mahip_sg_finish(mahip_ctx_t *c, uint32_t *n_arc);
/* local sorted arcs as packed rows {u, v, len, ol|del<<31} (16 B each) */
int mahip_asg_export_rows(mahip_ctx_t *c, void *d_dst);
/* replace the graph by the concatenation of n_ranks blocks of rows (block r holds counts[r] rows at d_src + r*stride rows) */
int mahip_asg_import_rows(mahip_ctx_t *c, const void *d_src, const uint32_t *counts, int n_ranks, size_t stride);
/* Tie repair on shards: after the arc exchange every rank holds the whole sorted graph and its tie census
 * (mahip_tie_stats).  With tie groups: (1) mahip_sg_push_conflicts puts this rank's pushed arcs into the stable (qid, qs, input position)
 * order of their hits and counts consecutive ones whose hits had equal (qid,qs); if the sum over the ranks is > 0, mahip_sg_push_fix puts this rank's pushed arcs into the reference's hit order -- it
 * walks ALL hit keys, so the context must hold the whole input (mahip_set_full_input(c, 1), the default; a caller that only handed
 * over the rank's own records says 0 and gets an error here); (2) the ranks exchange their push-order rows
 * (mahip_asg_export_rows_push) and mahip_asg_import_push_rows builds the reference's arc order from the global push sequence. */
int mahip_set_full_input(mahip_ctx_t *c, int full);
int mahip_sg_push_conflicts(mahip_ctx_t *c, uint64_t *n_conf);
int mahip_sg_push_fix(mahip_ctx_t *c);
int mahip_asg_export_rows_push(mahip_ctx_t *c, void *d_dst);
int mahip_asg_import_push_rows(mahip_ctx_t *c, const void *d_src, const uint32_t *counts, int n_ranks, size_t stride);
/* asg.c:148-186 marking for the vertices [v_beg, v_end) only, no cleanup */
int mahip_asg_del_trans_range(mahip_ctx_t *c, int fuzz, uint32_t v_beg, uint32_t v_end, uint32_t *n_reduced);
/* the ol|del column of arcs [first, first+count) out of / into the graph */
int mahip_asg_flags_out(mahip_ctx_t *c, void *d_dst, size_t first, size_t count);
int mahip_asg_flags_in(mahip_ctx_t *c, const void *d_src, size_t first, size_t count);
/* asg.c:72-80 asg_cleanup on the current graph */
int mahip_asg_cleanup(mahip_ctx_t *c, uint32_t *n_arc);

/* ---- collectives of the sharded mode (csrc/comm.hip), queued on the context's stream ----------------------------------
 * RCCL (librccl opened at run time): rank 0 makes an id, everybody calls mahip_comm_init with it.  mahip_comm_init_shm is a
 * host-staged test double over POSIX shared memory for boxes with a single GPU (N processes on one device). */
int mahip_comm_unique_id(void *id128);
int mahip_comm_init(mahip_ctx_t *c, const void *id128, int rank, int world);
int mahip_comm_init_shm(mahip_ctx_t *c, const char *name, int rank, int world);
void mahip_comm_destroy(mahip_ctx_t *c);
int mahip_comm_rank(mahip_ctx_t *c);
int mahip_comm_world(mahip_ctx_t *c);
int mahip_comm_active(mahip_ctx_t *c);   /* more than one rank -- or one RCCL rank with MA_RCCL_ONE_RANK=1 (the collectives really run) */
int mahip_comm_all_gather(mahip_ctx_t *c, const void *d_send, void *d_recv, size_t bytes_per_rank);  /* d_recv: world x bytes, rank-major */
int mahip_comm_all_reduce_max_u8(mahip_ctx_t *c, void *d_buf, size_t n);                               /* OR of 0/1 flag bytes */
int mahip_comm_all_reduce_sum_u64(mahip_ctx_t *c, uint64_t *h_vals, size_t n);                         /* <= 32 host counters */
/* A transport of the caller's own instead of RCCL / the shared-memory double: five callbacks on HOST buffers (0 = ok); the library stages the device buffers
 * through the host around them.  all_gather: every rank's `bytes` from send into recv, rank-major; all_to_all_v: as mahip_comm_all_to_all_v below (bytes[i *
 * world + j] from rank i to rank j, pieces back to back).  tests/test_dist_gloo.py: host/sharded.c over torch.distributed's gloo backend on the CPU build. */
typedef struct {
	void *user;
	int (*all_gather)(void *user, const void *send, void *recv, size_t bytes);
	int (*all_reduce_max_u8)(void *user, void *buf, size_t n);
	int (*all_reduce_sum_u64)(void *user, uint64_t *vals, size_t n);
	int (*all_reduce_sum_u32)(void *user, void *buf, size_t n);
	int (*all_to_all_v)(void *user, const void *send, void *recv, const uint64_t *bytes);
} mahip_comm_ext_t;
int mahip_comm_init_ext(mahip_ctx_t *c, int rank, int world, const mahip_comm_ext_t *ext);
/* for the ranks' own text ranges (host/ingest_sharded.c): a device-side sum of u32 words, a personalised exchange (bytes[i * world + j] = what rank i sends
 * to rank j, pieces back to back in destination order on the way out and in source order on the way in), and an all-gather of a few host words */
int mahip_comm_all_reduce_sum_u32(mahip_ctx_t *c, void *d_buf, size_t n);
int mahip_comm_all_to_all_v(mahip_ctx_t *c, const void *d_send, void *d_recv, const uint64_t *bytes);
int mahip_comm_all_gather_u64(mahip_ctx_t *c, const uint64_t *h_vals, size_t n, uint64_t *h_out);
int mahip_comm_barrier(mahip_ctx_t *c);
int mahip_xbuf(mahip_ctx_t *c, int slot, size_t bytes, void **d_ptr);                                  /* exchange buffers (slot 0 / 1) */

/* ---- instrumentation -------------------------------------------------------------------------------- */
/* Per-kernel timing with HIP events on the launch stream.  enable=1 brackets every kernel launch with
 * events (adds launch overhead; use for measurement runs only). */
int mahip_prof_enable(mahip_ctx_t *c, int enable);
int mahip_prof_reset(mahip_ctx_t *c);
/* writes up to max entries; returns the number of distinct kernels seen */
typedef struct { const char *name; uint64_t launches; double total_ms; double alg_bytes; } mahip_prof_t;
int mahip_prof_get(mahip_ctx_t *c, mahip_prof_t *out, int max);
/* phase marks: an event on the stream per slot (0..63); after a sync, ms[i] = time from mark first+i to mark first+i+1 (0 if either is missing) */
int mahip_mark(mahip_ctx_t *c, int slot);
int mahip_marks_ms(mahip_ctx_t *c, int first, int n, float *ms);
/* bytes of HBM currently held by the context: buffers in use + what its pool keeps idle for the next request */
size_t mahip_mem_bytes(mahip_ctx_t *c);
/* the idle part alone, and a way to hand it back to the driver (whole free allocations; waits for the context's stream).  A context whose
 * allocation fails trims itself and the process's other contexts on the same GPU on its own; other PROCESSES that share the GPU
 * (MA_COMM=shm ranks) are what this call is for: the sharded head calls it at its end.  released (optional) = bytes returned. */
size_t mahip_mem_pool_bytes(mahip_ctx_t *c);
int mahip_mem_trim(mahip_ctx_t *c, size_t *released);

/* Measurement hooks (csrc/diag.hip; tools/pmc_calibrate.py) -- NOT part of the pipeline: plain access patterns with a known byte count (a 16-byte
 * stream, 8 bytes of every 32-byte record, a random 32-byte fetch, the run-wise scatter of a radix pass, ...), timed with HIP events on the context's
 * stream.  They give the rate this GPU sustains for the patterns the hot path is made of, and a known byte count per pattern to calibrate rocprofv3's
 * FETCH_SIZE / WRITE_SIZE against.  mahip_diag_run: `reps` launches of `pattern` over `bytes` of data; *best_ms = the fastest, *moved = bytes read +
 * written by construction.  0 on success. */
int mahip_diag_patterns(void);
const char *mahip_diag_name(int pattern);
int mahip_diag_run(mahip_ctx_t *c, int pattern, size_t bytes, int reps, double *best_ms, double *moved);
/* device pointers for multi-GPU exchanges done outside (RCCL via torch.distributed): which = MAHIP_PTR_* */
#define MAHIP_PTR_SUB0    0
#define MAHIP_PTR_SUB1    1
#define MAHIP_PTR_RDFLAG  2
void *mahip_devptr(mahip_ctx_t *c, int which, size_t *bytes);

#ifdef __cplusplus
}
#endif
#endif
