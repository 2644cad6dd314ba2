/* ma_host.h -- internal declarations shared by the host C files (not part of the drop-in ABI) */
#ifndef MA_HOST_H
#define MA_HOST_H

#include "miniasm_amd.h"
#include "mahip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* log sink of the [M::...] lines (stderr unless redirected; bench.py silences it) */
extern FILE *ma_log_fp;
#define MA_LOG (ma_log_fp ? ma_log_fp : stderr)
void ma_set_log_path(const char *path);
void ma_set_reads_file(const char *fn); /* reads FASTA/FASTQ for unitig sequences (-f), NULL = none */

/* everything after ingest, hits resident in HBM (pipeline.c) */
int ma_pipeline_device(mahip_ctx_t *c, const ma_opt_t *opt, const sdict_t *d, const char *outfmt, int stage, int flags, FILE *out);
/* its two halves: device passes (st[4] = have_sub, squeezed, n_reduced, graph built) and the host part */
int ma_pipeline_head(mahip_ctx_t *c, const ma_opt_t *opt, const sdict_t *d, const char *outfmt, int stage, int flags, uint32_t st[4]);
int ma_pipeline_tail(mahip_ctx_t *c, const ma_opt_t *opt, const sdict_t *d, const char *outfmt, int stage, const uint32_t st[4], FILE *out);
/* the tail's two halves: fetch = the last use of the device for this input; finish = host only (may run on another thread
 * while the device already works on the next input), frees the job */
typedef struct ma_tail_job ma_tail_job_t;
ma_tail_job_t *ma_pipeline_tail_fetch(mahip_ctx_t *c, const ma_opt_t *opt, const sdict_t *d, const char *outfmt, int stage, const uint32_t st[4]);
int ma_pipeline_tail_finish(ma_tail_job_t *job, FILE *out);
int ma_pipeline_tail_finish_mem(ma_tail_job_t *job, char **buf, size_t *len);

/* the device passes on N GPUs (sharded.c): c carries the hits of this rank's read range and a communicator */
#define MA_SHARD_N_PHASES 16
typedef struct {
	uint64_t n_rem1, n_rem2, n_hits; uint32_t n_seq_new, n_arc, n_loc_arc, n_red, n_multi, n_asymm; uint64_t tie_groups, push_conflicts; int tie_repaired;
	uint32_t n_red_local; int reduced, have_phases; /* this rank's reduced arcs / the counters are sums over the ranks / phase_ms is filled in */
	float phase_ms[MA_SHARD_N_PHASES];        /* device time between the phase marks (ma_shard_phases(1)); names: ma_shard_phase_name[] */
	uint64_t xchg_bytes[MA_SHARD_N_PHASES];   /* bytes every rank receives in the exchange phases */
} ma_shard_stats_t; /* mirrored by miniasm_amd.ShardStats (ctypes) */
extern const char *const ma_shard_phase_name[MA_SHARD_N_PHASES];
void ma_shard_phases(int on);
int ma_pipeline_head_sharded(mahip_ctx_t *c, const ma_opt_t *opt, uint32_t n_seq, int full_input, ma_shard_stats_t *st);
int ma_shard_stats_reduce(mahip_ctx_t *c, ma_shard_stats_t *st);
int ma_pipeline_run_sharded(const ma_opt_t *opt, const char *fn, const char *outfmt, int stage, int flags, FILE *out, int world);
/* one rank's part of it, for a context that already has a communicator of any kind (collective; rank 0 writes `out`) */
int ma_pipeline_run_rank(mahip_ctx_t *c, const ma_opt_t *opt, const char *fn, const char *outfmt, int stage, int flags, FILE *out, int share_gpu);
ma_ug_t *ma_ug_from_device(mahip_ctx_t *c); /* unitigs of the graph resident in c (unitig_gfa.c over csrc/ug.hip) */
void ma_ug_print_mem(const ma_ug_t *ug, const sdict_t *d, const ma_sub_t *sub, char **buf, size_t *len); /* ma_ug_print's text in one malloc'ed block */
void ma_sd_reindex(sdict_t *d);    /* build the name index from seq[] (for dictionaries assembled by hand) */
void ma_sd_drop_index(sdict_t *d);
void ma_sd_fill(sdict_t *d, char *arena, size_t arena_len, uint32_t n_seq, const uint32_t *lens); /* bulk fill; the dictionary owns arena */
void ma_sd_adopt(sdict_t *d, char *arena, size_t arena_len, uint32_t n_seq, sd_seq_t *seq);
int ma_sd_recycle(sdict_t *d, size_t arena_len, uint32_t n_seq, char **arena, sd_seq_t **seq); /* the dictionary's own blocks, if they can hold the next fill */
void *ma_big_alloc(size_t n); /* 2 MiB-aligned, huge pages advised; free() releases it */         /* records ready-made; the dictionary owns both blocks */

/* the process-wide GPU context of the per-symbol entry points; exits with an error if no GPU is usable */
mahip_ctx_t *ma_gpu(void);
void ma_gpu_fail(const char *where); /* prints mahip_strerror() and exits */

int ma_paf_parse_line(int l, char *s, paf_rec_t *pr);

/* ingest without the sort: reference hit.c:70-101 (everything of ma_hit_read before ma_hit_sort) */
ma_hit_t *ma_hit_ingest(const char *fn, int min_span, int min_match, sdict_t *d, size_t *n, int bi_dir, const sdict_t *excl);
/* chunk-parallel variant for plain files (ingest_mt.c); NULL when not eligible (gzip, stdin, small input, one thread) */
ma_hit_t *ma_hit_ingest_mt(const char *fn, int min_span, int min_match, sdict_t *d, size_t *n, int bi_dir, const sdict_t *excl,
                           size_t *tot_lines, uint32_t *max_qs);
/* the same on the device (ingest_gpu.c + csrc/paf.hip): records stay in the context; 0 ok, -1 cannot open */
int ma_hit_ingest_gpu(mahip_ctx_t *c, const char *fn, int min_span, int min_match, sdict_t *d, size_t *n_hits, int bi_dir);
int ma_hit_ingest_gpu_excl(mahip_ctx_t *c, const char *fn, int min_span, int min_match, sdict_t *d, size_t *n_hits, int bi_dir, int no_cont, int max_hang, float int_frac);
/* the same on N ranks, every rank on its own byte range of a plain file (ingest_sharded.c): collective; -2 = not possible for this file (same verdict on every rank) */
typedef struct { uint64_t bytes_own, bytes_file, n_lines, n_records, n_hits_total, bytes_routed, tot_len; uint32_t max_qs; } ma_ingest_shard_info_t;
int ma_ingest_sharded_possible(const char *fn);
int ma_hit_ingest_sharded(mahip_ctx_t *c, const char *fn, int min_span, int min_match, sdict_t *d, size_t *n_hits_total, int bi_dir, ma_ingest_shard_info_t *si);
int ma_hit_ingest_loaded_excl(mahip_ctx_t *c, int min_span, int min_match, sdict_t *d, size_t *n_hits, int bi_dir, int release, int no_cont, int max_hang, float int_frac);
int ma_paf_load_file(mahip_ctx_t *c, const char *fn); /* plain / gzip / "-": text into HBM */
int ma_hit_ingest_loaded(mahip_ctx_t *c, int min_span, int min_match, sdict_t *d, size_t *n_hits, int bi_dir, int release);
int ma_gpu_parse_enabled(void); /* 0 when MA_HOST_PARSE=1 */
int ma_cpu_budget(void);     /* CPUs the process can keep busy: online CPUs cut by the control group's CPU quota */
int ma_ingest_threads(void); /* MA_THREADS or ma_cpu_budget(), capped */
uint32_t ma_ingest_max_qs(void); /* largest query start stored by the last ma_hit_ingest */

/* literal emulation of the reference's in-place MSD radix sort (ksort.h:134-183) on arcs keyed by ul */
void ma_refsort_arcs(asg_arc_t *beg, asg_arc_t *end);
/* the same procedure on (key, input index) pairs, sub-buckets on worker threads (refsort.c): exact-tie mode */
typedef struct { uint64_t key; uint32_t idx, pad; } ma_ki_t;
void ma_refsort_ki(ma_ki_t *a, size_t n, int n_threads);
int ma_refsort_perm(const uint64_t *keys, size_t n, uint32_t *perm); /* perm[i] = input position of the i-th record in reference order */
int ma_refsort_packed(uint64_t *pk, size_t n, int bl, int bi, int shift_top, const uint8_t *dig_top); /* the same on elements the caller packed (key above input position), in place */
int ma_refsort_packed_wanted(uint64_t *pk, size_t n, int bl, int bi, int shift_top, const uint8_t *dig_top, const uint32_t *wcum, uint64_t n_ids); /* ... exact only inside the hit groups of the wanted reads */

#ifdef __cplusplus
}
#endif
#endif
