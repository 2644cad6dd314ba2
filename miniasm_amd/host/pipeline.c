/* pipeline.c -- the whole hot path with the hits resident in HBM between passes.
 *
 * Mirrors the driver logic of the reference's main.c:108-199 (steps, -S stage gates, -p output modes, log
 * lines) but never brings the hits back to the host: ingest (host) -> upload -> sort -> sub/cut/flt ->
 * sub/cut/merge/contained -> ma_sg_gen -> transitive reduction + symm -> tip / bubble / short-overlap / internal / bi-loop
 * cleaning -> unitigs (all HIP, data stays in HBM) -> download of the unitigs -> GFA text (host).
 *
 *   ma_pipeline_device() = ma_pipeline_head() (device passes) + ma_pipeline_tail() (host part); bench.py times
 *   ma_pipeline_device with the unsorted hit records already in HBM.  In the sharded multi-GPU mode the head is
 *   replaced by ma_pipeline_head_sharded (sharded.c: same passes + RCCL exchanges) and rank 0 runs the tail.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <pthread.h>
#include <string.h>
#include "ma_host.h"

#define GPU(call) do { if ((call) != 0) ma_gpu_fail(__func__); } while (0)

FILE *ma_log_fp = 0;
static const char *g_reads_fn = 0;
void ma_set_reads_file(const char *fn) { g_reads_fn = fn; } /* -f of the CLI (reference main.c:193) */

void ma_set_log_path(const char *path)
{
	if (ma_log_fp && ma_log_fp != stderr) fclose(ma_log_fp);
	ma_log_fp = path && *path ? fopen(path, "w") : 0;
}

static void print_subs(const sdict_t *d, const ma_sub_t *sub, FILE *out) /* main.c:13-19 */
{
	uint32_t i;
	for (i = 0; i < d->n_seq; ++i)
		if (!d->seq[i].del && sub[i].s != sub[i].e)
			fprintf(out, "%s\t%d\t%d\n", d->seq[i].name, sub[i].s, sub[i].e);
}

static void print_hits(size_t n_hits, const ma_hit_t *hit, const sdict_t *d, const ma_sub_t *sub, FILE *out) /* main.c:21-30 */
{
	size_t i;
	for (i = 0; i < n_hits; ++i) {
		const ma_hit_t *p = &hit[i];
		const ma_sub_t *rq = &sub[p->qns >> 32], *rt = &sub[p->tn];
		fprintf(out, "%s:%d-%d\t%d\t%d\t%d\t%c\t%s:%d-%d\t%d\t%d\t%d\t%d\t%d\t255\n", d->seq[p->qns >> 32].name, rq->s + 1, rq->e, rq->e - rq->s,
				(uint32_t)p->qns, p->qe, "+-"[p->rev], d->seq[p->tn].name, rt->s + 1, rt->e, rt->e - rt->s, p->ts, p->te, p->ml, p->bl);
	}
}

/* Device passes up to (and including) transitive reduction + symm.  c holds the unsorted hits (uploaded or
 * adopted).  st[0] = have_sub, st[1] = squeezed, st[2] = n_reduced, st[3] = graph built. */
int ma_pipeline_head(mahip_ctx_t *c, const ma_opt_t *opt, const sdict_t *d, const char *outfmt, int stage, int flags, uint32_t st[4])
{
	int no_first = flags & 1, no_second = flags & 2, have_sub = 0, squeezed = 0;
	size_t n_hits = 0, n_rem = 0;
	uint32_t R = d->n_seq, n_seq_new = R, n_red = 0;
	float cov = 40.0f;
	FILE *lg = MA_LOG;

	const int graph_out = strcmp(outfmt, "ug") == 0 || strcmp(outfmt, "sg") == 0;
	const int fused = !no_first && !no_second && stage >= 5 && graph_out && !getenv("MA_NO_FUSE");
	size_t n_cont_hits = 0;

	const int ht = getenv("MA_PIPE_TIMING") && atoi(getenv("MA_PIPE_TIMING")) >= 2; /* per-pass wall time (adds a device sync per pass) */
	double ht0 = sys_realtime();
#define HT(label) do { if (ht) { mahip_sync(c); fprintf(stderr, "[T::head] %-28s %8.3f ms\n", label, (sys_realtime() - ht0) * 1e3); ht0 = sys_realtime(); } } while (0)
	GPU(mahip_hits_sort(c)); /* hit.c:104 */
	HT("sort");
	if (fused) {
		/* Same passes, same log lines, fewer sweeps over the hits: the first ma_hit_cut + ma_hit_flt ride inside the second
		 * coverage pass, the second ma_hit_cut rides with the flag pass of ma_hit_contained, and the squeeze of the hit
		 * array is left to the pass that reads the hits next (ma_sg_gen). */
		size_t n_cut = 0, n_flt = 0, n_rem2 = 0;
		fprintf(lg, "[M::%s] ===> Step 2: 1-pass (crude) read selection <===\n", "main");
		GPU(mahip_hits_sub(c, opt->min_dp, opt->min_iden, 0, 0, &n_rem));
		HT("sub #1");
		if (ma_verbose >= 3) fprintf(lg, "[M::%s::%s] %ld query sequences remain after sub\n", "ma_hit_sub", sys_timestamp(), (long)n_rem);
		GPU(mahip_hits_cutflt_sub(c, 0, opt->min_span, (int)(opt->max_hang * 1.5), (int)(opt->min_ovlp * .5), opt->min_dp, opt->min_iden, opt->min_span / 2,
		                          1, &n_cut, &n_flt, &cov, &n_rem2));
		HT("cut+flt+sub #2");
		if (ma_verbose >= 3) {
			fprintf(lg, "[M::%s::%s] %ld hits remain after cut\n", "ma_hit_cut", sys_timestamp(), (long)n_cut);
			fprintf(lg, "[M::%s::%s] %ld hits remain after filtering; crude coverage after filtering: %.2f\n", "ma_hit_flt", sys_timestamp(), (long)n_flt, cov);
		}
		fprintf(lg, "[M::%s] ===> Step 3: 2-pass (fine) read selection <===\n", "main");
		if (ma_verbose >= 3) fprintf(lg, "[M::%s::%s] %ld query sequences remain after sub\n", "ma_hit_sub", sys_timestamp(), (long)n_rem2);
		GPU(mahip_sub_merge(c)); /* only reads the two interval arrays: may run before the second cut */
		GPU(mahip_hits_cut_contained(c, 1, opt->min_span, opt, &n_hits, &n_seq_new));
		HT("merge + cut + contained");
		if (ma_verbose >= 3) fprintf(lg, "[M::%s::%s] %ld hits remain after cut\n", "ma_hit_cut", sys_timestamp(), (long)n_hits);
		have_sub = squeezed = 1;
	} else {
	if (!no_first) {
		fprintf(lg, "[M::%s] ===> Step 2: 1-pass (crude) read selection <===\n", "main");
		if (stage >= 2) {
			GPU(mahip_hits_sub(c, opt->min_dp, opt->min_iden, 0, 0, &n_rem));
			if (ma_verbose >= 3) fprintf(lg, "[M::%s::%s] %ld query sequences remain after sub\n", "ma_hit_sub", sys_timestamp(), (long)n_rem);
			GPU(mahip_hits_cut(c, 0, opt->min_span, &n_hits));
			if (ma_verbose >= 3) fprintf(lg, "[M::%s::%s] %ld hits remain after cut\n", "ma_hit_cut", sys_timestamp(), (long)n_hits);
			have_sub = 1;
		}
		if (stage >= 3) {
			GPU(mahip_hits_flt(c, 0, (int)(opt->max_hang * 1.5), (int)(opt->min_ovlp * .5), &n_hits, &cov)); /* main.c:125 */
			if (ma_verbose >= 3) fprintf(lg, "[M::%s::%s] %ld hits remain after filtering; crude coverage after filtering: %.2f\n", "ma_hit_flt", sys_timestamp(), (long)n_hits, cov);
		}
	}
	if (!no_second) {
		fprintf(lg, "[M::%s] ===> Step 3: 2-pass (fine) read selection <===\n", "main");
		if (stage >= 4) {
			int slot = no_first ? 0 : 1;
			GPU(mahip_hits_sub(c, opt->min_dp, opt->min_iden, opt->min_span / 2, slot, &n_rem));
			if (ma_verbose >= 3) fprintf(lg, "[M::%s::%s] %ld query sequences remain after sub\n", "ma_hit_sub", sys_timestamp(), (long)n_rem);
			GPU(mahip_hits_cut(c, slot, opt->min_span, &n_hits));
			if (ma_verbose >= 3) fprintf(lg, "[M::%s::%s] %ld hits remain after cut\n", "ma_hit_cut", sys_timestamp(), (long)n_hits);
			if (!no_first) GPU(mahip_sub_merge(c));
			have_sub = 1;
		}
		if (stage >= 5 && have_sub) {
			GPU(mahip_hits_contained(c, opt, 0, &n_seq_new, &n_hits));
			squeezed = 1;
			if (ma_verbose >= 3) fprintf(lg, "[M::%s::%s] %d sequences and %ld hits remain after containment removal\n", "ma_hit_contained", sys_timestamp(), n_seq_new, (long)n_hits);
		}
	}
	}
	st[0] = have_sub, st[1] = squeezed, st[2] = 0, st[3] = 0;
	if (strcmp(outfmt, "ug") == 0 || strcmp(outfmt, "sg") == 0) {
		uint32_t n_arc = 0, *len = 0, r;
		uint8_t *sdel = 0;
		if (!fused) fprintf(lg, "[M::%s] ===> Step 4: graph cleaning <===\n", "main");
		if (!have_sub) {
			len = (uint32_t*)malloc((R ? R : 1) * 4);
			for (r = 0; r < R; ++r) len[r] = d->seq[r].len;
		}
		if (!squeezed) {
			sdel = (uint8_t*)malloc(R ? R : 1);
			for (r = 0; r < R; ++r) sdel[r] = d->seq[r].del;
		}
		GPU(mahip_sg_gen(c, opt, have_sub, len, sdel, &n_arc));
		HT("sg_gen");
		if (getenv("MA_PIPE_TIMING")) { /* what the tie census found and what was done about it */
			mahip_tie_info_t ti;
			mahip_tie_stats(c, &ti);
			char reads[64] = "";
			if (ti.hit_walk && ti.hit_walk_reads) snprintf(reads, sizeof(reads), " (its order taken for %llu reads)", (unsigned long long)ti.hit_walk_reads);
			fprintf(stderr, "[T::ties] %llu arc tie groups (%llu arcs), %llu push conflicts (%llu of them in sight of the arc sort) -> arc walk %d, hit walk %d%s%s\n", (unsigned long long)ti.arc_tie_groups,
			        (unsigned long long)ti.arc_tie_arcs, (unsigned long long)ti.push_conflicts, (unsigned long long)ti.push_conflicts_seen, ti.arc_walk, ti.hit_walk, reads,
			        ti.unrepaired ? " (NOT repaired: stable order kept)" : "");
		}
		free(len); free(sdel);
		if (fused) { /* the hit count after the (postponed) squeeze is known now; keep the reference's line order */
			n_cont_hits = mahip_hits_live(c);
			if (ma_verbose >= 3) fprintf(lg, "[M::%s::%s] %d sequences and %ld hits remain after containment removal\n", "ma_hit_contained", sys_timestamp(), n_seq_new, (long)n_cont_hits);
			fprintf(lg, "[M::%s] ===> Step 4: graph cleaning <===\n", "main");
		}
		fprintf(lg, "[M::%s] read %d arcs\n", "ma_sg_gen", n_arc);
		if (stage >= 6) {
			fprintf(lg, "[M::%s] ===> Step 4.1: transitive reduction <===\n", "main");
			GPU(mahip_asg_del_trans(c, opt->gap_fuzz, &n_red));
			HT("del_trans");
			fprintf(lg, "[M::%s] transitively reduced %d arcs\n", "asg_arc_del_trans", n_red);
			if (n_red) {
				uint32_t n_multi = 0, n_asymm = 0;
				GPU(mahip_asg_symm(c, &n_multi, &n_asymm));
				HT("symm");
				fprintf(lg, "[M::%s] removed %d multi-arcs\n", "asg_arc_del_multi", n_multi);
				fprintf(lg, "[M::%s] removed %d asymmetric arcs\n", "asg_arc_del_asymm", n_asymm);
			}
		}
		st[2] = n_red, st[3] = 1;
	}
	return 0;
}

/* ---- graph cleaning on the device (reference main.c:160-187 over asg.c:83-101, 238-433): same call sequence, same log lines ---- */
static int dev_cut_tip(mahip_ctx_t *c, int max_ext)
{
	uint32_t n = 0;
	GPU(mahip_asg_cut_tip(c, max_ext, &n));
	fprintf(MA_LOG, "[M::%s] cut %d tips\n", "asg_cut_tip", n);
	return (int)n;
}
static int dev_pop_bubble(mahip_ctx_t *c, int max_dist)
{
	uint32_t n = 0, t = 0;
	GPU(mahip_asg_pop_bubble(c, max_dist, &n, &t)); /* the graph is symmetric here: asg_symm ran after the reduction */
	fprintf(MA_LOG, "[M::%s] popped %d bubbles and trimmed %d tips\n", "asg_pop_bubble", n, t);
	return (int)n;
}
static int dev_del_short(mahip_ctx_t *c, float ratio)
{
	uint32_t n = 0;
	GPU(mahip_asg_del_short(c, ratio, &n));
	if (n) { /* asg.c:95-98 */
		uint32_t n_multi = 0, n_asymm = 0;
		GPU(mahip_asg_symm(c, &n_multi, &n_asymm));
		fprintf(MA_LOG, "[M::%s] removed %d multi-arcs\n", "asg_arc_del_multi", n_multi);
		fprintf(MA_LOG, "[M::%s] removed %d asymmetric arcs\n", "asg_arc_del_asymm", n_asymm);
	}
	fprintf(MA_LOG, "[M::%s] removed %d short overlaps\n", "asg_arc_del_short", n);
	return (int)n;
}

static void dev_clean(mahip_ctx_t *c, const ma_opt_t *opt, int stage, int symm_done)
{
	FILE *lg = MA_LOG;
	int i;
	if (stage >= 7) {
		fprintf(lg, "[M::%s] ===> Step 4.2: initial tip cutting and bubble popping <===\n", "main");
		dev_cut_tip(c, opt->max_ext);
		if (!symm_done) { /* asg.c:418: asg_pop_bubble symmetrises a graph that nobody has symmetrised yet */
			uint32_t n_multi = 0, n_asymm = 0;
			GPU(mahip_asg_symm(c, &n_multi, &n_asymm));
			fprintf(lg, "[M::%s] removed %d multi-arcs\n", "asg_arc_del_multi", n_multi);
			fprintf(lg, "[M::%s] removed %d asymmetric arcs\n", "asg_arc_del_asymm", n_asymm);
		}
		dev_pop_bubble(c, opt->bub_dist);
	}
	if (stage >= 9) {
		fprintf(lg, "[M::%s] ===> Step 4.3: cutting short overlaps (%d rounds in total) <===\n", "main", opt->n_rounds + 1);
		for (i = 0; i <= opt->n_rounds; ++i) {
			float r = opt->min_ovlp_drop_ratio + (opt->max_ovlp_drop_ratio - opt->min_ovlp_drop_ratio) / opt->n_rounds * i;
			if (dev_del_short(c, r) != 0) {
				dev_cut_tip(c, opt->max_ext);
				dev_pop_bubble(c, opt->bub_dist);
			}
		}
	}
	if (stage >= 10) {
		uint32_t n = 0;
		fprintf(lg, "[M::%s] ===> Step 4.4: removing short internal sequences and bi-loops <===\n", "main");
		GPU(mahip_asg_cut_internal(c, 1, &n));
		fprintf(lg, "[M::%s] cut %d internal sequences\n", "asg_cut_internal", n);
		GPU(mahip_asg_cut_biloop(c, opt->max_ext, &n));
		fprintf(lg, "[M::%s] cut %d small bi-loops\n", "asg_cut_biloop", n);
		dev_cut_tip(c, opt->max_ext);
		dev_pop_bubble(c, opt->bub_dist);
	}
	if (stage >= 11) {
		fprintf(lg, "[M::%s] ===> Step 4.5: aggressively cutting short overlaps <===\n", "main");
		if (dev_del_short(c, opt->final_ovlp_drop_ratio) != 0) {
			dev_cut_tip(c, opt->max_ext);
			dev_pop_bubble(c, opt->bub_dist);
		}
	}
}

/* The tail in two halves.  ma_pipeline_tail_fetch() finishes the device work -- the graph is renumbered to the surviving reads,
 * cleaned (tips, bubbles, short overlaps, internal sequences, bi-loops) and turned into unitigs, all in HBM -- and brings to the
 * host what the text writer needs: the surviving reads (a shallow view of d: names shared), their kept intervals, and the unitigs
 * (or the cleaned string graph, or the hits for a paf dump).  It is the last thing that touches the device: a caller that
 * processes a stream of inputs can start the next one's device passes right after it.  ma_pipeline_tail_finish() is pure host
 * work: unitig sequences (-f) and the GFA / string-graph / bed / paf text; it frees the job.  d is not modified. */
struct ma_tail_job {
	ma_opt_t opt;
	const sdict_t *d;
	char outfmt[16];
	int stage, have_sub, squeezed, have_graph;
	uint32_t n_red;
	sdict_t view;
	ma_sub_t *sub;
	size_t n_hit;
	ma_hit_t *hit;
	asg_t *sg;
	ma_ug_t *ug;
	double t_fetch[5];
};

/* wall-clock laps (ms) of the most recent tail: [0] survivors' names + intervals to the host, [1] device cleaners, [2] unitigs / graph to the host, [3] text (all of it),
 * [4] of which formatting, [5] of which putting the pieces together (write / copy).  One set per process (bench.py reads it after its closing fence): a report, not a contract */
static double g_tail_laps[8];
double g_fmt_laps[2]; /* unitig_gfa.c: format, write / copy of the last ma_ug_print */
void ma_pipeline_last_laps(double out[8]) { int i; for (i = 0; i < 8; ++i) out[i] = g_tail_laps[i]; }

typedef struct { sd_seq_t *dst; const sd_seq_t *src; const uint32_t *old; uint32_t lo, hi; } view_job_t;
static void *view_worker(void *arg)
{
	view_job_t *v = (view_job_t*)arg;
	uint32_t k;
	for (k = v->lo; k < v->hi; ++k) v->dst[k] = v->src[v->old[k]], v->dst[k].del = 0, v->dst[k].aux = 0;
	return 0;
}

ma_tail_job_t *ma_pipeline_tail_fetch(mahip_ctx_t *c, const ma_opt_t *opt, const sdict_t *d, const char *outfmt, int stage, const uint32_t st[4])
{
	ma_tail_job_t *j = (ma_tail_job_t*)calloc(1, sizeof(ma_tail_job_t));
	const uint32_t R = d->n_seq;
	j->opt = *opt; j->d = d; j->stage = stage;
	if (strlen(outfmt) < sizeof(j->outfmt)) strcpy(j->outfmt, outfmt); /* a longer -p value names no output mode: the empty string matches none either */
	j->have_sub = st[0]; j->squeezed = st[1]; j->n_red = st[2]; j->have_graph = st[3];
	j->t_fetch[0] = sys_realtime();
	j->view.n_seq = R; j->view.seq = d->seq;
	if (j->squeezed) { /* O(survivors): the device hands over the list of surviving old ids */
		uint32_t k, n_new = mahip_n_seq_new(c), *old = (uint32_t*)malloc((n_new ? n_new : 1) * 4);
		GPU(mahip_survivors_download(c, old));
		j->view.seq = (sd_seq_t*)ma_big_alloc(((size_t)n_new + 1) * sizeof(sd_seq_t));
		if (n_new < 200000) {
			for (k = 0; k < n_new; ++k) j->view.seq[k] = d->seq[old[k]], j->view.seq[k].del = 0, j->view.seq[k].aux = 0;
		} else { /* millions of survivors (a graph-heavy input keeps every read): the records are picked by a few threads, each touching its own part of the fresh block */
			view_job_t vj[8];
			pthread_t th[8];
			int t, T = ma_ingest_threads(), ok[8];
			if (T > 8) T = 8;
			if (T < 1) T = 1;
			for (t = 0; t < T; ++t) { vj[t].dst = j->view.seq; vj[t].src = d->seq; vj[t].old = old; vj[t].lo = (uint32_t)((uint64_t)n_new * (uint64_t)t / (uint64_t)T); vj[t].hi = (uint32_t)((uint64_t)n_new * (uint64_t)(t + 1) / (uint64_t)T); }
			for (t = 1; t < T; ++t) { ok[t] = pthread_create(&th[t], 0, view_worker, &vj[t]) == 0; if (!ok[t]) view_worker(&vj[t]); }
			view_worker(&vj[0]);
			for (t = 1; t < T; ++t) if (ok[t]) pthread_join(th[t], 0);
		}
		j->view.n_seq = n_new;
		free(old);
	}
	if (j->have_sub) {
		j->sub = (ma_sub_t*)calloc(j->view.n_seq ? j->view.n_seq : 1, sizeof(ma_sub_t)); /* (a plain block: the runtime's copy into fresh huge-page memory took twice as long and held the other context's launches up) */
		GPU(mahip_sub_download(c, 0, j->sub, j->squeezed));
	}
	j->t_fetch[1] = sys_realtime();
	if (strcmp(outfmt, "paf") == 0) {
		j->n_hit = mahip_hits_live(c);
		j->hit = (ma_hit_t*)malloc((j->n_hit ? j->n_hit : 1) * sizeof(ma_hit_t));
		GPU(mahip_hits_download(c, j->hit, &j->n_hit));
	} else if (strcmp(outfmt, "bed") != 0 && j->have_graph) {
		GPU(mahip_asg_squeeze(c));
		dev_clean(c, opt, stage, j->n_red > 0);
		j->t_fetch[2] = sys_realtime();
		if (strcmp(outfmt, "ug") == 0) {
			fprintf(MA_LOG, "[M::%s] ===> Step 5: generating unitigs <===\n", "main");
			j->ug = ma_ug_from_device(c);
		} else {
			j->sg = asg_init();
			GPU(mahip_asg_download(c, j->sg));
			if (j->sg->n_seq != j->view.n_seq) { fprintf(stderr, "[E::%s] squeeze mismatch: host %u vs device %u reads\n", __func__, j->view.n_seq, j->sg->n_seq); exit(1); }
		}
	}
	j->t_fetch[3] = sys_realtime();
	g_tail_laps[0] = (j->t_fetch[1] - j->t_fetch[0]) * 1e3; g_tail_laps[1] = j->t_fetch[2] > 0 ? (j->t_fetch[2] - j->t_fetch[1]) * 1e3 : 0; g_tail_laps[2] = j->t_fetch[2] > 0 ? (j->t_fetch[3] - j->t_fetch[2]) * 1e3 : 0;
	if (getenv("MA_PIPE_TIMING") && j->have_graph && strcmp(outfmt, "paf") != 0 && strcmp(outfmt, "bed") != 0)
		fprintf(stderr, "[T::tail] names+sub %.3f  device cleaners %.3f  unitigs/graph to host %.3f ms\n", (j->t_fetch[1]-j->t_fetch[0])*1e3, (j->t_fetch[2]-j->t_fetch[1])*1e3, (j->t_fetch[3]-j->t_fetch[2])*1e3);
	return j;
}

/* out != 0: the text goes to the stream; out == 0: into one malloc'ed block (*buf, *len) -- the unitig GFA, tens of MB on a graph-heavy input, is put together by
 * the formatter's own threads (ma_ug_print_mem), everything else through a memory stream */
static int tail_finish_to(ma_tail_job_t *j, FILE *out, char **buf, size_t *len)
{
	const sdict_t *d = j->d;
	const char *outfmt = j->outfmt;
	const int squeezed = j->squeezed;
	sdict_t *view = &j->view;
	ma_sub_t *sub = j->sub;
	const int timing = getenv("MA_PIPE_TIMING") != 0;
	double t0 = sys_realtime();
	FILE *ms = 0;
	if (out == 0 && !(j->ug && strcmp(outfmt, "bed") != 0 && strcmp(outfmt, "paf") != 0)) { ms = open_memstream(buf, len); if (ms == 0) return -1; out = ms; }
	if (strcmp(outfmt, "bed") == 0) {
		if (sub) print_subs(view, sub, out);
	} else if (strcmp(outfmt, "paf") == 0) {
		if (sub) print_hits(j->n_hit, j->hit, view, sub, out);
	} else if (j->ug) {
		if (g_reads_fn) { /* main.c:193; the survivor view needs a name index of its own for sd_get */
			if (squeezed) ma_sd_reindex(view);
			ma_ug_seq(j->ug, squeezed ? view : d, sub, g_reads_fn);
			if (squeezed) ma_sd_drop_index(view);
		}
		if (out) ma_ug_print(j->ug, view, sub, out);
		else ma_ug_print_mem(j->ug, view, sub, buf, len);
	} else if (j->sg) ma_sg_print(j->sg, view, sub, out);
	if (ms) fclose(ms);
	g_tail_laps[3] = (sys_realtime() - t0) * 1e3; g_tail_laps[4] = g_fmt_laps[0]; g_tail_laps[5] = g_fmt_laps[1];
	if (timing) fprintf(stderr, "[T::tail] text %.3f ms\n", (sys_realtime() - t0) * 1e3);
	ma_ug_destroy(j->ug);
	asg_destroy(j->sg);
	free(j->hit);
	free(j->sub);
	if (squeezed) free(j->view.seq);
	free(j);
	return 0;
}

int ma_pipeline_tail_finish(ma_tail_job_t *j, FILE *out) { return tail_finish_to(j, out, 0, 0); }

int ma_pipeline_tail(mahip_ctx_t *c, const ma_opt_t *opt, const sdict_t *d, const char *outfmt, int stage, const uint32_t st[4], FILE *out)
{
	return ma_pipeline_tail_finish(ma_pipeline_tail_fetch(c, opt, d, outfmt, stage, st), out);
}

int ma_pipeline_tail_finish_mem(ma_tail_job_t *j, char **buf, size_t *len) { return tail_finish_to(j, 0, buf, len); }

int ma_pipeline_device(mahip_ctx_t *c, const ma_opt_t *opt, const sdict_t *d, const char *outfmt, int stage, int flags, FILE *out)
{
	uint32_t st[4];
	int rc;
	double t0 = sys_realtime(), t1;
	ma_pipeline_head(c, opt, d, outfmt, stage, flags, st);
	t1 = sys_realtime();
	rc = ma_pipeline_tail(c, opt, d, outfmt, stage, st, out);
	if (getenv("MA_PIPE_TIMING")) fprintf(stderr, "[T::pipeline] head %.3f ms  tail %.3f ms\n", (t1 - t0) * 1e3, (sys_realtime() - t1) * 1e3);
	return rc;
}

/* same, with the output text returned in a malloc'ed buffer (bench.py / tests) */
int ma_pipeline_device_mem(mahip_ctx_t *c, const ma_opt_t *opt, const sdict_t *d, const char *outfmt, int stage, int flags, char **buf, size_t *len)
{
	uint32_t st[4];
	int rc;
	double t0 = sys_realtime(), t1;
	ma_pipeline_head(c, opt, d, outfmt, stage, flags, st);
	t1 = sys_realtime();
	rc = tail_finish_to(ma_pipeline_tail_fetch(c, opt, d, outfmt, stage, st), 0, buf, len);
	if (getenv("MA_PIPE_TIMING")) fprintf(stderr, "[T::pipeline] head %.3f ms  tail %.3f ms\n", (t1 - t0) * 1e3, (sys_realtime() - t1) * 1e3);
	return rc;
}

int ma_pipeline_tail_mem(mahip_ctx_t *c, const ma_opt_t *opt, const sdict_t *d, const char *outfmt, int stage, const uint32_t st[4], char **buf, size_t *len)
{
	return tail_finish_to(ma_pipeline_tail_fetch(c, opt, d, outfmt, stage, st), 0, buf, len);
}

static void *gpu_warmup(void *arg) { (void)arg; (void)ma_gpu(); return 0; }
static void *free_bg(void *p) { free(p); return 0; }

int ma_pipeline_run(const ma_opt_t *opt, const char *fn, const char *outfmt, int stage, int flags, FILE *out)
{
	sdict_t *d = sd_init(), *excl = 0;
	mahip_ctx_t *c;
	pthread_t th_gpu, th_free;
	/* the GPU context comes up (HIP runtime, code objects: 0.1-0.3 s) while the host parses; a machine without a GPU
	 * still fails before any result is produced */
	int gpu_bg = pthread_create(&th_gpu, 0, gpu_warmup, 0) == 0;
	ma_hit_t *hit;
	size_t n_hits = 0;
	FILE *lg = MA_LOG;
	const int dev_parse = ma_gpu_parse_enabled() && strcmp(fn, "-") != 0;
	int on_device = 0;
	if (flags & 8) fprintf(lg, "[M::%s] ===> Step 0: removing contained reads <===\n", "main");
	if (dev_parse) { /* text -> records + dictionary on the device (csrc/paf.hip); with -R the pre-filter rides in the same parse */
		int rc;
		if (gpu_bg) pthread_join(th_gpu, 0), gpu_bg = 0;
		c = ma_gpu();
		if (!(flags & 8)) fprintf(lg, "[M::%s] ===> Step 1: reading read mappings <===\n", "main");
		rc = ma_hit_ingest_gpu_excl(c, fn, opt->min_span, opt->min_match, d, &n_hits, !(flags & 4), (flags & 8) != 0, opt->max_hang, opt->int_frac);
		if (rc == -1) {
			fprintf(stderr, "[E::%s] could not open PAF file %s\n", (flags & 8) ? "ma_hit_no_cont" : "ma_hit_read", fn);
			exit(1);
		}
		on_device = rc == 0; /* -2: does not fit -> host reader below */
	}
	if (!on_device) {
		if (flags & 8) excl = ma_hit_no_cont(fn, opt->min_span, opt->min_match, opt->max_hang, opt->int_frac);
		if (!dev_parse || (flags & 8)) fprintf(lg, "[M::%s] ===> Step 1: reading read mappings <===\n", "main");
	}
	if (!on_device) { /* MA_HOST_PARSE=1, stdin (a stream cannot be re-read after a failed device attempt), or a text too big for the device stage */
		const int timing = getenv("MA_PIPE_TIMING") != 0;
		double t0 = sys_realtime(), t1, t2;
		hit = ma_hit_ingest(fn, opt->min_span, opt->min_match, d, &n_hits, !(flags & 4), excl);
		if (gpu_bg) pthread_join(th_gpu, 0);
		c = ma_gpu();
		t1 = sys_realtime();
		GPU(mahip_set_shard(c, 0, 0xffffffffu));
		GPU(mahip_hits_upload(c, hit, n_hits, d->n_seq));
		GPU(mahip_set_hints(c, ma_ingest_max_qs()));
		GPU(mahip_set_run_stride(c, (flags & 4) ? 1 : 2)); /* the host reader stores a line's record and its mirror side by side unless -b (hit.c:87-98) */
		GPU(mahip_sync(c));
		t2 = sys_realtime();
		if (pthread_create(&th_free, 0, free_bg, hit) == 0) pthread_detach(th_free); /* returning 640 MB to the OS takes 60 ms: off the critical path */
		else free(hit);
		if (timing) fprintf(stderr, "[T::pipeline] ingest %.3f s  upload %.3f s (%.1f GB/s)\n", t1 - t0, t2 - t1, (double)n_hits * 32 / (t2 - t1 + 1e-12) / 1e9);
	}
	ma_pipeline_device(c, opt, d, outfmt, stage, flags, out);
	sd_destroy(d);
	if (excl) sd_destroy(excl);
	return 0;
}
