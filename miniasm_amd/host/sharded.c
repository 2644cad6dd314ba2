/* sharded.c -- the hot path on N GPUs, one process per GPU, orchestrated from C.
 *
 * Rank g owns the reads [g*C, (g+1)*C), C = ceil(R/N), and every hit whose QUERY is one of them; all hit passes are local to a
 * query group except the look-ups of the target's interval / flags and the neighbour look-ups of the reduction, so the
 * exchange points are few and sit between the same fused kernels the single-GPU pipeline uses:
 *
 *   sort | sub #1 | ALL-GATHER sub | cut + flt + sub #2 | ALL-GATHER sub | merge | cut + containment flags
 *   | MAX-ALL-REDUCE (contained, used) | squeeze map (identical on all ranks) | ma_sg_gen flags | MAX-ALL-REDUCE seq.del
 *   | local arcs, sorted | ALL-GATHER OF THE ARC BLOCKS (rank order = global (u,len) order) -> every rank holds graph + CSR
 *   | transitive reduction of the own vertices | ALL-GATHER of the del flags | rank 0: cleanup, symm
 *
 * The collectives are RCCL calls queued on the context's stream (csrc/comm.hip); nothing but the few per-pass counters and
 * the arc-block sizes comes back to the host in between.  This file is the ONE statement of the sequence: tests/test_dist_gloo.py
 * runs it over torch.distributed's gloo backend (mahip_comm_init_ext + miniasm_amd/dist_transport.py) on the CPU build of the kernels.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/wait.h>
#include <sys/prctl.h>
#include <signal.h>
#include "ma_host.h"

#define GPU(call) do { if ((call) != 0) ma_gpu_fail(__func__); } while (0)

/* rank's read range [*q0, *q1) and the length of the longest range (the slot size of the all-gathers): the context's table of hit-balanced
 * ranges when it has one for this world size (mahip_hits_balance), else equal read counts */
static void shard_range(mahip_ctx_t *c, uint32_t n_seq, int world, int rank, uint32_t *per, uint32_t *q0, uint32_t *q1, const uint32_t **bounds)
{
	int bw = 0, r;
	const uint32_t *b = mahip_shard_bounds(c, &bw);
	*bounds = 0;
	if (b && bw == world && b[world] == n_seq) {
		uint32_t longest = 1;
		for (r = 0; r < world; ++r) if (b[r + 1] - b[r] > longest) longest = b[r + 1] - b[r];
		*per = longest; *q0 = b[rank]; *q1 = b[rank + 1]; *bounds = b;
		return;
	}
	{
		uint32_t cc = world > 0 ? (uint32_t)(((uint64_t)n_seq + world - 1) / world) : n_seq;
		uint64_t lo = (uint64_t)rank * cc, hi = lo + cc;
		*per = cc;
		*q0 = (uint32_t)(lo < n_seq ? lo : n_seq);
		*q1 = (uint32_t)(hi < n_seq ? hi : n_seq);
	}
}

/* Per-phase device time of one sharded head (bench.py --gpus N reports the maximum over the ranks): a HIP event on the stream between the
 * phases, read after the step.  Off by default (ma_shard_phases(1) turns it on for the following calls). */
static int g_phases;
void ma_shard_phases(int on) { g_phases = on; }
size_t ma_shard_stats_sizeof(void) { return sizeof(ma_shard_stats_t); } /* tests/test_abi.py checks the ctypes mirror against it */
const char *const ma_shard_phase_name[MA_SHARD_N_PHASES] = {
	"sort", "sub#1", "x:sub0", "cut+flt+sub#2", "x:sub1", "merge+cut+contained", "x:flags", "squeeze+sg flags", "x:seq.del", "local arcs",
	"x:arc counts", "x:arc blocks", "tie repair", "reduction (own vertices)", "x:del flags", "rank 0: cleanup+symm" };
#define MARK(k) do { if (g_phases) GPU(mahip_mark(c, (k))); } while (0)

/* all-gather of the owned slices of one of the read-indexed arrays (element size es): afterwards every rank holds all n_seq entries */
static size_t exchange_slices(mahip_ctx_t *c, int which, size_t es, uint32_t n_seq, uint32_t per, uint32_t q0, uint32_t q1, int world, const uint32_t *bounds)
{
	char *loc, *all;
	int r;
	if (!mahip_comm_active(c)) return 0;
	GPU(mahip_xbuf(c, 0, (size_t)per * es, (void**)&loc));
	GPU(mahip_xbuf(c, 1, (size_t)per * es * world, (void**)&all));
	GPU(mahip_copy_out(c, which, loc, q0, q1 - q0));
	GPU(mahip_comm_all_gather(c, loc, all, (size_t)per * es));
	if (bounds == 0) GPU(mahip_copy_in(c, which, all, 0, n_seq)); /* equal ranges: the gathered slots ARE the array */
	else for (r = 0; r < world; ++r) GPU(mahip_copy_in(c, which, all + (size_t)r * per * es, bounds[r], bounds[r + 1] - bounds[r]));
	return (size_t)per * es * world;
}

/* OR of 0/1 byte flag arrays over the ranks = one max-all-reduce over their concatenation */
static size_t exchange_flags(mahip_ctx_t *c, uint32_t n_seq, int n_which, const int *which)
{
	char *t;
	int k;
	if (!mahip_comm_active(c)) return 0;
	GPU(mahip_xbuf(c, 0, (size_t)n_seq * n_which, (void**)&t));
	for (k = 0; k < n_which; ++k) GPU(mahip_copy_out(c, which[k], t + (size_t)k * n_seq, 0, n_seq));
	GPU(mahip_comm_all_reduce_max_u8(c, t, (size_t)n_seq * n_which));
	for (k = 0; k < n_which; ++k) GPU(mahip_copy_in(c, which[k], t + (size_t)k * n_seq, 0, n_seq));
	return (size_t)n_seq * n_which;
}

/* The device passes of one input on this rank's shard, up to the reduced graph.  c holds the unsorted hits of (at least) this
 * rank's read range and a communicator (mahip_comm_init*).  Afterwards rank 0's context holds the reduced, symmetrised graph.
 * Host round trips between the ranks: ONE (the sizes of the arc blocks); the counters of the passes stay local until
 * ma_shard_stats_reduce() is called -- after the step, off the critical path. */
int ma_pipeline_head_sharded(mahip_ctx_t *c, const ma_opt_t *opt, uint32_t n_seq, int full_input, ma_shard_stats_t *st)
{ /* full_input: c holds ALL hit records of the input (every rank parsed the text), not just this rank's: the tie repair can then also restore the hit order */
	const int world = mahip_comm_world(c), rank = mahip_comm_rank(c), active = mahip_comm_active(c);
	uint32_t per, q0, q1, n_loc = 0, n_red = 0, n_seq_new = 0, i;
	size_t n_rem1 = 0, n_rem2 = 0, n_cut = 0, n_flt = 0, n_hits = 0;
	float cov = 0;
	uint64_t cnt[64];
	uint32_t *counts = (uint32_t*)calloc((size_t)world + 1, 4);
	size_t stride = 1, first = 0, tot = 0;
	const uint32_t *bounds = 0;
	memset(st, 0, sizeof(*st));
	shard_range(c, n_seq, world, rank, &per, &q0, &q1, &bounds);
	GPU(mahip_set_full_input(c, full_input || world == 1));
	GPU(mahip_set_shard(c, world > 1 ? q0 : 0, world > 1 ? q1 : 0xffffffffu));
	MARK(0);
	GPU(mahip_hits_sort(c));
	MARK(1);
	GPU(mahip_hits_sub(c, opt->min_dp, opt->min_iden, 0, 0, &n_rem1));
	MARK(2);
	st->xchg_bytes[2] = exchange_slices(c, MAHIP_BUF_SUB0, 8, n_seq, per, q0, q1, world, bounds);
	MARK(3);
	GPU(mahip_hits_cutflt_sub(c, 0, opt->min_span, (int)(opt->max_hang * 1.5), (int)(opt->min_ovlp * .5), opt->min_dp, opt->min_iden, opt->min_span / 2,
	                          1, &n_cut, &n_flt, &cov, &n_rem2)); /* hit.c:162-216 + the second ma_hit_sub: needs the complete first-pass intervals */
	MARK(4);
	st->xchg_bytes[4] = exchange_slices(c, MAHIP_BUF_SUB1, 8, n_seq, per, q0, q1, world, bounds);
	MARK(5);
	GPU(mahip_sub_merge(c)); /* on the complete arrays: identical on every rank */
	GPU(mahip_hits_cut_contained_flags(c, 1, opt->min_span, opt));
	MARK(6);
	{ const int w[2] = { MAHIP_BUF_RCONT, MAHIP_BUF_RUSED }; st->xchg_bytes[6] = exchange_flags(c, n_seq, 2, w); }
	MARK(7);
	GPU(mahip_hits_cut_contained_finish(c, &n_cut, &n_seq_new));
	GPU(mahip_sg_flags(c, opt, 1, 0, 0));
	MARK(8);
	{ const int w[1] = { MAHIP_BUF_SDEL }; st->xchg_bytes[8] = exchange_flags(c, n_seq, 1, w); }
	MARK(9);
	GPU(mahip_sg_finish(c, &n_loc));
	n_hits = mahip_hits_live(c);
	MARK(10);
	/* the arc all-gather: block sizes first (one counter per rank), then the blocks padded to the largest */
	memset(cnt, 0, sizeof(cnt));
	if (world > 32) { fprintf(stderr, "[E::%s] at most 32 ranks\n", __func__); exit(1); }
	cnt[rank] = n_loc;
	GPU(mahip_comm_all_reduce_sum_u64(c, cnt, (size_t)world));
	for (i = 0; i < (uint32_t)world; ++i) { counts[i] = (uint32_t)cnt[i]; if (cnt[i] > stride) stride = cnt[i]; if ((int)i < rank) first += cnt[i]; tot += cnt[i]; }
	MARK(11);
	if (active) {
		void *rows, *all;
		GPU(mahip_xbuf(c, 0, stride * 16, &rows));
		GPU(mahip_xbuf(c, 1, stride * 16 * world, &all));
		GPU(mahip_asg_export_rows(c, rows));
		GPU(mahip_comm_all_gather(c, rows, all, stride * 16));
		GPU(mahip_asg_import_rows(c, all, counts, world, stride));
		st->xchg_bytes[11] = stride * 16 * world;
	}
	MARK(12);
	if (active) { /* tie order: the census ran on the merged graph; groups of equal (u,len) keys -> the reference's order */
		mahip_tie_info_t ti;
		mahip_tie_stats(c, &ti);
		st->tie_groups = ti.arc_tie_groups;
		if (ti.unrepaired) {
			uint64_t conf = 0, two[2];
			GPU(mahip_sg_push_conflicts(c, &conf));
			two[0] = conf; two[1] = mahip_hits_have_positions(c) ? 1 : 0;
			GPU(mahip_comm_all_reduce_sum_u64(c, two, 2));
			conf = two[0];
			st->push_conflicts = conf;
			if (conf == 0 || full_input || world == 1 || two[1] == (uint64_t)world) { /* own-records shards: only when EVERY rank knows where its records stood in the input */
				void *rows, *all;
				if (conf) GPU(mahip_sg_push_fix(c)); /* every rank walks the hit keys of the whole input (own-records shards: gathered from all ranks) and keeps the ranks of its own hits */
				GPU(mahip_xbuf(c, 0, stride * 16, &rows));
				GPU(mahip_xbuf(c, 1, stride * 16 * world, &all));
				GPU(mahip_asg_export_rows_push(c, rows));
				GPU(mahip_comm_all_gather(c, rows, all, stride * 16));
				GPU(mahip_asg_import_push_rows(c, all, counts, world, stride)); /* the same walk over the same global push sequence on every rank */
				st->tie_repaired = 1;
			}
		}
	} else { mahip_tie_info_t ti; mahip_tie_stats(c, &ti); st->tie_groups = ti.arc_tie_groups; st->tie_repaired = ti.arc_walk; }
	MARK(13);
	GPU(mahip_asg_del_trans_range(c, opt->gap_fuzz, 2 * q0, world > 1 ? 2 * q1 : 2 * n_seq, &n_red));
	MARK(14);
	if (active) { /* the del flags of the own block -> everyone (only rank 0 needs them; kept symmetric) */
		char *fl, *all;
		size_t off = 0;
		GPU(mahip_xbuf(c, 0, stride * 4, (void**)&fl));
		GPU(mahip_xbuf(c, 1, stride * 4 * world, (void**)&all));
		GPU(mahip_asg_flags_out(c, fl, first, n_loc));
		GPU(mahip_comm_all_gather(c, fl, all, stride * 4));
		for (i = 0; i < (uint32_t)world; ++i) {
			if ((int)i != rank && counts[i]) GPU(mahip_asg_flags_in(c, all + (size_t)i * stride * 4, off, counts[i]));
			off += counts[i];
		}
		st->xchg_bytes[14] = stride * 4 * world;
	}
	MARK(15);
	st->n_rem1 = n_rem1; st->n_rem2 = n_rem2; st->n_hits = n_hits; st->n_red = st->n_red_local = n_red; /* this rank's share: ma_shard_stats_reduce() sums them */
	st->n_seq_new = n_seq_new; st->n_arc = (uint32_t)tot; st->n_loc_arc = n_loc;
	if (rank == 0) { /* asg.c:187-190: cleanup + symm when anything was reduced.  Rank 0 holds every rank's del flags: what the cleanup removes IS the global count */
		uint32_t n_arc = 0;
		if (world > 1) {
			GPU(mahip_asg_cleanup(c, &n_arc));
			st->n_red = (uint32_t)(tot - n_arc);
			if (st->n_red) GPU(mahip_asg_symm(c, &st->n_multi, &st->n_asymm));
		} else if (st->n_red) {
			GPU(mahip_asg_cleanup(c, &n_arc));
			GPU(mahip_asg_symm(c, &st->n_multi, &st->n_asymm));
		}
	}
	MARK(16);
	if (g_phases) {
		GPU(mahip_sync(c));
		GPU(mahip_marks_ms(c, 0, MA_SHARD_N_PHASES, st->phase_ms));
		st->have_phases = 1;
	}
	free(counts);
	return 0;
}

/* the counters of the passes, summed over the ranks (log lines, statistics): a collective every rank must call; not part of the step */
int ma_shard_stats_reduce(mahip_ctx_t *c, ma_shard_stats_t *st)
{
	uint64_t sums[4];
	if (st->reduced) return 0;
	sums[0] = st->n_rem1; sums[1] = st->n_rem2; sums[2] = st->n_hits; sums[3] = st->n_red_local;
	GPU(mahip_comm_all_reduce_sum_u64(c, sums, 4));
	st->n_rem1 = sums[0]; st->n_rem2 = sums[1]; st->n_hits = sums[2];
	/* n_red has ONE meaning: the arcs the reduction deleted.  Rank 0 measured it as what its cleanup removed (it holds every rank's del flags), the ranks as the sum of
	 * what each deleted among its own vertices; the two must agree -- an arc deleted twice, or deleted by something else before the cleanup, would make the log line and
	 * the later "anything reduced?" gate disagree with what rank 0 did */
	if (mahip_comm_rank(c) == 0 && mahip_comm_world(c) > 1 && st->n_red != (uint32_t)sums[3]) {
		fprintf(stderr, "[E::%s] the ranks reduced %lu arcs, rank 0's cleanup removed %u\n", __func__, (unsigned long)sums[3], st->n_red);
		return -1;
	}
	st->n_red = (uint32_t)sums[3];
	st->reduced = 1;
	return 0;
}

/* What ONE rank of an N-rank run does once its context has a communicator (RCCL, the shared-memory double, or a transport of the caller's:
 * mahip_comm_init_ext): ingest -- its own byte range of a plain file, else the whole text --, the sharded head, and on rank 0 the tail into `out`.
 * Collective.  ma_pipeline_run_sharded() below calls it in every process it forked; tests/test_dist_gloo.py calls it on ranks that torch.distributed
 * started, over gloo.  share_gpu: the ranks sit on one device (test set-ups): idle pool memory goes back to the driver before the tail. */
int ma_pipeline_run_rank(mahip_ctx_t *c, const ma_opt_t *opt, const char *fn, const char *outfmt, int stage, int flags, FILE *out, int share_gpu)
{
	const int world = mahip_comm_world(c), rank = mahip_comm_rank(c);
	int r, own_records = 0;
	sdict_t *d = sd_init();
	size_t n_hits = 0;
	ma_shard_stats_t st;
	uint32_t pst[4];
	FILE *lg;
	lg = MA_LOG;
	fprintf(lg, "[M::%s] ===> Step %d: %s <===\n", "main", (flags & 8) ? 0 : 1, (flags & 8) ? "removing contained reads" : "reading read mappings");
	/* Ingest.  A plain file without -R: every rank loads and parses its own byte range, the ranks merge their name tables and route the records to the owners
	 * of their query reads (ingest_sharded.c) -- 1/N of the text per rank.  Otherwise (gzip, stdin, -R, MA_INGEST_WHOLE=1): every rank parses the whole text
	 * and keeps the hits of its read range, as in round 3.  The choice depends on the file and the options alone: every rank makes the same one. */
	own_records = !(flags & 8) && !(getenv("MA_INGEST_WHOLE") && atoi(getenv("MA_INGEST_WHOLE"))) && ma_ingest_sharded_possible(fn);
	if (own_records) {
		ma_ingest_shard_info_t si;
		memset(&si, 0, sizeof(si));
		r = ma_hit_ingest_sharded(c, fn, opt->min_span, opt->min_match, d, &n_hits, !(flags & 4), &si);
		if (r == -1) { fprintf(stderr, "[E::%s] could not open PAF file %s\n", "ma_hit_read", fn); exit(1); }
		if (r != 0) { fprintf(stderr, "[E::%s] rank %d: the ranges of the text could not be ingested\n", "ma_pipeline_run_sharded", rank); exit(1); }
		if (ma_verbose >= 3)
			fprintf(lg, "[M::%s::%s] read %ld hits; stored %ld hits and %d sequences (%ld bp)\n", "ma_hit_read", sys_timestamp(), (long)si.n_records, (long)si.n_hits_total, d->n_seq, (long)si.tot_len);
	} else {
	r = ma_hit_ingest_gpu_excl(c, fn, opt->min_span, opt->min_match, d, &n_hits, !(flags & 4), (flags & 8) != 0, opt->max_hang, opt->int_frac);
	if (r == -1) { fprintf(stderr, "[E::%s] could not open PAF file %s\n", "ma_hit_read", fn); exit(1); }
	if (r != 0) { fprintf(stderr, "[E::%s] the text does not fit the device stage; MA_GPUS > 1 needs the device parser\n", "ma_pipeline_run_sharded"); exit(1); }
	}
	{ /* test hook: a rank that dies in the middle of a run (tests/test_gpu_sharded.py checks that nobody is left waiting) */
		const char *e = getenv("MA_TEST_FAIL_RANK");
		if (e && atoi(e) == rank) { fprintf(stderr, "[E::%s] rank %d: MA_TEST_FAIL_RANK\n", "ma_pipeline_run_sharded", rank); _exit(3); }
	}
	if (!own_records) GPU(mahip_hits_balance(c, world, 0)); /* every rank holds the whole input here: the same hit-balanced read ranges everywhere (own records: mahip_hits_route made them) */
	ma_pipeline_head_sharded(c, opt, d->n_seq, !own_records, &st);
	if (ma_shard_stats_reduce(c, &st) != 0) exit(1); /* the log lines below want the sums */
	if (share_gpu) GPU(mahip_mem_trim(c, 0)); /* the ranks share ONE GPU here: what this rank's pool keeps idle (the text, the parser's columns) is memory rank 0's tail cannot have */
	if (rank == 0) {
		fprintf(lg, "[M::%s] ===> Step 2: 1-pass (crude) read selection <===\n", "main");
		if (ma_verbose >= 3) fprintf(lg, "[M::%s::%s] %ld query sequences remain after sub\n", "ma_hit_sub", sys_timestamp(), (long)st.n_rem1);
		fprintf(lg, "[M::%s] ===> Step 3: 2-pass (fine) read selection <===\n", "main");
		if (ma_verbose >= 3) {
			fprintf(lg, "[M::%s::%s] %ld query sequences remain after sub\n", "ma_hit_sub", sys_timestamp(), (long)st.n_rem2);
			fprintf(lg, "[M::%s::%s] %d sequences and %ld hits remain after containment removal\n", "ma_hit_contained", sys_timestamp(), st.n_seq_new, (long)st.n_hits);
		}
		fprintf(lg, "[M::%s] ===> Step 4: graph cleaning <===\n", "main");
		fprintf(lg, "[M::%s] read %d arcs\n", "ma_sg_gen", st.n_arc);
		if (st.tie_groups && !st.tie_repaired)
			fprintf(stderr, "[W::%s] %llu groups of arcs with equal (u,len) keys were left in the stable order: the output may differ from the reference's inside those groups\n",
			        "ma_pipeline_run_sharded", (unsigned long long)st.tie_groups);
		fprintf(lg, "[M::%s] ===> Step 4.1: transitive reduction <===\n", "main");
		fprintf(lg, "[M::%s] transitively reduced %d arcs\n", "asg_arc_del_trans", st.n_red);
		if (st.n_red) {
			fprintf(lg, "[M::%s] removed %d multi-arcs\n", "asg_arc_del_multi", st.n_multi);
			fprintf(lg, "[M::%s] removed %d asymmetric arcs\n", "asg_arc_del_asymm", st.n_asymm);
		}
		pst[0] = 1; pst[1] = 1; pst[2] = st.n_red; pst[3] = 1;
		ma_pipeline_tail(c, opt, d, outfmt, stage, pst, out);
	}
	GPU(mahip_comm_barrier(c));
	sd_destroy(d);
	return 0;
}

/* ---- the command line on N GPUs: MA_GPUS=N miniasm in.paf > out.gfa ------------------------------------------------------
 * The parent is rank 0; it forks N-1 children BEFORE any HIP call and hands them the RCCL id through pipes.  Every rank loads
 * and parses the whole text on its own GPU (its own PCIe link; the device parser makes this cheaper than routing records:
 * DESIGN section 6) and keeps the hits of its read range; rank 0 cleans the graph and writes the output, the others leave
 * after the last collective.  MA_COMM=shm selects the host-staged test double (all ranks on MA_GPU_DEVICE / device 0). */
/* A rank that fails must not leave the others waiting in a collective: children die with the parent (PR_SET_PDEATHSIG), the parent
 * ends the run when a child ends abnormally (SIGCHLD) and takes the remaining children with it when it leaves early (atexit). */
static pid_t *g_kids;
static int g_n_kids;
static volatile sig_atomic_t g_kids_done;

static void kids_kill(void)
{
	int r;
	for (r = 1; r < g_n_kids; ++r) if (g_kids && g_kids[r] > 0) kill(g_kids[r], SIGKILL);
}

static void on_sigchld(int sig)
{
	int status, r;
	pid_t p;
	(void)sig;
	while ((p = waitpid(-1, &status, WNOHANG)) > 0) {
		for (r = 1; r < g_n_kids; ++r) if (g_kids[r] == p) break;
		if (r == g_n_kids) continue; /* not a rank */
		g_kids[r] = 0;
		if (!g_kids_done && !(WIFEXITED(status) && WEXITSTATUS(status) == 0)) {
			static const char msg[] = "[E::ma_pipeline_run_sharded] a rank ended abnormally; stopping the run\n";
			if (write(2, msg, sizeof(msg) - 1) < 0) {}
			kids_kill();
			_exit(1);
		}
	}
}

int ma_pipeline_run_sharded(const ma_opt_t *opt, const char *fn, const char *outfmt, int stage, int flags, FILE *out, int world)
{
	const char *kind = getenv("MA_COMM");
	const int use_shm = kind && strcmp(kind, "shm") == 0;
	int rank = 0, r, (*pipes)[2] = (int(*)[2])calloc((size_t)world, sizeof(int[2]));
	pid_t *kids = (pid_t*)calloc((size_t)world, sizeof(pid_t));
	char id[128], shm_name[64];
	mahip_ctx_t *c;
	/* The sharded head is the full graph path (both read selections, containment, graph, reduction).  Any other request -- a hit dump, an
	 * early -S stage, -1 / -2 -- is decided BEFORE the ranks exist and runs on one GPU: the output is the same, only not spread out. */
	if ((strcmp(outfmt, "ug") != 0 && strcmp(outfmt, "sg") != 0) || (flags & 3) || stage < 6) {
		fprintf(stderr, "[W::%s] MA_GPUS=%d serves -p ug / -p sg with both read selections and -S >= 6; this request runs on one GPU\n", __func__, world);
		free(pipes); free(kids);
		return ma_pipeline_run(opt, fn, outfmt, stage, flags, out);
	}
	if (world > 32) { fprintf(stderr, "[E::%s] at most 32 ranks\n", __func__); exit(1); }
	memset(id, 0, sizeof(id));
	snprintf(shm_name, sizeof(shm_name), "miniasm_amd_%d", (int)getpid());
	fflush(stdout); fflush(stderr);
	g_kids = kids; g_n_kids = world; g_kids_done = 0;
	{
		struct sigaction sa;
		memset(&sa, 0, sizeof(sa));
		sa.sa_handler = on_sigchld;
		sa.sa_flags = SA_RESTART | SA_NOCLDSTOP;
		sigaction(SIGCHLD, &sa, 0);
		atexit(kids_kill);
	}
	{ /* a child that ends before its pid is in kids[] must not be taken for "not a rank": no SIGCHLD until the table is complete */
		sigset_t blk;
		sigemptyset(&blk); sigaddset(&blk, SIGCHLD);
		sigprocmask(SIG_BLOCK, &blk, 0);
	}
	for (r = 1; r < world; ++r) {
		const pid_t parent = getpid();
		if (pipe(pipes[r]) != 0) { perror("pipe"); exit(1); }
		kids[r] = fork();
		if (kids[r] < 0) { perror("fork"); exit(1); }
		if (kids[r] == 0) {
			rank = r; close(pipes[r][1]);
			g_kids = 0; g_n_kids = 0; signal(SIGCHLD, SIG_DFL);
			prctl(PR_SET_PDEATHSIG, SIGKILL);
			if (getppid() != parent) _exit(1); /* the parent left between fork and prctl */
			break;
		}
		close(pipes[r][0]);
	}
	{
		sigset_t blk;
		sigemptyset(&blk); sigaddset(&blk, SIGCHLD);
		sigprocmask(SIG_UNBLOCK, &blk, 0); /* (the children inherit the blocked mask: they have no children of their own) */
	}
	if (rank == 0) {
		if (!use_shm) GPU(mahip_comm_unique_id(id));
		for (r = 1; r < world; ++r) { if (write(pipes[r][1], id, sizeof(id)) != (ssize_t)sizeof(id)) { perror("write"); exit(1); } close(pipes[r][1]); }
	} else {
		if (read(pipes[rank][0], id, sizeof(id)) != (ssize_t)sizeof(id)) { fprintf(stderr, "[E::%s] rank %d: no id from rank 0\n", __func__, rank); _exit(1); }
		close(pipes[rank][0]);
		ma_set_log_path("/dev/null"); /* one copy of the log lines: rank 0's */
	}
	{
		char dev[16];
		const char *base = getenv("MA_GPU_DEVICE");
		snprintf(dev, sizeof(dev), "%d", use_shm ? (base ? atoi(base) : 0) : (base ? atoi(base) : 0) + rank);
		setenv("MA_GPU_DEVICE", dev, 1);
	}
	c = ma_gpu();
	if (use_shm) GPU(mahip_comm_init_shm(c, shm_name, rank, world));
	else GPU(mahip_comm_init(c, id, rank, world));
	ma_pipeline_run_rank(c, opt, fn, outfmt, stage, flags, out, use_shm);
	mahip_comm_destroy(c);
	if (rank != 0) exit(0); /* orderly: the context's atexit teardown runs */
	g_kids_done = 1; /* the work is done: from here a rank's exit status is only reported */
	for (r = 1; r < world; ++r) {
		int status = 0;
		const pid_t k = kids[r];
		if (k > 0 && waitpid(k, &status, 0) == k && (!WIFEXITED(status) || WEXITSTATUS(status) != 0)) fprintf(stderr, "[W::%s] rank %d ended abnormally\n", __func__, r);
		kids[r] = 0;
	}
	signal(SIGCHLD, SIG_DFL);
	g_kids = 0; g_n_kids = 0;
	free(pipes); free(kids);
	return 0;
}
