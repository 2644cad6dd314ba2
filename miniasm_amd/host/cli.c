/* main.c -- the miniasm command line (PAF in -> GFA out) over the MI355X hot path.
 * Same option string, defaults, usage text layout, -S/-p behaviour and closing log lines as the
 * reference driver (main.c:32-211); extra knobs come from the environment (MA_GPU_DEVICE) so the option
 * string stays a drop-in.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "ma_host.h"

#define MA_VERSION "0.3-r179" /* output-compatible with this reference version */
#define MA_BUILD "amd-gfx950"

static void usage(const ma_opt_t *o, const char *outfmt)
{
	FILE *f = stderr;
	fprintf(f, "Usage: miniasm [options] <in.paf>\n");
	fprintf(f, "Options:\n");
	fprintf(f, "  Pre-selection:\n");
	fprintf(f, "    -R          prefilter clearly contained reads (2-pass required)\n");
	fprintf(f, "    -m INT      min match length [%d]\n", o->min_match);
	fprintf(f, "    -i FLOAT    min identity [%.2g]\n", o->min_iden);
	fprintf(f, "    -s INT      min span [%d]\n", o->min_span);
	fprintf(f, "    -c INT      min coverage [%d]\n", o->min_dp);
	fprintf(f, "  Overlap:\n");
	fprintf(f, "    -o INT      min overlap [same as -s]\n");
	fprintf(f, "    -h INT      max over hang length [%d]\n", o->max_hang);
	fprintf(f, "    -I FLOAT    min end-to-end match ratio [%.2g]\n", o->int_frac);
	fprintf(f, "  Layout:\n");
	fprintf(f, "    -g INT      max gap differences between reads for trans-reduction [%d]\n", o->gap_fuzz);
	fprintf(f, "    -d INT      max distance for bubble popping [%d]\n", o->bub_dist);
	fprintf(f, "    -e INT      small unitig threshold [%d]\n", o->max_ext);
	fprintf(f, "    -f FILE     read sequences []\n");
	fprintf(f, "    -n INT      rounds of short overlap removal [%d]\n", o->n_rounds + 1);
	fprintf(f, "    -r FLOAT[,FLOAT]\n");
	fprintf(f, "                max and min overlap drop ratio [%.2g,%.2g]\n", o->max_ovlp_drop_ratio, o->min_ovlp_drop_ratio);
	fprintf(f, "    -F FLOAT    aggressive overlap drop ratio in the end [%.2g]\n", o->final_ovlp_drop_ratio);
	fprintf(f, "  Miscellaneous:\n");
	fprintf(f, "    -p STR      output information: bed, paf, sg or ug [%s]\n", outfmt);
	fprintf(f, "    -b          both directions of an arc are present in input\n");
	fprintf(f, "    -1          skip 1-pass read selection\n");
	fprintf(f, "    -2          skip 2-pass read selection\n");
	fprintf(f, "    -V          print version number\n");
	fprintf(f, "\nSee miniasm.1 for detailed description of the command-line options.\n");
}

int main(int argc, char *argv[])
{
	ma_opt_t opt;
	int c, i, stage = 100, flags = 0, o_set = 0;
	const char *outfmt = "ug", *fn_reads = 0;

	ma_opt_init(&opt);
	while ((c = getopt(argc, argv, "n:m:s:c:S:i:d:g:o:h:I:r:f:e:p:12VBRbF:")) >= 0) {
		switch (c) {
		case 'm': opt.min_match = atoi(optarg); break;
		case 'i': opt.min_iden = atof(optarg); break;
		case 's': opt.min_span = atoi(optarg); break;
		case 'c': opt.min_dp = atoi(optarg); break;
		case 'o': opt.min_ovlp = atoi(optarg), o_set = 1; break;
		case 'S': stage = atoi(optarg); break;
		case 'd': opt.bub_dist = atoi(optarg); break;
		case 'g': opt.gap_fuzz = atoi(optarg); break;
		case 'h': opt.max_hang = atoi(optarg); break;
		case 'I': opt.int_frac = atof(optarg); break;
		case 'e': opt.max_ext = atoi(optarg); break;
		case 'f': fn_reads = optarg; break;
		case 'p': outfmt = optarg; break;
		case '1': flags |= 1; break;
		case '2': flags |= 2; break;
		case 'n': opt.n_rounds = atoi(optarg) - 1; break;
		case 'B': flags &= ~4; break;
		case 'b': flags |= 4; break;
		case 'R': flags |= 8; break;
		case 'F': opt.final_ovlp_drop_ratio = atof(optarg); break;
		case 'V': printf("%s\n", MA_VERSION); return 0;
		case 'r': {
			char *s;
			opt.max_ovlp_drop_ratio = strtod(optarg, &s);
			if (*s == ',') opt.min_ovlp_drop_ratio = strtod(s + 1, &s);
			break;
		}
		default: break;
		}
	}
	if (o_set == 0) opt.min_ovlp = opt.min_span;
	if (argc == optind) { usage(&opt, outfmt); return 1; }

	sys_init();
	ma_set_reads_file(fn_reads); /* -f: unitig sequences are stitched from this file before the GFA is written */
	{
		const char *g = getenv("MA_GPUS"); /* N > 1: one process per GPU, read-range shards, RCCL exchanges (host/sharded.c) */
		const int world = g ? atoi(g) : 1;
		const char *one = getenv("MA_RCCL_ONE_RANK"); /* with MA_GPUS=1: the sharded runner on a one-rank RCCL communicator, every collective really called */
		if (world > 1 || (g && world == 1 && one && atoi(one) != 0)) ma_pipeline_run_sharded(&opt, argv[optind], outfmt, stage, flags, stdout, world);
		else ma_pipeline_run(&opt, argv[optind], outfmt, stage, flags, stdout);
	}

	fprintf(stderr, "[M::%s] Version: %s (%s)\n", __func__, MA_VERSION, MA_BUILD);
	fprintf(stderr, "[M::%s] CMD:", __func__);
	for (i = 0; i < argc; ++i) fprintf(stderr, " %s", argv[i]);
	fprintf(stderr, "\n[M::%s] Real time: %.3f sec; CPU: %.3f sec\n", __func__, sys_realtime(), sys_cputime());
	/* The result is complete.  What a normal return would still do -- hand every device buffer back one hipFree at a time (tens of GB at BASELINE configs[3]: each unmaps),
	 * then the HIP runtime's own exit handlers -- is 0.1 s and more of a 0.5 s run and produces nothing: the driver reclaims a process's memory when it is gone.
	 * So: flush, report a failed write as the exit status, and leave.  MA_CLEAN_EXIT=1 takes the long road (leak checks, tools that want the teardown exercised). */
	{
		int bad = fflush(stdout) != 0 || ferror(stdout);
		fflush(stderr);
		if (bad) { fprintf(stderr, "[E::%s] could not write the output\n", __func__); _exit(1); }
		if (getenv("MA_CLEAN_EXIT") == 0) _exit(0);
	}
	return 0;
}
