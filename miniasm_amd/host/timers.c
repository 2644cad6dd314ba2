/* sys.c -- wall/CPU timers and the "real*cpu/real" stamp used in the [M::fn::stamp] log lines.
 * Same interface and text format as the reference's sys.h:8-11 / sys.c:38-46 (logs double as checksums). */
#include <stdio.h>
#include <sys/time.h>
#include <sys/resource.h>
#include "miniasm_amd.h"

static double t_origin;

static double wall_now(void)
{
	struct timeval tv;
	gettimeofday(&tv, 0);
	return tv.tv_sec + 1e-6 * tv.tv_usec;
}

double sys_cputime(void)
{
	struct rusage ru;
	getrusage(RUSAGE_SELF, &ru);
	return (ru.ru_utime.tv_sec + ru.ru_stime.tv_sec) + 1e-6 * (ru.ru_utime.tv_usec + ru.ru_stime.tv_usec);
}

double sys_realtime(void) { return wall_now() - t_origin; }

void sys_liftrlimit(void)
{ /* reference sys.c:22-30: the address-space soft limit goes up to the hard limit */
	struct rlimit rl;
	if (getrlimit(RLIMIT_AS, &rl) == 0) { rl.rlim_cur = rl.rlim_max; setrlimit(RLIMIT_AS, &rl); }
}

void sys_init(void)
{
	sys_liftrlimit();
	t_origin = 0.;
	t_origin = wall_now();
}

const char *sys_timestamp(void)
{
	static __thread char stamp[256]; /* the tail of one batch may log beside the head of the next (two contexts, two threads) */
	double rt = sys_realtime(), ct = sys_cputime();
	snprintf(stamp, sizeof(stamp) - 1, "%.3f*%.2f", rt, ct / rt);
	return stamp;
}
