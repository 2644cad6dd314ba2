import sys, os, time, subprocess
sys.path.insert(0, '/root/repo')
import miniasm_amd as ma
paf = '/tmp/ib.paf'
if not os.path.exists(paf):
    subprocess.run([ma.PAFGEN_PATH, '-r', '200000', '-n', '10000000', '-s', '1', '-o', paf], stderr=subprocess.DEVNULL, check=True)
for th in sys.argv[1:]:
    os.environ['MA_THREADS'] = th
    os.environ['MA_PIPE_TIMING'] = '1'
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); ing = ma.Ingest(paf, ma.default_opt()); dt = time.perf_counter() - t0; ing.close()
        best = min(best, dt)
    print("threads %s: best %.3f s = %.1f M lines/s" % (th, best, 10 / best), flush=True)
