#!/bin/bash
# kernel timeline of the graph-heavy input with both contexts at work (rocprofv3 --kernel-trace): where the head's stream waits
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
GH="--reads 2000000 --lines 100000000 --seed 4 --model fixed --no-cpu --no-legs --no-text --steps 6 --warmup 2 --prof-steps 0"
rm -rf gpurun_out/ghtrace; mkdir -p gpurun_out/ghtrace
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --memory-copy-trace -d /root/repo/gpurun_out/ghtrace -o r --output-format csv -- python /root/repo/bench.py $GH > /root/repo/gpurun_out/ghtrace/bench.json 2> /root/repo/gpurun_out/ghtrace/bench.log); echo "rc=$?"
python3 - <<'PY'
import csv, json, collections
d = json.load(open('gpurun_out/ghtrace/bench.json')); print('step %.3f ms' % d['ms_per_step'], d['phases'] and {k: d['phases'][k] for k in ('head_wall_ms', 'tail_wall_ms')})
rows = list(csv.DictReader(open('gpurun_out/ghtrace/r_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last full step: from the last k_hit_keys_runs back one
starts = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('void k_hit_keys_runs')]
i0, i1 = starts[-2], starts[-1]
t0 = int(rows[i0]['Start_Timestamp'])
print('one step = %.3f ms between two k_hit_keys_runs' % ((int(rows[i1]['Start_Timestamp']) - t0) / 1e6))
byq = collections.defaultdict(list)
for r in rows[i0:i1]:
    byq[r['Queue_Id']].append(r)
for q, rs in byq.items():
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs) / 1e6
    print('queue %s: %d kernels, busy %.3f ms, span %.3f .. %.3f ms' % (q, len(rs), busy, (int(rs[0]['Start_Timestamp']) - t0) / 1e6, (int(rs[-1]['End_Timestamp']) - t0) / 1e6))
    prev = None
    for r in rs:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = (s - prev) / 1e3 if prev else 0
        if gap > 300 or (e - s) > 1000000:
            print('   %9.3f ms  gap %8.1f us  dur %8.1f us  %s' % ((s - t0) / 1e6, gap, (e - s) / 1e3, r['Kernel_Name'][:44]))
        prev = e
PY
find gpurun_out/ghtrace -name "*.csv" -size +8M -delete
