#!/bin/bash
# One GPU-box visit: parity tests, smoke, a bench run and (optionally) rocprofv3 summaries -> gpurun_out/
# usage: tools/gpu_round.sh [tests|bench|prof|all] ...
cd "$(dirname "$0")/.." || exit 1
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
mkdir -p gpurun_out
what="${*:-tests}"
rocminfo 2>/dev/null | grep -m1 -E "gfx9" > gpurun_out/gpu.txt
for w in $what; do
case $w in
tests)
  timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/tests_parity.log 2>&1; echo "rc=$?" >> gpurun_out/tests_parity.log
  grep -vE "^\[M::|^\[pafgen" gpurun_out/tests_parity.log | tail -60 ;;
all_tests)
  timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/tests_all.log 2>&1; echo "rc=$?" >> gpurun_out/tests_all.log
  grep -vE "^\[M::|^\[pafgen" gpurun_out/tests_all.log | tail -40 ;;
graphapi)
  timeout 1500 python -m pytest tests/test_gpu_graph_api.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/tests_graphapi.log 2>&1; echo "rc=$?" >> gpurun_out/tests_graphapi.log
  grep -vE "^\[M::|^\[pafgen" gpurun_out/tests_graphapi.log | tail -40 ;;
benchtiming)
  MA_PIPE_TIMING=1 timeout 900 python bench.py --no-cpu --no-text --steps 5 --warmup 1 > gpurun_out/bench_timing.json 2> gpurun_out/bench_timing.log; echo "rc=$?"
  grep -E "^\[bench\]|T::paf|T::xfer|T::ingest" gpurun_out/bench_timing.log | head -40; python3 -c "
import json; d=json.load(open('gpurun_out/bench_timing.json')); print(d['ms_per_step'], d['e2e'], d['setup']); [print(k['name'], k['launches_per_step'], k['avg_ms']) for k in d['kernels'][:8]]" ;;
e2ecfg4)
  NOREF=1 bash tools/e2e_cfg4.sh | tail -60 ;;
probe)
  # random 32-byte-record gather ceiling + what FETCH_SIZE reports for that access pattern (tools/probes/gather_probe.hip, built here by hipcc)
  tools/probes/gather_probe > gpurun_out/gather_probe.txt 2>&1; cat gpurun_out/gather_probe.txt
  rm -rf gpurun_out/probe_pmc; mkdir -p gpurun_out/probe_pmc
  (cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /root/repo/gpurun_out/probe_pmc -o r --output-format csv -- /root/repo/tools/probes/gather_probe > /dev/null 2>&1); echo "probe pmc rc=$?"
  python tools/pmc_generic.py gpurun_out/probe_pmc --filter k_ --each >> gpurun_out/gather_probe.txt 2>&1; tail -30 gpurun_out/gather_probe.txt ;;
inflight)
  # batches in flight on the one GPU (bench.py --inflight N, each on a context and host thread of its own)
  for cfg in "" "--reads 200000 --lines 10000000 --seed 1"; do for n in 1 2 3; do
    timeout 900 python bench.py --no-cpu --no-legs --no-text --prof-steps 0 --steps 24 --warmup 3 --inflight $n $cfg > gpurun_out/bench_sync.json 2> gpurun_out/bench_sync.log; echo "[$cfg | inflight $n] rc=$?"
    python3 -c "import json; d=json.load(open('gpurun_out/bench_sync.json')); print('   ms_per_step %.3f  value %.3g' % (d['ms_per_step'], d['value']))"
  done; done ;;
tiewalk)
  # the host walk that reproduces the reference's tie order: laps of host/refsort.c on the 50 M noisy input and (TIEWALK_CFG5=1) on BASELINE configs[4]
  miniasm_amd/bin/pafgen -r 1000000 -n 50000000 -s 3 -L uniform -d 0.35 -x 0.03 -o /tmp/tw50.paf 2>/dev/null
  for k in 1 2; do t0=$(date +%s.%N); timeout 600 miniasm_amd/bin/miniasm /tmp/tw50.paf 2> gpurun_out/tiewalk50_plain.log | md5sum; t1=$(date +%s.%N); python3 -c "print(\"plain run: %.3f s wall\" % ($t1 - $t0))"; grep "Real time" gpurun_out/tiewalk50_plain.log; done
  MA_REFSORT_TIMING=1 MA_PIPE_TIMING=2 timeout 600 miniasm_amd/bin/miniasm /tmp/tw50.paf 2> gpurun_out/tiewalk50.log | md5sum
  grep -E "T::refsort|T::ties|T::head\\] sg_gen|Real time|T::xfer.*HBM->host" gpurun_out/tiewalk50.log | head -30
  if [ -n "$TIEWALK_CFG5" ]; then
    miniasm_amd/bin/pafgen -r 5000000 -n 500000000 -s 3 -L uniform -d 0.35 -x 0.03 -o /tmp/tw5.paf 2>/dev/null
    t0=$(date +%s.%N); timeout 900 miniasm_amd/bin/miniasm /tmp/tw5.paf 2> gpurun_out/tiewalk5_plain.log | md5sum; t1=$(date +%s.%N); python3 -c "print(\"plain run: %.3f s wall\" % ($t1 - $t0))"; grep "Real time" gpurun_out/tiewalk5_plain.log
    MA_REFSORT_TIMING=1 MA_PIPE_TIMING=2 timeout 900 miniasm_amd/bin/miniasm /tmp/tw5.paf 2> gpurun_out/tiewalk5.log | md5sum
    echo "## MA_XFER_THREADS=16"; MA_XFER_THREADS=16 MA_PIPE_TIMING=2 timeout 900 miniasm_amd/bin/miniasm /tmp/tw5.paf 2>&1 >/dev/null | grep -E "T::xfer|walk: (packed|order)|Real time" | head -12
    grep -E "T::refsort|T::ties|T::head\\] sg_gen|Real time|T::xfer.*HBM->host" gpurun_out/tiewalk5.log | head -40
    echo "(reference md5 of this input, profiles/r03_e2e_cfg5_500M.txt: fa9c76984d44526d1a9a9e70132d01da)"
  fi ;;
tiewalk2)
  # A/B on one box: the walk's array on huge pages or not (50 M noisy; TIEWALK_CFG5=1: BASELINE configs[4], default and THP)
  miniasm_amd/bin/pafgen -r 1000000 -n 50000000 -s 3 -L uniform -d 0.35 -x 0.03 -o /tmp/tw50.paf 2>/dev/null
  miniasm_amd/bin/pafgen -r 250000 -n 5000000 -s 5 -q 16 -L uniform -d 0.3 -x 0.03 -o /tmp/twrich.paf 2>/dev/null
  for v in "" "MA_HOST_THP=1"; do
    echo "## tie-rich 5 M (bench.py legs.tie_rich's input) [${v:-default}]"
    env $v MA_REFSORT_TIMING=1 MA_PIPE_TIMING=2 timeout 600 miniasm_amd/bin/miniasm /tmp/twrich.paf 2> gpurun_out/tiewalkrich_ab.log | md5sum
    grep -E "top walk|buckets|walk: (packed|order|free|host)|T::head\\] sg_gen|Real time" gpurun_out/tiewalkrich_ab.log | head -6
  done
  for v in "" "MA_HOST_THP=1" ""; do
    echo "## 50 M noisy [${v:-default}]"
    env $v MA_REFSORT_TIMING=1 MA_PIPE_TIMING=2 timeout 600 miniasm_amd/bin/miniasm /tmp/tw50.paf 2> gpurun_out/tiewalk50_ab.log | md5sum
    grep -E "top walk|buckets|walk: (packed|order|free|host)|T::head\] sg_gen|Real time" gpurun_out/tiewalk50_ab.log | head -8
  done
  if [ -n "$TIEWALK_CFG5" ]; then
    miniasm_amd/bin/pafgen -r 5000000 -n 500000000 -s 3 -L uniform -d 0.35 -x 0.03 -o /tmp/tw5.paf 2>/dev/null
    for v in "" "MA_HOST_THP=1"; do
      echo "## BASELINE configs[4] [${v:-default}]"
      env $v MA_REFSORT_TIMING=1 MA_PIPE_TIMING=2 timeout 900 miniasm_amd/bin/miniasm /tmp/tw5.paf 2> gpurun_out/tiewalk5_ab.log | md5sum
      grep -E "top walk|buckets|walk: (packed|order|free|host)|T::head\] sg_gen|Real time" gpurun_out/tiewalk5_ab.log | head -8
    done
  fi ;;
cleanprof)
  # per-kernel times of one CLI run on the 50 M noisy input (where the cleaners and the tie repair are what is left of the device time)
  [ -f /tmp/tw50.paf ] || miniasm_amd/bin/pafgen -r 1000000 -n 50000000 -s 3 -L uniform -d 0.35 -x 0.03 -o /tmp/tw50.paf 2>/dev/null
  rm -rf gpurun_out/cleanprof; mkdir -p gpurun_out/cleanprof
  (cd /tmp && MA_PIPE_TIMING=2 timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/cleanprof -o r --output-format csv -- /root/repo/miniasm_amd/bin/miniasm /tmp/tw50.paf > /dev/null 2> /root/repo/gpurun_out/cleanprof/run.log); echo "rc=$?"
  f=$(find gpurun_out/cleanprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" | cut -c1-150
  grep -E "T::tail|T::clean" gpurun_out/cleanprof/run.log | head -20
  find gpurun_out/cleanprof -name "*trace*.csv" -size +8M -delete ;;
bubwave)
  # the overflow tiers of the bubble probes: a wave per source (default) against a thread per source (MA_BUBBLE_THREAD_TIERS=1, round 2's form); device time of the cleaners
  [ -f /tmp/tw50.paf ] || miniasm_amd/bin/pafgen -r 1000000 -n 50000000 -s 3 -L uniform -d 0.35 -x 0.03 -o /tmp/tw50.paf 2>/dev/null
  [ -f /tmp/twrich.paf ] || miniasm_amd/bin/pafgen -r 250000 -n 5000000 -s 5 -q 16 -L uniform -d 0.3 -x 0.03 -o /tmp/twrich.paf 2>/dev/null
  for f in /tmp/tw50.paf /tmp/twrich.paf; do for v in "" "MA_BUBBLE_THREAD_TIERS=1" "" "MA_BUBBLE_THREAD_TIERS=1"; do
    echo "## $f [${v:-wave per source}]"; env $v MA_PIPE_TIMING=2 timeout 600 miniasm_amd/bin/miniasm $f 2> gpurun_out/bubwave.log | md5sum; grep -E "device cleaners" gpurun_out/bubwave.log; done; done ;;
bubtune)
  # tier 0 of the bubble probes: table size (MA_BUBBLE_CAP0) and launch width (MA_BUBBLE_THREADS0) against the device time of the cleaners
  [ -f /tmp/tw50.paf ] || miniasm_amd/bin/pafgen -r 1000000 -n 50000000 -s 3 -L uniform -d 0.35 -x 0.03 -o /tmp/tw50.paf 2>/dev/null
  [ -f /tmp/twrich.paf ] || miniasm_amd/bin/pafgen -r 250000 -n 5000000 -s 5 -q 16 -L uniform -d 0.3 -x 0.03 -o /tmp/twrich.paf 2>/dev/null
  for v in "" "MA_BUBBLE_THREADS0=131072" "MA_BUBBLE_THREADS0=524288" "MA_BUBBLE_CAP0=16 MA_BUBBLE_CAP1=1024 MA_BUBBLE_THREADS0=524288" "MA_BUBBLE_CAP0=64 MA_BUBBLE_CAP1=1024"; do
    echo "## tie-rich [${v:-default}]"; env $v MA_PIPE_TIMING=2 timeout 600 miniasm_amd/bin/miniasm /tmp/twrich.paf 2> gpurun_out/bubtune.log | md5sum; grep -E "device cleaners" gpurun_out/bubtune.log; done
  for v in "" "MA_BUBBLE_THREADS0=131072" "MA_BUBBLE_THREADS0=524288" "MA_BUBBLE_CAP0=16 MA_BUBBLE_CAP1=1024 MA_BUBBLE_THREADS0=524288" "MA_BUBBLE_CAP0=64 MA_BUBBLE_CAP1=1024" ""; do
    echo "## [${v:-default}]"; env $v MA_PIPE_TIMING=2 timeout 600 miniasm_amd/bin/miniasm /tmp/tw50.paf 2> gpurun_out/bubtune.log | md5sum; grep -E "device cleaners" gpurun_out/bubtune.log; done ;;
poolab)
  # the context's device-memory pool (freed buffers handed out again) against plain hipMalloc / hipFree: CLI runs, wall and the phases that allocate
  miniasm_amd/bin/pafgen -r 2000000 -n 100000000 -s 2 -o /tmp/cfg4.paf 2>/dev/null
  [ -f /tmp/tw50.paf ] || miniasm_amd/bin/pafgen -r 1000000 -n 50000000 -s 3 -L uniform -d 0.35 -x 0.03 -o /tmp/tw50.paf 2>/dev/null
  for f in /tmp/cfg4.paf /tmp/tw50.paf; do for v in 1 0 1 0; do
    t0=$(date +%s.%N); MA_DEV_POOL=$v MA_PIPE_TIMING=2 timeout 600 miniasm_amd/bin/miniasm $f 2> gpurun_out/poolab.log | md5sum | cut -c1-12; t1=$(date +%s.%N)
    python3 -c "print(\"## $f MA_DEV_POOL=$v: %.3f s wall\" % ($t1 - $t0))"; grep -E "hipMalloc of the text|T::head\] (sort|sub #1) |Real time" gpurun_out/poolab.log | tr '\n' ' '; echo; done; done ;;
walkthreads)
  # threads of the tie walk's lower levels (MA_THREADS; default min(cores, 64)) against the `buckets` lap, 50 M noisy and (TIEWALK_CFG5=1) BASELINE configs[4]
  [ -f /tmp/tw50.paf ] || miniasm_amd/bin/pafgen -r 1000000 -n 50000000 -s 3 -L uniform -d 0.35 -x 0.03 -o /tmp/tw50.paf 2>/dev/null
  [ -z "$TIEWALK_CFG5" ] || [ -f /tmp/tw5.paf ] || miniasm_amd/bin/pafgen -r 5000000 -n 500000000 -s 3 -L uniform -d 0.35 -x 0.03 -o /tmp/tw5.paf 2>/dev/null
  for f in /tmp/tw50.paf ${TIEWALK_CFG5:+/tmp/tw5.paf}; do for t in 64 128 96 64; do
    echo "## $f MA_THREADS=$t"; MA_THREADS=$t MA_REFSORT_TIMING=1 MA_PIPE_TIMING=2 timeout 900 miniasm_amd/bin/miniasm $f 2>&1 >/dev/null | grep -E "top walk|buckets|sg_gen|Real time" | head -4 | tr '\n' ' '; echo; done; done ;;
walkprobe)
  # the walk's dependent chain alone on this box's CPU (tools/probes/walk_probe.c): forms x bucket counts x page size
  gcc -O2 -o /tmp/walk_probe tools/probes/walk_probe.c && for nb in 4 16 77; do for form in 0 3 5 1; do for thp in 0 1; do /tmp/walk_probe 100000000 $nb $form $thp; done; done; done 2>&1 | tee gpurun_out/walk_probe.txt ;;
parseprof)
  # per-kernel times of the text-resident leg (device parse + dictionary inside the step)
  rm -rf gpurun_out/parseprof; mkdir -p gpurun_out/parseprof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/parseprof -o r --output-format csv -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu --no-legs --prof-steps 0 > /root/repo/gpurun_out/parseprof/bench.json 2> /root/repo/gpurun_out/parseprof/bench.log); echo "rc=$?"
  f=$(find gpurun_out/parseprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "k_paf|k_dict|k_rec_|k_scan" "$f" | cut -c1-60,200-400 | head -20; python3 -c "import json; d=json.load(open('gpurun_out/parseprof/bench.json')); print('from_text %.2f ms/step' % d['from_text']['ms_per_step'])"
  find gpurun_out/parseprof -name "*trace*.csv" -size +8M -delete ;;
shardsort)
  timeout 900 python tools/shard_sort_probe.py 2>&1 | grep -E "^rank|Error|error" ;;
texttiming)
  # where a text -> GFA step spends its time: the ingest's and the pipeline's own laps (MA_PIPE_TIMING) around the from_text leg
  MA_PIPE_TIMING=1 timeout 900 python bench.py --no-cpu --no-legs --steps 4 --warmup 1 --prof-steps 0 > gpurun_out/bench_texttiming.json 2> gpurun_out/bench_texttiming.log; echo "rc=$?"
  grep -E "T::ingest_gpu|T::paf|T::head|T::tail|T::pipe" gpurun_out/bench_texttiming.log | tail -40
  python3 -c "import json; d=json.load(open('gpurun_out/bench_texttiming.json')); print('   step %.3f ms, from_text %.2f ms/step' % (d['ms_per_step'], d['from_text']['ms_per_step']))" ;;
sqtext)
  # SQ counters (instruction mix, wait cycles) of the ingest kernels: three passes (8 counter slots)
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
    i=$((i+1)); rm -rf gpurun_out/sqt_$i; mkdir -p gpurun_out/sqt_$i
    (cd /tmp && timeout 900 rocprofv3 --pmc $set --kernel-trace -d /root/repo/gpurun_out/sqt_$i -o r --output-format csv -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu --no-text --no-legs --prof-steps 0 > /root/repo/gpurun_out/sqt_$i/bench.json 2> /root/repo/gpurun_out/sqt_$i/bench.log); echo "sq set $i rc=$?"
  done
  python tools/pmc_generic.py gpurun_out/sqt_1 gpurun_out/sqt_2 gpurun_out/sqt_3 --filter "k_paf|k_dict_insert" > gpurun_out/sq_summary_text.txt 2>&1; head -90 gpurun_out/sq_summary_text.txt
  find gpurun_out/sqt_1 gpurun_out/sqt_2 gpurun_out/sqt_3 -name "*.csv" -size +8M -delete ;;
ghtiming)
  # the graph-heavy input (200 M arcs, every read survives, 87 MB of GFA): where a pass spends its time outside the kernels (MA_PIPE_TIMING laps of the tail)
  MA_PIPE_TIMING=1 timeout 900 python bench.py --reads 2000000 --lines 100000000 --seed 4 --model fixed --no-cpu --no-legs --no-text --steps 6 --warmup 2 --prof-steps 0 > gpurun_out/bench_ghtiming.json 2> gpurun_out/bench_ghtiming.log; echo "rc=$?"
  grep -E "T::tail|T::pipe|T::ug" gpurun_out/bench_ghtiming.log | tail -24
  python3 -c "import json; d=json.load(open('gpurun_out/bench_ghtiming.json')); print('   step %.3f ms' % d['ms_per_step'], d.get('latency'))" ;;
benchtext)
  timeout 900 python bench.py --no-cpu --no-legs --steps 6 --warmup 2 --prof-steps 0 > gpurun_out/bench_text.json 2> gpurun_out/bench_text.log; echo "rc=$?"
  python3 -c "import json; d=json.load(open('gpurun_out/bench_text.json')); print('   step %.3f ms, from_text %.2f ms/step, parse+dictionary %.3f s' % (d['ms_per_step'], d['from_text']['ms_per_step'], d['setup']['parse_dictionary_s']))" ;;
projection)
  timeout 1500 python tools/shard_projection.py --ranks 1,2,4,8 --steps 3 > gpurun_out/shard_projection.log 2>&1; echo "rc=$?"; grep -E "^N=" gpurun_out/shard_projection.log ;;
bigthree)
  bash tools/e2e_three.sh ;;
bignoisy)
  MA_PIPE_TIMING=2 bash tools/e2e_big.sh 1000000 50000000 3 "-L uniform -d 0.35 -x 0.03" ref | tail -70 ;;
ties)
  timeout 1500 python -m pytest tests/test_gpu_cli.py -m gpu -q --tb=short -p no:cacheprovider -k "tie" > gpurun_out/tests_ties.log 2>&1; echo "rc=$?" >> gpurun_out/tests_ties.log
  grep -vE "^\[M::|^\[pafgen" gpurun_out/tests_ties.log | tail -60 ;;
cli)
  timeout 1500 python -m pytest tests/test_gpu_cli.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/tests_cli.log 2>&1; echo "rc=$?" >> gpurun_out/tests_cli.log
  grep -vE "^\[M::|^\[pafgen" gpurun_out/tests_cli.log | tail -60 ;;
ingest)
  timeout 1500 python -m pytest tests/test_gpu_ingest.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/tests_ingest.log 2>&1; echo "rc=$?" >> gpurun_out/tests_ingest.log
  grep -vE "^\[M::|^\[pafgen" gpurun_out/tests_ingest.log | tail -60 ;;
sharded)
  timeout 1500 python -m pytest tests/test_gpu_sharded.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/tests_sharded.log 2>&1; echo "rc=$?" >> gpurun_out/tests_sharded.log
  grep -vE "^\[M::|^\[pafgen" gpurun_out/tests_sharded.log | tail -40 ;;
torchrun1)
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --reads 20000 --lines 1000000 --steps 3 --warmup 1 --no-cpu > gpurun_out/bench_torchrun1.json 2> gpurun_out/bench_torchrun1.log; echo "rc=$?"
  tail -3 gpurun_out/bench_torchrun1.log; cut -c1-400 gpurun_out/bench_torchrun1.json ;;
torchrun2debug)
  # 2 processes on the one GPU of this box, gloo collectives: the multi-process control flow of bench.py end to end
  MA_BENCH_ONE_GPU_DEBUG=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --reads 20000 --lines 1000000 --steps 3 --warmup 1 > gpurun_out/bench_torchrun2.json 2> gpurun_out/bench_torchrun2.log; echo "rc=$?"
  tail -5 gpurun_out/bench_torchrun2.log; cut -c1-600 gpurun_out/bench_torchrun2.json ;;
smoke)
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "rc=$?" >> gpurun_out/smoke.log
  grep -vE "^\[M::" gpurun_out/smoke.log | tail -15 ;;
benchsmall)
  timeout 900 python bench.py --reads 20000 --lines 1000000 --steps 3 --warmup 1 > gpurun_out/bench_small.json 2> gpurun_out/bench_small.log; echo "rc=$?" >> gpurun_out/bench_small.log
  tail -5 gpurun_out/bench_small.log; cat gpurun_out/bench_small.json ;;
bench)
  timeout 1700 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.log; echo "rc=$?" >> gpurun_out/bench.log
  grep -E "^\[bench\]|rc=" gpurun_out/bench.log | tail -12; cat gpurun_out/bench.json ;;
tailctx)
  # the device tail on a second context (bench.py --tail-ctx, mahip_tail_handoff): same workload with and without, cfg4 and cfg2, GFA compared
  timeout 900 python -m pytest tests/test_gpu_graph_api.py -m gpu -q --tb=short -p no:cacheprovider -k "second_context or streaming" > gpurun_out/tests_tailctx.log 2>&1; echo "tests rc=$?"
  for v in "" "--no-tail-ctx"; do for cfg in "" "--reads 200000 --lines 10000000 --seed 1"; do
    timeout 900 python bench.py --no-cpu --no-legs --no-text --prof-steps 0 --steps 20 --warmup 3 $cfg $v > gpurun_out/bench_tailctx.json 2> gpurun_out/bench_tailctx.log; echo "[$cfg $v] rc=$?"
    python3 -c "import json; d=json.load(open('gpurun_out/bench_tailctx.json')); print('   ms_per_step %.3f  value %.3g' % (d['ms_per_step'], d['value']))"
  done; done ;;
expbuild)
  # run this HERE before the GPU visit (hipcc cross-compiles): the experiment libraries travel with the snapshot.
  # VARIANTS="name:-DFLAG name2:'-DA -DB'" (always builds `base`)
  eval "bash tools/variants.sh build base:\"\" $VARIANTS" | tail -8 ;;
exprun)
  # parity first (a variant that is not bit-exact is not worth timing), then bench.py per variant
  for v in ${PARITY_NAMES:-$VARIANT_NAMES}; do
    case $v in *+*) continue ;; esac
    MINIASM_AMD_LIB=$PWD/build/variants/$v/libminiasm_amd.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=line -p no:cacheprovider -x > gpurun_out/tests_var_$v.log 2>&1; echo "[$v] parity rc=$?"
  done
  bash tools/variants.sh run base $VARIANT_NAMES ;;
benchcfg2)
  timeout 900 python bench.py --reads 200000 --lines 10000000 --seed 1 --no-legs > gpurun_out/bench_cfg2.json 2> gpurun_out/bench_cfg2.log; echo "rc=$?" >> gpurun_out/bench_cfg2.log
  grep -E "^\[bench\]|rc=" gpurun_out/bench_cfg2.log | tail -8; cat gpurun_out/bench_cfg2.json ;;
benchfixed)
  timeout 1500 python bench.py --model fixed --no-cpu > gpurun_out/bench_fixed.json 2> gpurun_out/bench_fixed.log; echo "rc=$?" >> gpurun_out/bench_fixed.log
  tail -3 gpurun_out/bench_fixed.log; cat gpurun_out/bench_fixed.json ;;
sq)
  # instruction-mix / occupancy counters of the coverage kernels (two passes: the SQ block has 8 counter slots)
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
    i=$((i+1)); rm -rf gpurun_out/sq_$i; mkdir -p gpurun_out/sq_$i
    (cd /tmp && timeout 900 rocprofv3 --pmc $set --kernel-trace -d /root/repo/gpurun_out/sq_$i -o r --output-format csv -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu --no-text --no-legs --prof-steps 0 > /root/repo/gpurun_out/sq_$i/bench.json 2> /root/repo/gpurun_out/sq_$i/bench.log); echo "sq set $i rc=$?"
  done
  python tools/pmc_generic.py gpurun_out/sq_1 gpurun_out/sq_2 gpurun_out/sq_3 --filter k_hit_sub > gpurun_out/sq_summary.txt 2>&1; cat gpurun_out/sq_summary.txt | head -80
  find gpurun_out/sq_1 gpurun_out/sq_2 gpurun_out/sq_3 -name "*.csv" -size +8M -delete ;;
benchexact)
  MA_EXACT_TIES=1 timeout 1500 python bench.py --no-cpu --steps 3 --warmup 1 --prof-steps 0 > gpurun_out/bench_exact.json 2> gpurun_out/bench_exact.log; echo "rc=$?" >> gpurun_out/bench_exact.log
  tail -3 gpurun_out/bench_exact.log; cat gpurun_out/bench_exact.json ;;
prof)
  rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
  (cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof -o r --output-format csv -- python /root/repo/bench.py --steps 5 --warmup 1 --no-cpu --no-legs --no-text --prof-steps 0 > /root/repo/gpurun_out/prof/bench_under_prof.json 2> /root/repo/gpurun_out/prof/bench_under_prof.log); echo "rc=$?"
  find gpurun_out/prof -name "*stats*" | head; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f" ;;
pmc)
  # HBM traffic counters, one counter per run (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2: they do not fit one pass)
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_$ctr; mkdir -p gpurun_out/pmc_$ctr
    (cd /tmp && timeout 1500 rocprofv3 --pmc $ctr --kernel-trace -d /root/repo/gpurun_out/pmc_$ctr -o r --output-format csv -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu --no-legs --no-text --prof-steps 0 > /root/repo/gpurun_out/pmc_$ctr/bench.json 2> /root/repo/gpurun_out/pmc_$ctr/bench.log); echo "pmc $ctr rc=$?"
    find gpurun_out/pmc_$ctr -name "*.csv" | head -5
  done
  python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > gpurun_out/pmc_summary.json 2> gpurun_out/pmc_summary.log; tail -3 gpurun_out/pmc_summary.log; head -c 3000 gpurun_out/pmc_summary.json ;;
runsab)
  # the hit sort on RUNS of records (default) against records (MA_SORT_RUNS=0): step time and the sort group's kernels at BASELINE configs[3]; parity first
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ingest.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | grep -vE "^\[M::|^\[pafgen" | tail -3
  for v in "" "MA_SORT_RUNS=0" "" "MA_SORT_RUNS=0"; do
    env $v timeout 900 python bench.py --steps 10 --warmup 2 --prof-steps 3 --no-cpu --no-legs --no-text > gpurun_out/bench_runsab.json 2> gpurun_out/bench_runsab.log; echo "[${v:-runs}] rc=$?"
    python3 -c "
import json; d=json.load(open('gpurun_out/bench_runsab.json')); r=d['roofline']; print('   ms_per_step %.3f  sort group %.3f ms frac %.3f' % (d['ms_per_step'], r['avg_launch_ms'], r['frac'])); [print('   %-26s x%-4g %.3f ms' % (k['name'], k['launches_per_step'], k['avg_ms'])) for k in d['kernels'] if k['name'] in ('k_hit_keys','k_radix_hist','k_radix_scatter','k_radix_colscan','k_runs_expand','k_group_close','k_hit_sub<gather>','k_hit_sub<cut+flt>','k_hit_cut_contained')]"
  done ;;
transab)
  # asg_arc_del_trans, first tier: round 5's pipelined kernel (default) against round 4's (MA_TRANS_OLD=1) on the graph-heavy input (100 M overlaps, 200 M arcs): HIP-event kernel times of bench.py
  GH="--reads 2000000 --lines 100000000 --seed 4 --model fixed"
  for v in "" "MA_TRANS_OLD=1" ""; do
    env $v timeout 900 python bench.py $GH --steps 4 --warmup 1 --prof-steps 3 --no-cpu --no-legs --no-text > gpurun_out/bench_transab.json 2> gpurun_out/bench_transab.log; echo "[${v:-pipelined}] rc=$?"
    python3 -c "
import json; d=json.load(open('gpurun_out/bench_transab.json')); print('   ms_per_step %.3f' % d['ms_per_step']); [print('   %-26s x%-4g %.3f ms' % (k['name'], k['launches_per_step'], k['avg_ms'])) for k in d['kernels'] if k['name'] in ('k_asg_trans','k_arc_group_sort','k_arc_rm','k_arc_index','k_asg_symm')]"
  done ;;
initlaps)
  # where a command-line run spends its start and its end: [T::init] laps, wall with the fast exit (default) and with the full teardown (MA_CLEAN_EXIT=1); configs[2] stand-in (40 M) and configs[3] (100 M)
  miniasm_amd/bin/pafgen -r 1200000 -n 40000000 -s 4 -o /tmp/il40.paf 2>/dev/null
  miniasm_amd/bin/pafgen -r 2000000 -n 100000000 -s 2 -o /tmp/il100.paf 2>/dev/null
  for f in /tmp/il40.paf /tmp/il100.paf; do
    cat $f > /dev/null
    for v in "" "MA_CLEAN_EXIT=1" "" "MA_CLEAN_EXIT=1"; do
      t0=$(date +%s.%N); env $v timeout 600 miniasm_amd/bin/miniasm $f 2> gpurun_out/initlaps.log | md5sum | cut -c1-12; t1=$(date +%s.%N)
      python3 -c "print(\"## $f [${v:-fast exit}]: %.3f s wall\" % ($t1 - $t0))"; grep "Real time" gpurun_out/initlaps.log
    done
    MA_PIPE_TIMING=2 timeout 600 miniasm_amd/bin/miniasm $f 2>&1 >/dev/null | grep -E "T::init|T::paf|T::xfer|T::ingest|T::head|T::pipeline|T::tail|ma_hit_read|Real time" | head -30
    LD_DEBUG=statistics miniasm_amd/bin/miniasm -V 2>&1 | grep -E "total startup|relocation|load" | head -4
  done ;;
sqgh)
  # SQ counters (instruction mix, wait cycles) of the graph phase's kernels on the graph-heavy input (200 M arcs) and of the sort group at configs[3]: three passes each (8 counter slots)
  GH="--reads 2000000 --lines 100000000 --seed 4 --model fixed"
  for wl in gh cfg4; do
    case $wl in cfg4) A=""; F="k_hit_keys_runs|k_runs_expand|k_runs_count|k_radix_scatter|k_hit_sub";; gh) A="$GH"; F="k_asg_trans|k_arc_group_sort|k_arc_rm|k_sg_emit|k_sg_arcs";; esac
    i=0
    for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
      i=$((i+1)); rm -rf gpurun_out/sq_${wl}_$i; mkdir -p gpurun_out/sq_${wl}_$i
      (cd /tmp && timeout 900 rocprofv3 --pmc $set --kernel-trace -d /root/repo/gpurun_out/sq_${wl}_$i -o r --output-format csv -- python /root/repo/bench.py $A --steps 1 --warmup 1 --no-cpu --no-text --no-legs --prof-steps 0 > /root/repo/gpurun_out/sq_${wl}_$i/bench.json 2> /root/repo/gpurun_out/sq_${wl}_$i/bench.log); echo "sq $wl set $i rc=$?"
    done
    python tools/pmc_generic.py gpurun_out/sq_${wl}_1 gpurun_out/sq_${wl}_2 gpurun_out/sq_${wl}_3 --filter "$F" > gpurun_out/sq_summary_$wl.txt 2>&1; head -70 gpurun_out/sq_summary_$wl.txt
    find gpurun_out/sq_${wl}_1 gpurun_out/sq_${wl}_2 gpurun_out/sq_${wl}_3 -name "*.csv" -size +8M -delete
  done ;;
evidence)
  # the round's evidence at the current commit (what tools/gpu_r4z.sh did in round 4): rocprofv3 kernel stats and PMC traffic of the bench command at BASELINE configs[3] and on
  # the graph-heavy input -> gpurun_out/ev/{rocprofv3_kernel_stats,pmc_traffic}_{cfg4,gh}.*; copy them to profiles/rNN_* afterwards
  O=gpurun_out/ev; mkdir -p $O
  GH="--reads 2000000 --lines 100000000 --seed 4 --model fixed"
  for wl in ${EVIDENCE_WORKLOADS:-cfg4 gh}; do
    case $wl in cfg4) A="";; gh) A="$GH";; esac
    rm -rf $O/prof_$wl; mkdir -p $O/prof_$wl
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof_$wl -o r --output-format csv -- python /root/repo/bench.py $A --steps 5 --warmup 1 --no-cpu --no-legs --no-text --prof-steps 0 > /root/repo/$O/prof_$wl/bench.json 2> /root/repo/$O/prof_$wl/bench.log); echo "rocprof $wl rc=$?"
    f=$(find $O/prof_$wl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/rocprofv3_kernel_stats_$wl.csv && head -12 $f | cut -c1-140
    find $O/prof_$wl -name "*.csv" ! -name "*stats*" -delete 2>/dev/null
    for ctr in FETCH_SIZE WRITE_SIZE; do
      rm -rf $O/pmc_${wl}_$ctr; mkdir -p $O/pmc_${wl}_$ctr
      (cd /tmp && timeout 900 rocprofv3 --pmc $ctr --kernel-trace -d /root/repo/$O/pmc_${wl}_$ctr -o r --output-format csv -- python /root/repo/bench.py $A --steps 2 --warmup 1 --no-cpu --no-legs --no-text --prof-steps 0 > /root/repo/$O/pmc_${wl}_$ctr/bench.json 2> /root/repo/$O/pmc_${wl}_$ctr/bench.log); echo "pmc $wl $ctr rc=$?"
    done
    python tools/pmc_summary.py $O/pmc_${wl}_FETCH_SIZE $O/pmc_${wl}_WRITE_SIZE > $O/pmc_traffic_$wl.json 2> $O/pmc_$wl.log; tail -1 $O/pmc_$wl.log
    find $O/pmc_${wl}_FETCH_SIZE $O/pmc_${wl}_WRITE_SIZE -name "*.csv" -size +1M -delete 2>/dev/null
  done ;;
benchsum)
  # the default bench line (as the driver runs it) + a one-screen summary
  timeout 1700 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.log; echo "bench rc=$?"
  python3 - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/bench_default.json"))
    print("ms_per_step %.3f  value %.4g  gfa_identical %s  latency %s  e2e %s  from_text %s" % (d["ms_per_step"], d["value"], d["gfa_identical"], d.get("latency") and d["latency"]["ms"], d.get("e2e") and round(d["e2e"]["wall_s"], 3), d.get("from_text") and round(d["from_text"]["ms_per_step"], 2)))
    r = d["roofline"]; print("roofline (sort group): %.3f ms frac %.3f | dominant %s %.3f ms frac %.3f | hit_chain %.3f" % (r["avg_launch_ms"], r["frac"], r["dominant_kernel"]["kernel"], r["dominant_kernel"]["avg_launch_ms"], r["dominant_kernel"]["frac"], r["hit_chain"]["frac"]))
    rg = r.get("reduce_group"); print("reduce_group:", rg and (rg["ms_per_step"], rg["frac"], rg["slowest_by_8d"]), rg and [(k["name"], k["avg_ms"]) for k in rg["kernels"]])
    for n, l in d["legs"].items(): print("leg %-12s %s  identical %s / %s" % (n, ("%.3f ms/step" % l["ms_per_step"]) if "ms_per_step" in l else ("%.3f s wall" % l["wall_s"]), l.get("gfa_identical"), l.get("gfa_md5_matches_reference", l.get("gfa_md5_matches_recorded_reference"))))
    print("cpu_baseline", d["cpu_baseline"] and d["cpu_baseline"]["value"])
    for k in d["kernels"][:12]: print("  %-28s x%-4g %.3f ms" % (k["name"], k["launches_per_step"], k["avg_ms"]))
except Exception as e:
    print("bench summary failed:", e)
PY
  ;;
esac
done
