cd /root/repo
P=/tmp/cfg4.paf
[ -f $P ] || miniasm_amd/bin/pafgen -r 2000000 -n 100000000 -s 2 -o $P 2>/dev/null
for rep in 1 2; do for v in X=1 MA_XFER_THREADS=4 MA_XFER_THREADS=6 MA_XFER_THREADS=12 MA_XFER_THREADS=16; do
  sleep 3
  t0=$(date +%s.%N); env $v MA_PIPE_TIMING=1 miniasm_amd/bin/miniasm $P 2> /tmp/x.log > /tmp/x.gfa; t1=$(date +%s.%N)
  echo "[$v] wall $(python3 -c "print('%.3f' % ($t1 - $t0))") $(grep 'file->HBM' /tmp/x.log | sed 's/.*workers: //') | $(grep 'T::init' /tmp/x.log | sed 's/.*main: //' | cut -c1-60) | $(grep 'T::pipeline' /tmp/x.log)"
done; done
echo "## per-pass walls of the first (only) pass of a process, MA_PIPE_TIMING=2"
sleep 3; MA_PIPE_TIMING=2 miniasm_amd/bin/miniasm $P 2>&1 >/dev/null | grep -E "T::head|T::paf_parse|T::tail|T::pipeline" | head -30
