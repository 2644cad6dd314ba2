#!/bin/bash
# experiment driver: rocprofv3 kernel stats of the coverage / gather kernels under different switches (one GPU visit)
run() { # name, env...
  name=$1; shift
  rm -rf gpurun_out/exp_$name; mkdir -p gpurun_out/exp_$name
  (cd /tmp && env "$@" timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/exp_$name -o r --output-format csv -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu --no-legs --no-text --prof-steps 0 > /root/repo/gpurun_out/exp_$name/b.json 2> /root/repo/gpurun_out/exp_$name/b.log)
  echo "== $name: step $(python3 -c "import json; print(round(json.load(open('gpurun_out/exp_$name/b.json'))['ms_per_step'],2))")"
  grep -E "k_hit_sub<false|k_hit_gather|k_hit_goff" gpurun_out/exp_$name/r_kernel_stats.csv | awk '{n=split($0,a,","); print substr($1,1,34), a[n-4]}'
  rm -f gpurun_out/exp_$name/r_kernel_trace.csv
}
cd /root/repo
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  run $name X=1 $envs
done
