#!/bin/bash
# Inference workloads' counter passes (run on the GPU box):  bash tools/profile_infer.sh r05
# For BASELINE configs[3] / configs[4] (`bench.py --mode infer --config 4|5`): rocprofv3 kernel stats, then FETCH_SIZE and
# WRITE_SIZE in their own --pmc passes (with --kernel-trace only), summarised per kernel and launch by tools/pmc_traffic.py
# into gpurun_out/<tag>_pmc_traffic_infer_config{4,5}.json -- bench.py's roofline.traffic for these workloads is looked up there.
tag=${1:-rXX}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
for c in 4 5; do
  st=$([ $c = 4 ] && echo 4 || echo 1)
  out=/tmp/prof_inf_$c; rm -rf $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $R/bench.py --mode infer --config $c --steps $st --warmup 1 \
     > $R/gpurun_out/${tag}_infer_config${c}_traced.json 2> /dev/null
  cp "$(find $out -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/${tag}_infer_config${c}_kernel_stats.csv
  for p in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmci_${c}_$p
    REFID_PIPELINE=0 rocprofv3 --kernel-trace --pmc $p --output-format csv -d /tmp/pmci_${c}_$p -- \
      python $R/bench.py --mode infer --config $c --steps 1 --warmup 0 --no-roofline > /dev/null 2>&1
  done
  python $R/tools/pmc_traffic.py "$(find /tmp/pmci_${c}_FETCH_SIZE -name '*counter_collection.csv' | head -1)" \
     "$(find /tmp/pmci_${c}_WRITE_SIZE -name '*counter_collection.csv' | head -1)" $R/gpurun_out/${tag}_pmc_traffic_infer_config${c}.json
done
cd $R
