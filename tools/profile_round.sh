#!/bin/bash
# One round's profile artefacts (run on the GPU box):  bash tools/profile_round.sh r02
# -> gpurun_out/<tag>_*: rocprofv3 kernel stats of the default bench command (weight gradients on the side stream) and of the
#    single-stream variant, B=1 and bf16 variants, FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs), Winograd tile traces.
tag=${1:-rXX}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash $R/tools/prof_step.sh ${tag}_bench_n1 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null
REFID_OVERLAP_WGRAD=0 REFID_PIPELINE=0 bash $R/tools/prof_step.sh ${tag}_bench_n1_nooverlap --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null
bash $R/tools/prof_step.sh ${tag}_b1 --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null
REFID_OVERLAP_WGRAD=0 REFID_PIPELINE=0 bash $R/tools/prof_step.sh ${tag}_bf16_nooverlap --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null
REFID_OVERLAP_WGRAD=0 REFID_PIPELINE=0 bash $R/tools/prof_step.sh ${tag}_bf16x3_nooverlap --dtype bf16x3 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > /dev/null 2>&1
done
f=$(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
w=$(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python $R/tools/pmc_traffic.py "$f" "$w" $R/gpurun_out/${tag}_pmc_traffic.json
cd $R
python tools/probes/wino_trace.py > gpurun_out/${tag}_wino_tile_trace.txt 2>&1
python tools/probes/wino_trace.py --persistent > gpurun_out/${tag}_wino_persistent_trace.txt 2>&1
python tools/bench_wino2.py > gpurun_out/${tag}_wino_tiles_bench.txt 2>&1
python tools/probes/split_trace.py > gpurun_out/${tag}_split_tile_trace.txt 2>&1
python tools/bench_split.py > gpurun_out/${tag}_split_tiles_bench.txt 2>&1
ZERO=1 python tools/bench_split.py > gpurun_out/${tag}_split_tiles_bench_zero_data.txt 2>&1
tools/probes/bin/mfma_lds_feed > gpurun_out/${tag}_mfma_lds_feed.txt 2>&1
python tools/grad_error_report.py > gpurun_out/${tag}_grad_error_report.txt 2>&1
for d in fp32 bf16x3 bf16; do python bench.py --dtype $d --steps 5 --warmup 2 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_n1_$d.json; done
ls -la gpurun_out | tail -30
