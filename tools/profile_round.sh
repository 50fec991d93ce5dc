#!/bin/bash
# One round's profile artefacts (run on the GPU box):  bash tools/profile_round.sh r03
# -> gpurun_out/<tag>_*: rocprofv3 kernel stats of the default bench command (weight gradients on the side stream) and of the
#    single-stream variant, B=1 and bf16 variants, FETCH_SIZE / WRITE_SIZE and SQ counter passes (separate --pmc runs, with
#    --kernel-trace only), the per-kernel roofline table (tools/roofline_report.py), bench lines of the three dtypes.
tag=${1:-rXX}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
# PROFILE_LIGHT=1: the step-level artefacts only (kernel stats, counters, roofline table, bench lines) -- not the tile benches,
# ablations and the gradient-error report, whose kernels a late change did not touch
# PROFILE_LIGHT=2: additionally without the B=1 trace, the bf16 traces / counter passes / lines and the inference lines (fp32 train
# step only: a late change that touched nothing else)
LIGHT=${PROFILE_LIGHT:-0}
# the ablation libraries must have been built from the current sources (tools/build_probes.sh, CPU container)
[ "$LIGHT" != 0 ] || python - <<PY || { echo "profile_round: stale or missing probe libraries -- run tools/build_probes.sh first"; exit 1; }
import sys; sys.path.insert(0, "$R/tools/probes")
import wino6_ablate as a, wgrad_wino_ablate as b
a.check_fresh(a.lib_path(0)); b.check_fresh(b.lib_path(0))
PY
bash $R/tools/prof_step.sh ${tag}_bench_n1 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null
REFID_OVERLAP_WGRAD=0 REFID_PIPELINE=0 bash $R/tools/prof_step.sh ${tag}_bench_n1_nooverlap --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null
[ "$LIGHT" = 2 ] || bash $R/tools/prof_step.sh ${tag}_b1 --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null
for b in 1 2 4; do python $R/bench.py --batch $b --steps 8 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $R/gpurun_out/${tag}_b${b}_bench_untraced.json; done
[ "$LIGHT" = 2 ] || REFID_OVERLAP_WGRAD=0 REFID_PIPELINE=0 bash $R/tools/prof_step.sh ${tag}_bf16_nooverlap --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null
cd /tmp
PASSES="FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CU_CYCLES,GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_INSTS_VALU,SQ_INSTS_LDS"
i=0
for c in $PASSES; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  REFID_OVERLAP_WGRAD=0 REFID_PIPELINE=0 rocprofv3 --kernel-trace --pmc ${c//,/ } --output-format csv -d /tmp/pmc_$i -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > /dev/null 2>&1
done
f=$(find /tmp/pmc_1 -name "*counter_collection.csv" | head -1)
w=$(find /tmp/pmc_2 -name "*counter_collection.csv" | head -1)
python $R/tools/pmc_traffic.py "$f" "$w" $R/gpurun_out/${tag}_pmc_traffic.json
# the same two passes for the bf16 mode (roofline.traffic of `bench.py --dtype bf16`)
[ "$LIGHT" = 2 ] || for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcb_$c
  REFID_OVERLAP_WGRAD=0 REFID_PIPELINE=0 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcb_$c -- python $R/bench.py --dtype bf16 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > /dev/null 2>&1
done
[ "$LIGHT" = 2 ] || python $R/tools/pmc_traffic.py "$(find /tmp/pmcb_FETCH_SIZE -name '*counter_collection.csv' | head -1)" "$(find /tmp/pmcb_WRITE_SIZE -name '*counter_collection.csv' | head -1)" $R/gpurun_out/${tag}_pmc_traffic_bf16.json
cd $R
python tools/profile_step.py --json gpurun_out/${tag}_algorithmic.json > gpurun_out/${tag}_profile_step.txt 2>&1
python tools/roofline_report.py --stats gpurun_out/${tag}_bench_n1_nooverlap_kernel_stats.csv --steps 4 \
  --algo gpurun_out/${tag}_algorithmic.json --traffic gpurun_out/${tag}_pmc_traffic.json \
  --sq $(find /tmp/pmc_3 /tmp/pmc_4 -name "*counter_collection.csv") \
  --out gpurun_out/${tag}_roofline_per_kernel.csv --summary gpurun_out/${tag}_pmc_summary.txt > gpurun_out/${tag}_roofline_report.txt 2>&1
if [ "$LIGHT" = 0 ]; then
python tools/bench_wino6.py > gpurun_out/${tag}_wino6_tiles_bench.txt 2>&1
python tools/probes/wino6_ablate.py > gpurun_out/${tag}_wino6_ablation.txt 2>&1
python tools/probes/w24_ablate.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_wgrad_w24_ablation.txt
python tools/bench_wgrad_wino.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_wgrad_tiles_bench.txt
tools/probes/bin/wino6_loop > gpurun_out/${tag}_wino6_loop_probe.txt 2>&1
python tools/grad_error_report.py > gpurun_out/${tag}_grad_error_report.txt 2>&1
fi
DTYPES="fp32 bf16x3 bf16"; [ "$LIGHT" = 2 ] && DTYPES="fp32"
for d in $DTYPES; do python bench.py --dtype $d --steps 5 --warmup 2 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_n1_$d.json; done
# GPU busy fraction without a tracer in the timed run: serial kernel time (traced durations) / untraced single-stream step
REFID_OVERLAP_WGRAD=0 REFID_PIPELINE=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_n1_single_stream_untraced.json
python tools/busy_report.py --stats gpurun_out/${tag}_bench_n1_nooverlap_kernel_stats.csv --stat-steps 4 \
  --single gpurun_out/${tag}_bench_n1_single_stream_untraced.json --default gpurun_out/${tag}_bench_n1_fp32.json > gpurun_out/${tag}_gpu_busy.txt 2>&1
# the driver's command line
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_driver_style.json
# inference lines (BASELINE configs[3] / configs[4])
if [ "$LIGHT" != 2 ]; then
python bench.py --mode infer --config 4 --steps 10 --warmup 2 2>/dev/null | tail -1 > gpurun_out/${tag}_infer_config4.json
python bench.py --mode infer --config 5 --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/${tag}_infer_config5.json
fi
ls -la gpurun_out | tail -30
# an artefact that is a traceback (or empty) is a FAILED collection, never something to commit
bad=$(grep -l -E "Traceback|Error:|error:" gpurun_out/${tag}_*.txt gpurun_out/${tag}_*.json 2>/dev/null; find gpurun_out -name "${tag}_*" -size 0)
if [ -n "$bad" ]; then echo "profile_round: FAILED artefacts:"; echo "$bad"; exit 1; fi
echo "profile_round: all ${tag} artefacts collected"
