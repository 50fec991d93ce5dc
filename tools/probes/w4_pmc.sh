#!/bin/bash
# Hardware counters of ONE weight-gradient kernel at one shape (run on the GPU box):
#   bash tools/probes/w4_pmc.sh <tag> [f4|regs] [shape index into tools/bench_wgrad_wino.py SHAPES]
# Separate --pmc passes (with --kernel-trace only); prints the per-launch mean of every counter for the kernel.
tag=${1:-w4}; which=${2:-f4}; shape=${3:-4}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
SETS="SQ_WAVES,SQ_BUSY_CU_CYCLES,GRBM_GUI_ACTIVE,SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU,SQ_INSTS_MFMA,SQ_INSTS_LDS,SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_LDS_ADDR_CONFLICT,SQ_LDS_DATA_FIFO_FULL SQ_WAIT_INST_LDS,SQ_WAIT_INST_ANY,SQ_WAIT_ANY,SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS,SQ_ACTIVE_INST_VMEM,SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM,SQ_LDS_CMD_FIFO_FULL,SQ_LEVEL_WAVES,SQ_BUSY_CYCLES FETCH_SIZE WRITE_SIZE"
i=0
for c in $SETS; do
  i=$((i+1)); rm -rf /tmp/w4pmc_$i
  ONLY_SHAPE=$shape REPS=3 rocprofv3 --kernel-trace --pmc ${c//,/ } --output-format csv -d /tmp/w4pmc_$i -- python $R/tools/bench_wgrad_wino.py $which > /dev/null 2>&1
done
python - <<PY > $R/gpurun_out/${tag}_pmc.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("/tmp/w4pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wgrad_wino" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]:
            a = agg[(r["Kernel_Name"].split("(")[0][-28:], r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(agg.items()):
    print(f"{k:30s} {c:32s} {v / n:16.1f}  ({n} launches)")
PY
cat $R/gpurun_out/${tag}_pmc.txt
