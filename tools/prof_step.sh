#!/bin/bash
# rocprofv3 kernel stats of one bench configuration:  tools/prof_step.sh <tag> <bench args...>
# (run on the GPU box; writes gpurun_out/prof_<tag>/ and copies the kernel-stats CSV to gpurun_out/<tag>_kernel_stats.csv)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench.err
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv
t=$(find $out -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/trace_idle.py "$t" 0.6 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_idle.txt 2>&1
rm -rf $out
cat $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench.json | cut -c1-300; cat $GRAFT_REPO_ROOT/gpurun_out/${tag}_idle.txt
