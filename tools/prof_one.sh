#!/bin/bash
# One workload under rocprofv3 --kernel-trace --stats on the GPU box: bash tools/prof_one.sh <tag> <bench.py arguments...>
# -> gpurun_out/prof_<tag>/{kt_kernel_stats.csv, bench.json, bench.err}
set -e
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python3 $R/bench.py "$@" > $O/bench.json 2> $O/bench.err
find $O -name "*_agent_info.csv" -delete
find $O -name "*kernel_trace.csv" -delete
f=$(find $O -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && mv $f $O/kt_kernel_stats.csv
find $O -mindepth 1 -type d -empty -delete
cat $O/bench.json | cut -c1-600
head -40 $O/kt_kernel_stats.csv | cut -c1-200
