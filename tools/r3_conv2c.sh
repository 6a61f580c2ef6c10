#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_edge -o kt -- python3 $R/tools/ab_edge.py $R/lct_vqa_amd/_C/liblctvqa_hip.so --rounds 3 > $R/gpurun_out/prof_edge.txt 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/prof_edge/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
with open('$R/gpurun_out/r3_conv2_kern.txt','w') as o:
  for r in rows[:70]:
    o.write(f"{r['Name'][:110]:110s} {r['Calls']:>5s} {float(r['TotalDurationNs'])/1e6:8.2f} ms {float(r['AverageNs'])/1e3:8.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%\n")
PY
find $R/gpurun_out/prof_edge -name "*kernel_trace.csv" -delete
