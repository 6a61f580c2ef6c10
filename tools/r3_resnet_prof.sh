#!/bin/bash
# trainable ResNet-152 + GRU (configs[2] with the encoder trained): kernel-time breakdown of the fp32 step
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_resnet_train
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python3 $R/bench.py --workload resnet_gru --train-encoder --steps 3 --warmup 1 --no-cpu-baseline > $O/line.json 2> $O/err.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*_agent_info.csv" -delete
python3 - <<PY
import csv,glob,json
print(json.load(open('$O/line.json'))['ms_per_step'])
f=glob.glob('$O/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('kernel ms per step', tot/1e6/4)
for r in rows[:28]:
    n=r['Name'].replace('void ','').replace('(anonymous namespace)::','')
    print(f"{n[:80]:80s} {int(r['Calls'])/4:7.1f} {float(r['TotalDurationNs'])/1e6/4:8.3f} ms/step avg {float(r['AverageNs'])/1e3:8.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
