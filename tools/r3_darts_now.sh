#!/bin/bash
# darts_vqa weight step at B=64 (bench line) and its kernel-time breakdown by family
R=$GRAFT_REPO_ROOT
python3 $R/bench.py --workload darts_vqa --batch 64 --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r3_darts_now.json 2> $R/gpurun_out/r3_darts_now.err || { tail -5 $R/gpurun_out/r3_darts_now.err; exit 1; }
python3 -c "
import json; d=json.load(open('$R/gpurun_out/r3_darts_now.json')); print('darts_vqa B=64:', d['ms_per_step'], 'ms/step', d['value'], d.get('roofline',{}).get('frac'))"
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_darts_now
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python3 $R/bench.py --workload darts_vqa --batch 64 --steps 4 --warmup 1 --no-cpu-baseline > $O/line.json 2> $O/err.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*_agent_info.csv" -delete
python3 - <<PY
import csv,re,collections,glob
f=glob.glob('$O/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
cat=collections.Counter(); calls=collections.Counter()
for r in rows:
    n=r['Name']
    m=re.search(r'(edge_\w+?_kernel|gemm_\w+|__amd_rocclr_\w+|\w+_kernel)',n)
    k=m.group(1) if m else n[:40]
    if 'at::native' in n:
        mm=re.search(r'(\w*Functor\w*)',n); k='aten:'+(mm.group(1) if mm else n[20:60])
    cat[k]+=float(r['TotalDurationNs']); calls[k]+=int(r['Calls'])
print('kernel time total ms', tot/1e6, '(5 steps incl warmup)')
for k,v in cat.most_common(24): print(f"  {k:50s} {calls[k]:6d} {v/1e6:8.2f} {100*v/tot:5.1f}%")
PY
