#!/bin/bash
# several bench.py lines in one GPU call: bash tools/run_bench_set.sh <out prefix> -- "<args of line 1>" "<args of line 2>" ...
R=$GRAFT_REPO_ROOT
P=$1; shift; shift
i=0
for a in "$@"; do
  i=$((i+1))
  timeout -k 10 400 python3 $R/bench.py $a > $R/gpurun_out/${P}_$i.json 2> $R/gpurun_out/${P}_$i.err || { echo "line $i failed"; tail -5 $R/gpurun_out/${P}_$i.err; exit 1; }
  echo "== $a"; cut -c1-100 $R/gpurun_out/${P}_$i.json; python3 -c "
import json,sys
d=json.load(open('$R/gpurun_out/${P}_$i.json'))
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'loss', d.get('loss'))
r=d.get('roofline') or {}
for f in r.get('families', [])[:12]: print('   ', f['kernel'], f['bound'], f['achieved'], f['unit'], 'frac', f['frac'], 'ms/step', f['ms_per_step'], 'n', f['launches'])
"
done
