#!/bin/bash
# kernel-trace summaries of the PC-DARTS paths (encoder fwd+bwd at B=16, darts_vqa weight step + Architect.step at B=64) and the
# F(4x4,3x3) workgroup-order A/B; outputs under gpurun_out/prof_r03
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_r03
mkdir -p $O
python3 $R/tools/ab_wino.py $R/build/ab/w4_base.so $R/build/ab/w4_band.so --f43 --rounds 4 > $O/ab_w4_band.txt 2>&1
tail -12 $O/ab_w4_band.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_darts_enc -o kt -- python3 $R/tools/bench_darts.py --batch 16 --iters 6 > $O/kt_darts_enc.txt 2> $O/kt_darts_enc.err
tail -2 $O/kt_darts_enc.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_darts_vqa -o kt -- python3 $R/bench.py --workload darts_vqa --batch 64 --steps 4 --warmup 1 --no-cpu-baseline > $O/kt_darts_vqa.json 2> $O/kt_darts_vqa.err
tail -c 600 $O/kt_darts_vqa.json
find $O -name "*_agent_info.csv" -delete
find $O -name "*kernel_trace.csv" -size +20M -delete
ls -R $O | head -40
