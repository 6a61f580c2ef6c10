#!/bin/bash
# Round profiles on the GPU box: kernel-trace summaries and the three PMC passes (FETCH_SIZE / WRITE_SIZE / MFMA busy) of
# bench.py, fp32 (headline) and bf16.  Usage (from the repo root on the box): bash tools/profile_round.sh
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_r02
mkdir -p $O
for dt in f32 bf16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$dt -o kt -- python3 $R/bench.py --dtype $dt --steps 7 --warmup 2 --no-cpu-baseline > $O/kt_$dt.json 2> $O/kt_$dt.err
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $pass | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/pmc_${dt}_$tag -o p -- python3 $R/bench.py --dtype $dt --steps 4 --warmup 1 --no-cpu-baseline --no-pipeline > $O/pmc_${dt}_$tag.json 2> $O/pmc_${dt}_$tag.err
    echo "pass $dt $tag done"
  done
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_lct -o kt -- python3 $R/bench.py --workload lct --steps 5 --warmup 2 --no-cpu-baseline > $O/kt_lct.json 2> $O/kt_lct.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_lct_bf16 -o kt -- python3 $R/bench.py --workload lct --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > $O/kt_lct_bf16.json 2> $O/kt_lct_bf16.err
find $O -name "*_agent_info.csv" -delete
find $O -name "*.csv" -size +20M -delete
ls -R $O | head -60
