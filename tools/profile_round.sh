#!/bin/bash
# Round profiles on the GPU box: kernel-trace summaries and the three PMC passes (FETCH_SIZE / WRITE_SIZE / MFMA busy) of
# bench.py, fp32 (headline) and bf16, kernel-trace summaries of the LCT and PC-DARTS workloads, and -- AFTER the PMC passes, so that
# the line can quote their traffic -- a plain run of each under the kernel-trace profiler (r0N_bench_*_under_rocprof.json).
# Usage (from the repo root on the box): bash tools/profile_round.sh [round tag, default r04] [part: all | base | train]
# Then, in the build container: python tools/collect_profiles.py [round tag]   (stamps and copies the summaries into profiles/)
set -e
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
PART=${2:-all}
if [ "$PART" != "train" ]; then
for dt in f32 bf16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$dt -o kt -- python3 $R/bench.py --dtype $dt --steps 7 --warmup 2 --no-cpu-baseline > $O/kt_$dt.json 2> $O/kt_$dt.err
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $pass | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/pmc_${dt}_$tag -o p -- python3 $R/bench.py --dtype $dt --steps 4 --warmup 1 --no-cpu-baseline --no-pipeline > $O/pmc_${dt}_$tag.json 2> $O/pmc_${dt}_$tag.err
    echo "pass $dt $tag done"
  done
  # merge the passes into the per-kernel JSON bench.py reads its traffic from, then the run whose line quotes it
  python3 $R/tools/pmc_profile.py $O/pmc_conv_$dt.json $O/pmc_${dt}_FETCH_SIZE $O/pmc_${dt}_WRITE_SIZE $O/pmc_${dt}_SQ_VALU_MFMA_BUSY_CYCLES \
    "rocprofv3 --pmc passes of bench.py --dtype $dt --no-pipeline at B=256 (workload w); FETCH_SIZE/WRITE_SIZE in KB, FETCH_SIZE doubled (gfx950 correction)"
  LCTVQA_PMC_FILE=$O/pmc_conv_$dt.json rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt2_$dt -o kt -- python3 $R/bench.py --dtype $dt --steps 7 --warmup 2 > $O/bench_${dt}_under_rocprof.json 2> $O/bench_${dt}_under_rocprof.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_lct -o kt -- python3 $R/bench.py --workload lct --steps 5 --warmup 2 --no-cpu-baseline > $O/kt_lct.json 2> $O/kt_lct.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_lct_bf16 -o kt -- python3 $R/bench.py --workload lct --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > $O/kt_lct_bf16.json 2> $O/kt_lct_bf16.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_darts_enc -o kt -- python3 $R/tools/bench_darts.py --batch 16 --iters 6 > $O/kt_darts_enc.txt 2> $O/kt_darts_enc.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_darts_vqa -o kt -- python3 $R/bench.py --workload darts_vqa --batch 64 --steps 4 --warmup 1 --no-cpu-baseline > $O/kt_darts_vqa.json 2> $O/kt_darts_vqa.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_darts_vqa_bf16 -o kt -- python3 $R/bench.py --workload darts_vqa --batch 64 --dtype bf16 --steps 4 --warmup 1 --no-cpu-baseline > $O/kt_darts_vqa_bf16.json 2> $O/kt_darts_vqa_bf16.err
# the bf16 LCT step on ONE stream: kernel durations without a convolution sharing the chip (the recurrent step's own share)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_lct_bf16_1s -o kt -- python3 $R/bench.py --workload lct --dtype bf16 --no-pipeline --steps 5 --warmup 2 --no-cpu-baseline > $O/kt_lct_bf16_1s.json 2> $O/kt_lct_bf16_1s.err
fi
if [ "$PART" != "base" ]; then
# round 4: the bf16-native trainable trunks -- kernel-trace summaries, and the PMC passes of the ResNet-152 + GRU step (its
# conv_gemm_bf16 / wgrad_bf16 / bn_*_bf16 families are HBM-bound: FETCH_SIZE / WRITE_SIZE per launch against their algorithmic bytes)
RT="--workload resnet_gru --train-encoder --dtype bf16 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_resnet_train_bf16 -o kt -- python3 $R/bench.py $RT --steps 5 --warmup 2 > $O/kt_resnet_train_bf16.json 2> $O/kt_resnet_train_bf16.err
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/pmc_rt_$tag -o p -- python3 $R/bench.py $RT --steps 2 --warmup 1 > $O/pmc_rt_$tag.json 2> $O/pmc_rt_$tag.err
  echo "pass resnet_train $tag done"
done
python3 $R/tools/pmc_profile.py $O/pmc_resnet_train_bf16.json $O/pmc_rt_FETCH_SIZE $O/pmc_rt_WRITE_SIZE $O/pmc_rt_SQ_VALU_MFMA_BUSY_CYCLES \
  "rocprofv3 --pmc passes of bench.py --workload resnet_gru --train-encoder --dtype bf16 at B=256; FETCH_SIZE/WRITE_SIZE in KB, FETCH_SIZE doubled (gfx950 correction); per kernel, per launch"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_lct_train_bf16 -o kt -- python3 $R/bench.py --workload lct --train-encoder --dtype bf16 --steps 4 --warmup 2 --no-cpu-baseline > $O/kt_lct_train_bf16.json 2> $O/kt_lct_train_bf16.err
fi
find $O -name "*_agent_info.csv" -delete
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
find $O -name "*.csv" -size +20M -delete
ls -R $O | head -80
