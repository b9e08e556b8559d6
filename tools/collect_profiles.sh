#!/bin/bash
# Collects the round's measurement artifacts on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh <out dir under gpurun_out/>      e.g. gpurun_out/prof_r02
# 1. the default bench line (hipGraph replay timed, per-kernel events after it, copy probe, CPU baseline)
# 2. rocprofv3 --kernel-trace --stats of the same command (kernel table)
# 3. PMC passes in their own runs (--pmc only): FETCH_SIZE, WRITE_SIZE, SQ counters
# 4. other configurations
set -o pipefail
O=${1:-gpurun_out/prof_r04}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== bench default"; python -u bench.py > $O/bench_fp16_b32.json 2> $O/bench_fp16_b32.err || exit 1
tail -c 600 $O/bench_fp16_b32.json
# per-kernel passes: one chain (LLIE_ENHANCE_SPLIT=0), so that kernels of the two half-batch graph branches do not stretch each other
export LLIE_ENHANCE_SPLIT=0
echo "== rocprof kernel trace"; rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python bench.py --no-cpu-baseline > $O/bench_fp16_b32_rocprof_run.json 2> $O/kt.err || exit 1
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/bench_fp16_b32_kernel_stats.csv; rm -rf $O/kt
echo "== pmc fetch"; rocprofv3 --pmc FETCH_SIZE -d $O/pf -o pf --output-format csv -- python bench.py --no-cpu-baseline --no-roofline --steps 2 --warmup 1 > $O/pf.log 2>&1 || exit 1
echo "== pmc write"; rocprofv3 --pmc WRITE_SIZE -d $O/pw -o pw --output-format csv -- python bench.py --no-cpu-baseline --no-roofline --steps 2 --warmup 1 > $O/pw.log 2>&1 || exit 1
python tools/pmc_summary.py $O/pf $O/pw $O/pmc_traffic.json "round 4; python bench.py --no-cpu-baseline --no-roofline --steps 2 --warmup 1" > $O/pmc_traffic.txt 2>&1; rm -rf $O/pf $O/pw
echo "== pmc sq"; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/ps -o ps --output-format csv -- python bench.py --no-cpu-baseline --no-roofline --steps 2 --warmup 1 > $O/ps.log 2>&1 || exit 1
python tools/pmc_sq_summary.py $O/ps $O/pmc_sq.json > $O/pmc_sq.txt 2>&1; rm -rf $O/ps
echo "== pmc lds"; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pl -o pl --output-format csv -- python bench.py --no-cpu-baseline --no-roofline --steps 2 --warmup 1 > $O/pl.log 2>&1 || exit 1
python tools/pmc_lds_summary.py $O/pl 30 > $O/pmc_lds.txt 2>&1; rm -rf $O/pl
unset LLIE_ENHANCE_SPLIT
echo "== other configurations"
python -u bench.py --batch 1 --no-cpu-baseline > $O/bench_fp16_b1.json 2>> $O/other.err
python -u bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16_b32.json 2>> $O/other.err
python -u bench.py --dtype fp32 --batch 8 --no-cpu-baseline > $O/bench_fp32_b8.json 2>> $O/other.err
python -u bench.py --variant large --image_size 512 --batch 8 --dtype bf16 --no-cpu-baseline > $O/bench_large512_bf16_b8.json 2>> $O/other.err
python -u bench.py --variant base --lcm_steps 8 --batch 32 --no-cpu-baseline > $O/bench_base256_fp16_b32_n8.json 2>> $O/other.err
python -u bench.py --train --batch 8 --dtype bf16 > $O/train_bf16_b8.json 2>> $O/other.err
python -u bench.py --train --batch 32 --dtype bf16 --no-cpu-baseline > $O/train_bf16_b32.json 2>> $O/other.err
python -u bench.py --train --train-autograd --batch 8 --dtype bf16 --no-cpu-baseline > $O/train_autograd_bf16_b8.json 2>> $O/other.err
python -u tools/gpu_layers.py fp16 32 > $O/layers_fp16_b32.txt 2>&1
python -u tools/gpu_layers.py fp16 1 > $O/layers_fp16_b1.txt 2>&1
ls -la $O
