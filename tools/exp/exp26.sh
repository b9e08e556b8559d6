set -e
mkdir -p gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/r4/ktt -o kt --output-format csv -- python bench.py --train --batch 8 --dtype bf16 --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r4/exp26_train.json 2> gpurun_out/r4/exp26.err
python tools/train_trace_summary.py gpurun_out/r4/ktt > gpurun_out/r4/exp26_summary.txt 2>&1
cat gpurun_out/r4/exp26_summary.txt
rm -rf gpurun_out/r4/ktt
