set -e
mkdir -p gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r4/kt_graph
rocprofv3 --kernel-trace -d gpurun_out/r4/kt_graph -o kt --output-format csv -- python tools/gpu_trace_run.py 32 > gpurun_out/r4/kt_graph.log 2>&1
python tools/overlap_summary.py gpurun_out/r4/kt_graph > gpurun_out/r4/exp20_overlap.txt 2>&1
rm -rf gpurun_out/r4/kt_graph
