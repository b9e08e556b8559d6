set -e
mkdir -p gpurun_out/r4
show() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], d.get('final_loss'))" $1; }
for i in 1 2; do
timeout -k 10 300 python -u bench.py --train --batch 8 --dtype bf16 --no-cpu-baseline > gpurun_out/r4/exp27_fused_$i.json 2>> gpurun_out/r4/exp27.err; show gpurun_out/r4/exp27_fused_$i.json
timeout -k 10 300 python -u bench.py --train --train-autograd --batch 8 --dtype bf16 --no-cpu-baseline > gpurun_out/r4/exp27_autograd_$i.json 2>> gpurun_out/r4/exp27.err; show gpurun_out/r4/exp27_autograd_$i.json
done
timeout -k 10 300 python -u bench.py --train --batch 32 --dtype bf16 --no-cpu-baseline > gpurun_out/r4/exp27_fused_b32.json 2>> gpurun_out/r4/exp27.err; show gpurun_out/r4/exp27_fused_b32.json
