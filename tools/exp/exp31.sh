mkdir -p gpurun_out/r4
for i in 1 2 3; do timeout -k 10 300 python -u bench.py --no-cpu-baseline 2>/dev/null > gpurun_out/r4/exp31_$i.json; python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['peak_measured'])" gpurun_out/r4/exp31_$i.json; done
