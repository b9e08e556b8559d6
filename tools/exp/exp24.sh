set -e
mkdir -p gpurun_out/r4
timeout -k 10 300 python -m pytest tests/test_gpu_round4.py -x -q -m gpu -k "half_empty" > gpurun_out/r4/exp24_t0.txt 2>&1 || { tail -30 gpurun_out/r4/exp24_t0.txt; exit 1; }
tail -3 gpurun_out/r4/exp24_t0.txt
timeout -k 10 300 python -u bench.py --variant base --lcm_steps 8 --batch 32 --no-cpu-baseline > gpurun_out/r4/exp24_base.json 2> gpurun_out/r4/exp24_base.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/exp24_base.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
for k,v in list(d['roofline']['step_breakdown'].items())[:12]: print('  ',k,v)
PY
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py -x -q -m gpu > gpurun_out/r4/exp24_tests.txt 2>&1 || { tail -30 gpurun_out/r4/exp24_tests.txt; exit 1; }
tail -3 gpurun_out/r4/exp24_tests.txt
