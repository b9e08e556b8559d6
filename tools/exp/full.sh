set -e
mkdir -p gpurun_out/r4
python -m pytest tests -x -q -m gpu > gpurun_out/r4/gpu_tests.txt 2>&1 || { tail -30 gpurun_out/r4/gpu_tests.txt; exit 1; }
tail -3 gpurun_out/r4/gpu_tests.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r4/bench.json 2> gpurun_out/r4/bench.err
python -c "
import json; d=json.load(open('gpurun_out/r4/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['whole_path']['frac']); print(json.dumps(d['roofline'].get('families'), indent=1)); print(d['roofline'].get('worst')); print(d['quality'])"
