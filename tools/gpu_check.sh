#!/bin/bash
# One call on the GPU box: the driver's default bench line (with the other configurations' shares) and the whole -m gpu suite.
#   gpurun -- bash tools/gpu_check.sh      ->  gpurun_out/bench_default.json, gpurun_out/gpu_full.log
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python3 bench.py --steps 20 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python3 - <<'PY'
import json
b = json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
print(b['value'], b['ms_per_step'], b['roofline']['frac'], b['roofline']['kernel'])
for k, v in b['extra']['other_configs'].items():
    print(k, v.get('kernel'), v.get('kernel_ms'), v.get('frac'), v.get('error'))
print(b['cpu_baseline']['value'], b['cpu_baseline']['cores'])
PY
python3 -m pytest tests -q -m gpu > gpurun_out/gpu_full.log 2>&1
tail -8 gpurun_out/gpu_full.log
