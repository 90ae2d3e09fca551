#!/bin/bash
mkdir -p gpurun_out
for w in decode infer resnet34 preprocess map; do
  timeout -k 10 300 python bench.py --workload $w --steps 20 --warmup 3 > gpurun_out/ak_$w.json 2> gpurun_out/ak_$w.err
  echo "$w rc=$? $(python - $w <<'PY'
import json, sys
try:
    d = json.loads(open(f'gpurun_out/ak_{sys.argv[1]}.json').read().strip().splitlines()[-1])
    print(d['metric'][:40], d['value'], d['unit'], 'roofline', d['roofline']['frac'], 'cpu', d['cpu_baseline']['value'])
except Exception as e:
    print('parse failed', e)
PY
)"
done
timeout -k 10 600 python bench.py --variant 512 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ak_512.json 2> gpurun_out/ak_512.err
echo "ssd512 rc=$? $(python -c "import json; d=json.loads(open('gpurun_out/ak_512.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
