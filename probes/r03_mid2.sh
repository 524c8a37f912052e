#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_decode_step.py tests/test_gpu_prefill.py tests/test_gpu_decode.py tests/test_gpu_runner.py tests/test_gpu_configs.py -q -m gpu --tb=line > $O/t5.log 2>&1; tail -6 $O/t5.log
grep -q "Memory access fault" $O/t5.log && exit 1
timeout -k 10 900 python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err || { tail -5 $O/bench_n1.err; exit 1; }
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r03/bench_n1.json'))
print('headline', d['value'], d['ms_per_step'], d['roofline']['kernel_us']['mean'], d['roofline']['frac'], d['config'].get('tiles_per_workgroup'))
print('e2e', d['e2e']['value'], d['e2e']['roofline']['frac'])
for k, v in d['configs'].items():
    print(k, v.get('value'), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), (v.get('roofline') or {}).get('kernel_us', {}).get('mean'), v.get('tiles_per_workgroup'), v.get('S_PP'), v.get('S_PP_f16_io'), v.get('S_TG'), v.get('error'))
PY
timeout -k 10 400 python3 probes/gemv_shapes.py --packed --rounds 5 > $O/gemv_shapes_cold.txt 2>&1; grep -v amdgpu.ids $O/gemv_shapes_cold.txt
timeout -k 10 400 python3 probes/gemv_shapes.py --packed --rounds 5 --hot 2 > $O/gemv_shapes_hot.txt 2>&1; grep -v amdgpu.ids $O/gemv_shapes_hot.txt
