#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 400 python3 probes/gemv_shapes.py --libs probes/bin/libg_abl1.so probes/bin/libg_abl2.so > $O/gemv_shapes_abl.txt 2>&1 || { tail -5 $O/gemv_shapes_abl.txt; exit 1; }
grep -v amdgpu.ids $O/gemv_shapes_abl.txt
echo "== tpw32 decode tests"
PKV_LIB_PATH=$PWD/probes/bin/libq_tpw32.so timeout -k 10 600 python -m pytest tests/test_gpu_decode.py tests/test_gpu_decode_step.py -m gpu -x -q 2>&1 | tail -3 || exit 1
for kv in q8_0 q4_0; do
PKV_LIB_PATH=$PWD/probes/bin/libq_tpw32.so timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-e2e --n-seq 32 --kv-type $kv > $O/b_${kv}_n32_tpw32.json 2> $O/b_${kv}_n32_tpw32.err || { tail -5 $O/b_${kv}_n32_tpw32.err; exit 1; }
python3 -c "import json; d=json.load(open('$O/b_${kv}_n32_tpw32.json')); print('$kv n32 tpw32', d['value'], 'tok/s', d['roofline']['kernel_us']['mean'], 'us frac', d['roofline']['frac'])"
PKV_LIB_PATH=$PWD/probes/bin/libq_tpw32.so timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-e2e --kv-type $kv > $O/b_${kv}_n1_tpw32.json 2> $O/b_${kv}_n1_tpw32.err || { tail -5 $O/b_${kv}_n1_tpw32.err; exit 1; }
python3 -c "import json; d=json.load(open('$O/b_${kv}_n1_tpw32.json')); print('$kv n1 tpw32', d['value'], 'tok/s', d['roofline']['kernel_us']['mean'], 'us frac', d['roofline']['frac'])"
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-e2e --kv-type $kv > $O/b_${kv}_n1.json 2> $O/b_${kv}_n1.err || { tail -5 $O/b_${kv}_n1.err; exit 1; }
python3 -c "import json; d=json.load(open('$O/b_${kv}_n1.json')); print('$kv n1 tpw16', d['value'], 'tok/s', d['roofline']['kernel_us']['mean'], 'us frac', d['roofline']['frac'])"
done
echo "== fused merge"
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-e2e --fused-merge > $O/b_fused.json 2> $O/b_fused.err || { tail -5 $O/b_fused.err; exit 1; }
python3 -c "import json; d=json.load(open('$O/b_fused.json')); print('fused merge', d['value'], 'tok/s', d['ms_per_step'], 'ms/step', d['roofline']['kernel_us'])"
