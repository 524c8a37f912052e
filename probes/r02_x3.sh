#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests_x3.log 2>&1; rc=$?
tail -4 $O/tests_x3.log
[ $rc -ne 0 ] && { grep -E "^E  " $O/tests_x3.log | head -12; exit $rc; }
timeout -k 10 400 python3 probes/gemv_shapes.py > $O/gemv_shapes_x3.txt 2>&1 || { tail -5 $O/gemv_shapes_x3.txt; exit 1; }
grep -v amdgpu.ids $O/gemv_shapes_x3.txt
echo "e2e:"; timeout -k 10 300 python3 probes/e2e_profile.py --steps 20 2>&1 | tail -1
for kv in q8_0 q4_0; do
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-e2e --n-seq 32 --kv-type $kv > $O/b_${kv}_n32_x3.json 2> $O/b_${kv}_n32_x3.err || { tail -5 $O/b_${kv}_n32_x3.err; exit 1; }
python3 -c "import json; d=json.load(open('$O/b_${kv}_n32_x3.json')); print('$kv n32', d['value'], 'tok/s', d['roofline']['kernel_us']['mean'], 'us frac', d['roofline']['frac'])"
done
VL=$(ls probes/bin/libv_*.so 2>/dev/null | tr '\n' ' ')
timeout -k 10 300 python3 probes/ab_decode.py --model 70b --ctx 16384 --b $VL > $O/ab_x3_70b.txt 2>&1 || { tail -5 $O/ab_x3_70b.txt; exit 1; }
grep -v amdgpu.ids $O/ab_x3_70b.txt
timeout -k 10 300 python3 probes/ab_decode.py --b $VL > $O/ab_x3.txt 2>&1 || { tail -5 $O/ab_x3.txt; exit 1; }
grep -v amdgpu.ids $O/ab_x3.txt
