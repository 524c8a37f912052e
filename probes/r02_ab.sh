#!/bin/bash
# one GPU call: decode-kernel A/B of the variant builds under probes/bin/libv_*.so + e2e tok/s of the gemv variants probes/bin/libg_*.so
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
T=$1
VL=$(ls probes/bin/libv_*.so 2>/dev/null | tr '\n' ' ')
if [ -n "$VL" ]; then
timeout -k 10 300 python3 probes/ab_decode.py --b $VL > $O/ab_$T.txt 2>&1 || { tail -5 $O/ab_$T.txt; exit 1; }
grep -v amdgpu.ids $O/ab_$T.txt
timeout -k 10 300 python3 probes/ab_decode.py --model 70b --ctx 16384 --b $VL > $O/ab_${T}_70b.txt 2>&1 || { tail -5 $O/ab_${T}_70b.txt; exit 1; }
grep -v amdgpu.ids $O/ab_${T}_70b.txt
fi
for kv in q8_0 q4_0; do
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-e2e --n-seq 32 --kv-type $kv > $O/b_${kv}_n32_$T.json 2> $O/b_${kv}_n32_$T.err || { tail -5 $O/b_${kv}_n32_$T.err; exit 1; }
python3 -c "import json; d=json.load(open('$O/b_${kv}_n32_$T.json')); print('$kv n32', d['value'], 'tok/s', d['roofline']['kernel_us']['mean'], 'us frac', d['roofline']['frac'])"
done
echo "e2e (this tree):"; timeout -k 10 300 python3 probes/e2e_profile.py --steps 20 2>&1 | tail -1
for l in probes/bin/libg_*.so; do [ -f "$l" ] || continue; echo "e2e $l:"; PKV_LIB_PATH=$PWD/$l timeout -k 10 300 python3 probes/e2e_profile.py --steps 20 2>&1 | tail -1; done
