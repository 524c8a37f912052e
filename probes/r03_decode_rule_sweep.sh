#!/bin/bash
# the tile rule against one tile per workgroup across launch sizes (bench.py, attention path only, ms per step of all 32 layers)
out=gpurun_out/r03/decode_rule_sweep.txt
: > $out
B="python3 bench.py --no-cpu-baseline --no-e2e --no-configs --reps 1 --kernel-reps 4 --steps 30 --warmup 5"
for ctx in 640 4096; do for ns in 1 2 3 4 6 8 12 16 24 32; do
  a=$(PKV_DECODE_NTILE=0 $B --n-seq $ns --ctx $ctx 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f %s' % (d['ms_per_step'], d['config'].get('tiles_per_workgroup')))")
  b=$(PKV_DECODE_NTILE=1 $B --n-seq $ns --ctx $ctx 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f' % d['ms_per_step'])")
  echo "ctx $ctx n_seq $ns: rule $a | one tile $b" | tee -a $out
done; done
