#!/bin/bash
# the multi-tile decode form after the scalar-table fix: library's choice vs one tile per workgroup (bench.py, attention path only)
out=gpurun_out/r03/decode_tiles2.txt
: > $out
B="python3 bench.py --no-cpu-baseline --no-e2e --no-configs --reps 2 --kernel-reps 10"
run() { local label=$1 nt=$2; shift 2
    local r
    r=$(PKV_DECODE_NTILE=$nt $B "$@" | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f tok/s  %.4f ms/step  kernel %.2f us  frac %.3f  tiles %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_us']['mean'], d['roofline']['frac'], d['config'].get('tiles_per_workgroup')))")
    echo "$label ntile=$nt: $r" | tee -a $out
}
for nt in 0 1 2 4 8; do run "70b ctx16384 f16" $nt --model 70b --ctx 16384 --steps 20 --warmup 5; done
for nt in 0 1; do run "8b 32seq ctx640 f16" $nt --n-seq 32 --ctx 640 --steps 50 --warmup 10; done
for nt in 0 1 16 32; do run "8b 32seq ctx4096 f16" $nt --n-seq 32 --steps 20 --warmup 5; done
for nt in 0 1 4 16; do run "8b 32seq ctx4096 q8_0" $nt --n-seq 32 --kv-type q8_0 --steps 20 --warmup 5; done
for nt in 0 1 4 16; do run "8b 32seq ctx4096 q4_0" $nt --n-seq 32 --kv-type q4_0 --steps 20 --warmup 5; done
