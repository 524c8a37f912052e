#!/bin/bash
# A/B: the multi-tile f16 decode form at gqa 8 compiled for 4 waves per SIMD (128 VGPRs, two workgroups per CU; 40 VGPRs spilled) against the product build (166 VGPRs, one workgroup per CU)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03; mkdir -p $O
run() { python3 bench.py --no-cpu-baseline --no-e2e --no-configs --reps 1 --kernel-reps 4 --steps 20 --warmup 5 --model 70b --ctx 16384 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['kernel_us']['mean'], d['config'].get('tiles_per_workgroup'))"; }
for nt in 0 2 4; do
  export PKV_DECODE_NTILE=$nt
  unset PKV_LIB_PATH; run "product ntile=$nt"
  export PKV_LIB_PATH=$PWD/llama.cpp_test_amd/libmi355_paged_kv_occ4.so; run "occ4    ntile=$nt"
done 2>&1 | tee $O/occ_ab.txt
