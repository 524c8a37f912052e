#!/bin/bash
# timing-only builds of the decode-step products (results wrong where noted): probes/bin/libpkv_gemv_abl{1,2,3,4}.so
#   1 = no dots (loads + reductions), 2 = no column prologue, 3 = no requests past a wave's last unit (behind a branch), 4 = two units ahead of the prologue
set -e
cd "$(dirname "$0")/../llama.cpp_test_amd/csrc"
make -j8 >/dev/null
mkdir -p ../../probes/bin
for a in "$@"; do
  if [ "$a" = stamps ]; then FLAG="-DPKV_GEMV_STAMPS=2"; elif [ "$a" = stamps1 ]; then FLAG="-DPKV_GEMV_STAMPS=1"; else FLAG="-DPKV_GEMV_ABLATE=$a"; fi
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=8 $FLAG -c pkv_gemv.hip -o /tmp/gemv_abl$a.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../probes/bin/libpkv_gemv_abl$a.so build/pkv_capi.cpp.o build/pkv_cache.cpp.o build/pkv_attn_decode.hip.o \
        build/pkv_attn_prefill.hip.o build/pkv_attn_flat.hip.o build/pkv_kv_ops.hip.o build/pkv_layer_ops.hip.o /tmp/gemv_abl$a.o build/pkv_gemm.hip.o ) &
done
wait
