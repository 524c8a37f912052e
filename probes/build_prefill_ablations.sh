#!/bin/bash
# timing-only builds of the prefill kernel (headline instantiation only: -DPKV_PREFILL_DEV) with parts of the loop removed
# (-DPKV_PREFILL_ABLATE=mask, see pkv_attn_prefill.hip) -> probes/bin/libpkv_abl_<mask>.so; run with probes/r02_p_abl.sh on the GPU box
set -e
cd "$(dirname "$0")/../llama.cpp_test_amd/csrc"; mkdir -p ../../probes/bin
make -s >/dev/null
for ab in ${ABL:-0 1 2 4 8 16 24 6 64 256}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result -Wno-pass-failed -mllvm -amdgpu-kernarg-preload-count=8 -DPKV_PREFILL_DEV -DPKV_PREFILL_ABLATE=$ab -c pkv_attn_prefill.hip -o /tmp/pf_abl.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../probes/bin/libpkv_abl_$ab.so build/pkv_capi.cpp.o build/pkv_cache.cpp.o build/pkv_attn_decode.hip.o /tmp/pf_abl.o build/pkv_kv_ops.hip.o build/pkv_layer_ops.hip.o build/pkv_gemv.hip.o build/pkv_gemm.hip.o
done
ls ../../probes/bin/libpkv_abl_*.so
