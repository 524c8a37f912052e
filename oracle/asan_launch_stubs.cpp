// oracle/asan_launch_stubs.cpp — TEST INFRASTRUCTURE (the sanitizer build of the host side, `make -C oracle asan`): the kernel launchers of the product library
// (csrc/*.hip) are not part of a host-only build; every one of them answers PKV_ERR_NO_DEVICE here, so that pkv_capi.cpp + pkv_cache.cpp — the whole host side of the
// C ABI: the llama_kv_cache_* wrappers, find_slot's range checks, the session-file metadata, the cache view, the fork toys — link into a library that g++ builds with
// -fsanitize=address,undefined and the CPU suite drives (tests/test_block_manager.py, test_kv_iface.py, test_state_io.py, test_fork_toys.py).  Never shipped, never
// loaded by the product path.  (Generated from the undefined symbols of the host objects; signatures = csrc/pkv_common.h.)
#include "../llama.cpp_test_amd/csrc/pkv_common.h"

namespace pkv {

int launch_gemv(pkv_gemv const&, hipStream_t) { return PKV_ERR_NO_DEVICE; }
int launch_silu(float const*, float*, long, hipStream_t) { return PKV_ERR_NO_DEVICE; }
int launch_binary(int, float const*, float const*, float*, long, long, long, hipStream_t) { return PKV_ERR_NO_DEVICE; }
int launch_k_shift(void*, int, int const*, int const*, int const*, int, int, int, int, pkv::RopeDev const&, hipStream_t) { return PKV_ERR_NO_DEVICE; }
int launch_mul_mat(void const*, int, float const*, float*, long, long, long, hipStream_t) { return PKV_ERR_NO_DEVICE; }
int launch_rows_io(void*, int, int, void*, int const*, int const*, int, int, int, int, bool, hipStream_t, long, long) { return PKV_ERR_NO_DEVICE; }
size_t act_image_bytes(long) { return 0; }
int launch_get_rows(void const*, int, int const*, float*, long, long, long, hipStream_t) { return PKV_ERR_NO_DEVICE; }
int launch_kv_store(float*, int, float const*, float const*, void*, void*, int, int, int const*, int const*, int const*, pkv::RopeDev const*, int, int, int, int, hipStream_t, float const*) { return PKV_ERR_NO_DEVICE; }
int launch_rms_norm(float const*, float*, long, long, float, hipStream_t) { return PKV_ERR_NO_DEVICE; }
int launch_rope_f16(void const*, int const*, void*, int, int, int, pkv::RopeDev const&, int, hipStream_t, long, long) { return PKV_ERR_NO_DEVICE; }
int launch_rope_f32(float const*, int const*, float*, int, int, int, pkv::RopeDev const&, int, hipStream_t, long, long) { return PKV_ERR_NO_DEVICE; }
int launch_silu_mul(float const*, float const*, float*, float*, long, hipStream_t) { return PKV_ERR_NO_DEVICE; }
int launch_soft_max(float const*, void const*, int, float*, long, long, long, long, float, float, hipStream_t) { return PKV_ERR_NO_DEVICE; }
int launch_pack_rows(void const*, void*, int, long, long, int, hipStream_t) { return PKV_ERR_NO_DEVICE; }
int launch_copy_cells(void*, int, int, int const*, int const*, int const*, int const*, int, int, int, int, hipStream_t) { return PKV_ERR_NO_DEVICE; }
int launch_rope_cache(int const*, float*, int, int, pkv::RopeDev const&, hipStream_t) { return PKV_ERR_NO_DEVICE; }
int launch_attn_decode(pkv_attn_args const&, hipStream_t, pkv_step_args const*) { return PKV_ERR_NO_DEVICE; }
int launch_copy_blocks(void*, unsigned long, int const*, int const*, int, hipStream_t) { return PKV_ERR_NO_DEVICE; }
int launch_cpy_strided(void const*, int, long const*, long const*, void*, int, long const*, long const*, hipStream_t) { return PKV_ERR_NO_DEVICE; }
int launch_attn_flat(pkv_flat_attn_args const&, hipStream_t) { return PKV_ERR_NO_DEVICE; }
size_t flat_workspace_bytes(int, int, int, int) { return 0; }
int launch_quantize_blocks(float const*, void*, int, long, long, hipStream_t) { return PKV_ERR_NO_DEVICE; }
void set_prefill_stamps(void*) {}
int launch_attn_prefill(pkv_attn_args const&, hipStream_t, float const*) { return PKV_ERR_NO_DEVICE; }
int launch_convert_rows(float const*, void*, int, long, long, hipStream_t) { return PKV_ERR_NO_DEVICE; }
int launch_quantize_act(float const*, void*, long, hipStream_t) { return PKV_ERR_NO_DEVICE; }
int launch_rms_norm_mul(float const*, float const*, float*, float const*, float*, float*, long, long, float, hipStream_t) { return PKV_ERR_NO_DEVICE; }
int launch_mul_mat_multi(int, void const* const*, int, float const*, float* const*, long const*, long, long, hipStream_t) { return PKV_ERR_NO_DEVICE; }
int launch_rms_norm_mul_q(float const*, float const*, float*, float const*, float*, void*, long, float, hipStream_t) { return PKV_ERR_NO_DEVICE; }
size_t decode_workspace_bytes(int, int, int, int) { return 0; }
int launch_mul_mat_strided(void const*, int, long const*, long const*, float const*, long const*, long const*, float*, hipStream_t) { return PKV_ERR_NO_DEVICE; }
int launch_qkv_store_mirror(pkv_qkv_store_args const*, pkv::RopeDev const&, hipStream_t) { return PKV_ERR_NO_DEVICE; }
size_t prefill_workspace_bytes(int, int, int) { return 0; }
int launch_attn_decode_stamps(pkv_attn_args const&, pkv_step_args const*, unsigned long long *, int, hipStream_t) { return PKV_ERR_NO_DEVICE; }
int launch_convert_rows_mirror(float const*, void*, int, void*, int, long, long, int, int, int, hipStream_t) { return PKV_ERR_NO_DEVICE; }
int launch_rope_convert_mirror(float const*, long, long, int const*, float*, void*, int, void*, long, long, int, int, int, pkv::RopeDev const&, hipStream_t) { return PKV_ERR_NO_DEVICE; }
int make_rope(pkv_rope_params const*, int, pkv::RopeDev*) { return PKV_ERR_NO_DEVICE; }
}  // namespace pkv
