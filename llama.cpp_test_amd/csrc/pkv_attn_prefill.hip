// pkv_attn_prefill.hip — batched-prefill paged attention (flash-style, MFMA).  Placeholder until the MFMA kernel lands.
#include "pkv_common.h"

namespace pkv {

size_t prefill_workspace_bytes(int, int, int) { return 0; }

int launch_attn_prefill(const pkv_attn_args &, hipStream_t) {
    set_error("pkv_attn_prefill: not built yet");
    return PKV_ERR_UNSUPPORTED;
}

}  // namespace pkv
