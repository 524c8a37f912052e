"""-m gpu: the REFERENCE's own parity harness (tests/test-backend-ops.cpp, compiled in place into oracle/_ref by
`make -C oracle ref`) driving our kernels through the ggml-backend plugin (llama.cpp_test_amd/libggml-mi355.so, loaded
via GGML_BACKEND_PATH exactly as ggml_backend_load_all does, /root/reference/ggml/src/ggml-backend-reg.cpp:581-585).
Every case our device reports as supported must pass the harness's own criterion (NMSE vs the reference CPU backend:
5e-4 for FLASH_ATTN_EXT and MUL_MAT, 1e-7 for ROPE and the element-wise ops, 1e-6 for CPY).  Skipped where the prebuilt binaries did not travel."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TBO = os.path.join(ROOT, "oracle", "_ref", "test-backend-ops")
PLUGIN = os.path.join(ROOT, "llama.cpp_test_amd", "libggml-mi355.so")


@pytest.mark.skipif(not (os.path.exists(TBO) and os.path.exists(PLUGIN)), reason="oracle/_ref/test-backend-ops or the plugin not built")
@pytest.mark.parametrize("op,min_ok", [("FLASH_ATTN_EXT", 960),     # ALL of them (r03: what the paged kernels decline goes to pkv_flash_attn_ext_flat)
                                       ("ROPE", 150), ("ROPE_BACK", 150), ("CPY", 28),
                                       # first slice of the rest of the layer: harness bars 1e-7 / 5e-4 (MUL_MAT)
                                       ("RMS_NORM", 3), ("ADD", 10), ("MUL", 10), ("SILU", 2), ("MUL_MAT", 30),
                                       # the softmax of the non-flash-attention path (a13): mask f32 / f16, scale, ALiBi; harness bar 1e-6
                                       ("SOFT_MAX", 40), ("CONT", 2)])
def test_reference_harness(op, min_ok):
    env = dict(os.environ, GGML_BACKEND_PATH=PLUGIN)
    r = subprocess.run([TBO, "test", "-o", op], env=env, capture_output=True, text=True, timeout=900)
    out = re.sub(r"\x1b\[[0-9;]*m", "", r.stdout + r.stderr)
    assert "Backend MI355: OK" in out, out[-2000:]
    ok = len(re.findall(rf"{op}\(.*\): OK", out))
    assert "FAIL" not in out
    assert ok >= min_ok, (ok, out[-1000:])
    assert r.returncode == 0
