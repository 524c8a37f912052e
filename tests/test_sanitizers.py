"""CPU: the host side of the product (csrc/pkv_cache.cpp = the block manager, csrc/pkv_capi.cpp = every C-ABI wrapper's host code) built by plain g++ with
-fsanitize=address,undefined (`make -C oracle asan`; the kernel launchers answer PKV_ERR_NO_DEVICE) and the block-manager suites run against it — find_slot incl. the
evict-own-block and pinned-ubatch cases, copy-on-write, seq_add on a full pool, session-file metadata, the cache view, the fork toys, header == exports.  The
reference's counterpart: LLAMA_SANITIZE_ADDRESS / _UNDEFINED (/root/reference/CMakeLists.txt:71-73).  Round 1's heap over-read in find_slot was found by hand;
this is the net under it.  Never run on the GPU box (GPU sanitizers are not available there; this build has no device code at all)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None or not os.path.isdir("/opt/rocm/include"), reason="needs g++ and the HIP headers")
def test_block_manager_suites_under_asan_ubsan():
    env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "PKV_LIB_PATH")}
    p = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan-test"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-4000:]
    assert " passed" in p.stdout and "AddressSanitizer" not in p.stdout and "runtime error" not in p.stdout, p.stdout[-4000:]
