#!/usr/bin/env python3
"""registers / spills / LDS of every kernel in a hipcc object, read from the code object's metadata notes (no recompile):
   python3 probes/co_resources.py llama.cpp_test_amd/csrc/build/pkv_attn_decode.hip.o [name filter]"""
import re, subprocess, sys, tempfile, os
B = "/opt/rocm/lib/llvm/bin"
obj, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
with tempfile.TemporaryDirectory() as d:
    fb, co = os.path.join(d, "fb"), os.path.join(d, "co")
    subprocess.check_call([f"{B}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fb])
    subprocess.check_call([f"{B}/clang-offload-bundler", "--type=o", f"--input={fb}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}", "--unbundle"])
    t = subprocess.run([f"{B}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
for k in re.split(r"\n\s+- \.agpr_count:", t)[1:]:
    name = re.search(r"\.name:\s+(\S+)", k).group(1)
    g = lambda f: re.search(r"\." + f + r":\s+(\d+)", k).group(1)
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"pkv::\(anonymous namespace\)::", "", dn)
    dn = re.sub(r"\(.*", "", dn)
    if flt in dn:
        print(f"{dn[:110]:110s} vgpr {g('vgpr_count'):>3} agpr {k.split()[0]:>3} sgpr {g('sgpr_count'):>3} spill {g('vgpr_spill_count'):>3} scratch {g('private_segment_fixed_size'):>4} lds {g('group_segment_fixed_size'):>6}")
