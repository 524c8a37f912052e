"""The RoPE kernels restate the host libm's cosf / sinf (pkv_rowcodec.h: libm_cosf / libm_sinf) so that device RoPE is bit-identical to
the reference's.  oracle/check_trig.c holds the same restatement in plain C: here it is compiled and compared with this host's cosf / sinf
over the RoPE angles of the tested geometries and a strided sweep of all finite floats (CPU only, a few seconds)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_libm_sincosf_restatement_matches_host_libm(tmp_path):
    exe = str(tmp_path / "check_trig")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "oracle", "check_trig.c"), "-lm"])
    out = subprocess.run([exe, "4099"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "cos mm=0 sin mm=0" in out.stdout
