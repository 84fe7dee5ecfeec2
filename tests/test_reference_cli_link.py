"""CPU: the unmodified reference CLI (main.c) compiles and links against libvoxtral.so.

Only runs where the reference checkout exists (the build container); nothing is copied — the
compiler reads main.c from /root/reference and our headers from include/."""
import os
import subprocess

import pytest

from conftest import ROOT

REF_MAIN = "/root/reference/main.c"


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="reference checkout not present")
def test_reference_main_links_against_libvoxtral(tmp_path):
    exe = tmp_path / "voxtral_hip"
    libdir = os.path.join(ROOT, "voxtral_c_amd")
    cmd = ["gcc", "-O2", "-I" + os.path.join(ROOT, "include"), REF_MAIN, "-o", str(exe),
           "-L" + libdir, "-lvoxtral", "-Wl,-rpath," + libdir, "-lm"]
    subprocess.check_call(cmd)
    out = subprocess.run([str(exe), "-h"], capture_output=True, text=True)
    assert out.returncode == 0
    assert "Usage" in out.stderr
    # without a GPU the CLI must fail at vox_load, loudly, not fall back to anything
    import voxtral_c_amd as v
    if v.device_count() == 0:
        from conftest import model_dir
        r = subprocess.run([str(exe), "-d", model_dir("tiny"), "-i", "/nonexistent.wav"], capture_output=True, text=True)
        assert r.returncode != 0 and "no HIP device" in r.stderr
