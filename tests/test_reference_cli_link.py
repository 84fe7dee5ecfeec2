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


REF_FIELDS = """encoder encoder.conv1_bias encoder.layers[0].wq_weight_bf16 encoder.layers[5].w2_bias encoder.layers[31].ffn_norm
encoder.norm adapter adapter.linear1_weight_bf16 decoder decoder.tok_embeddings_bf16 decoder.layers[0].ada_norm_down
decoder.layers[3].wo_weight_bf16 decoder.layers[25].ffn_norm decoder.norm safetensors model_dir kv_cache_k kv_cache_len
kv_cache_max kv_pos_offset delay_tokens t_cond ada_scale use_bf16 enc_kv_cache_k enc_kv_cache_len enc_kv_cache_max
enc_kv_pos_offset enc_inc_cap enc_inc_rope_freqs dec_x dec_rope_freqs""".split()
REF_TYPES = ["vox_enc_layer_t", "vox_encoder_t", "vox_dec_layer_t", "vox_decoder_t", "vox_adapter_t"]


def _layout(tmp_path, incdir, tag):
    src = tmp_path / f"layout_{tag}.c"
    body = "".join(f'    printf("{f} %zu\\n", offsetof(vox_ctx_t, {f}));\n' for f in REF_FIELDS)
    body += "".join(f'    printf("sizeof({t}) %zu\\n", sizeof({t}));\n' for t in REF_TYPES)
    src.write_text('#include "voxtral.h"\n#include <stdio.h>\n#include <stddef.h>\nint main(void) {\n' + body + "    return 0;\n}\n")
    exe = tmp_path / f"layout_{tag}"
    subprocess.check_call(["gcc", "-I" + incdir, str(src), "-o", str(exe)])
    return subprocess.check_output([str(exe)], text=True)


@pytest.mark.skipif(not os.path.exists("/root/reference/voxtral.h"), reason="reference checkout not present")
def test_vox_ctx_starts_with_the_reference_layout(tmp_path):
    """vox_enc_layer_t ... vox_adapter_t are declared with the reference's sizes and every field of the
    reference's vox_ctx_t sits at the reference's offset in ours (the engine's own fields are appended behind them): code
    compiled against the reference header reads the right bytes."""
    ours = _layout(tmp_path, os.path.join(ROOT, "include"), "ours")
    ref = _layout(tmp_path, "/root/reference", "ref")
    assert ours == ref, "\n".join(f"{a}   |   {b}" for a, b in zip(ours.splitlines(), ref.splitlines()) if a != b)
