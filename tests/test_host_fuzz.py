"""CPU: the host-side parsers (WAV, safetensors header, tekken.json) under AddressSanitizer + UBSan
with mutated inputs (tools/fuzz_host.c).  They read user-supplied files; a sanitizer report or a
crash fails the test."""
import json
import os
import struct
import subprocess
import wave

import numpy as np
import pytest

from audio_util import synth_speech
from conftest import ROOT, model_dir

HOST = os.path.join(ROOT, "voxtral_c_amd", "host")


@pytest.fixture(scope="module")
def fuzz_bin(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("fuzz") / "fuzz_host")
    cmd = ["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-I" + os.path.join(ROOT, "include"), "-I" + HOST,
           os.path.join(ROOT, "tools", "fuzz_host.c"), os.path.join(ROOT, "tools", "fuzz_stubs.c"),
           os.path.join(HOST, "vox_safetensors.c"), os.path.join(HOST, "vox_tokenizer.c"), os.path.join(HOST, "vox_audio.c"),
           "-lm", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build unavailable: " + r.stderr[-300:])
    return out


def _seeds(tmp_path):
    wav = str(tmp_path / "seed.wav")
    pcm = np.clip(np.round(synth_speech(0.5, 3) * 32767), -32768, 32767).astype("<i2")
    with wave.open(wav, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm.tobytes())
    tensors = {"mm_streams_embeddings.embedding_module.tok_embeddings.weight": ("BF16", [64, 16]),
               "layers.0.attention.wq.weight": ("BF16", [32, 16]), "norm.weight": ("F32", [16])}
    hdr, off, blobs = {}, 0, []
    for name, (dt, shape) in tensors.items():
        n = int(np.prod(shape)) * (2 if dt == "BF16" else 4)
        hdr[name] = {"dtype": dt, "shape": shape, "data_offsets": [off, off + n]}
        blobs.append(bytes((i * 37) & 0xff for i in range(n)))
        off += n
    hj = json.dumps(hdr).encode()
    hj += b" " * ((8 - len(hj) % 8) % 8)
    st = str(tmp_path / "seed.safetensors")
    with open(st, "wb") as f:
        f.write(struct.pack("<Q", len(hj)) + hj + b"".join(blobs))
    import base64
    voc = [{"rank": i, "token_bytes": base64.b64encode((" t%d" % i).encode() if i else b"\x00").decode(), "token_str": None}
           for i in range(12)]
    voc[3]["token_bytes"] = base64.b64encode("\u00e9\\u00e9 \"q\"".encode()).decode()
    sp = [{"rank": i, "token_str": "<SPECIAL_%d>" % i, "is_control": True} for i in range(6)]
    sp[1]["token_str"] = "<s>\\u0041\\n"
    sp[2]["token_str"] = "</s>"
    small = str(tmp_path / "tekken_small.json")
    with open(small, "w") as f:
        json.dump({"config": {"default_vocab_size": 18, "default_num_special_tokens": 6}, "vocab": voc, "special_tokens": sp}, f)
    return wav, st, (os.path.join(model_dir("tiny"), "tekken.json"), small)


@pytest.mark.parametrize("seed,which,iters", [(1, 0, 300), (32, 1, 3000)])
def test_parsers_survive_mutated_files(fuzz_bin, tmp_path, seed, which, iters):
    """Regressions found this way: a zero WAV sample rate divided by zero in the resampler (as in the
    reference), a negative safetensors data offset passed the bounds check, an empty header reached
    qsort with a null base."""
    wav, st, tks = _seeds(tmp_path)
    r = subprocess.run([fuzz_bin, wav, st, tks[which], str(iters), str(seed)], capture_output=True, timeout=600)
    out, err = r.stdout.decode("utf-8", "replace"), r.stderr.decode("utf-8", "replace")   # parsers echo mutated names
    assert r.returncode == 0, (out[-500:], err[-3000:])
    assert "no sanitizer report" in out
