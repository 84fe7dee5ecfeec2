"""GPU: the real HIP shard path (vox_hip_shard_*) under torch.distributed with 2 and 3 ranks.
A single-GPU box cannot host an RCCL communicator with several ranks on one device, so the
ranks share GPU 0 and exchange through gloo (host tensors mirrored by device buffers); the
engine calls, the K/V wavefront, the gather and the rank-0 decode are exactly those of the
RCCL run.  The tokens must equal the single-process transcription of the same clip."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from audio_util import synth_speech
from conftest import ROOT, model_dir

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("preset,seconds,world", [("tiny", 14.0, 3), ("small", 70.0, 2)])
def test_sharded_transcription_equals_single_gpu(tmp_path, preset, seconds, world):
    import voxtral_c_amd as v
    if v.device_count() < 1:
        pytest.fail("no HIP device")
    out = str(tmp_path / "toks.npy")
    env = dict(os.environ, VOX_DIST_BACKEND="gloo", VOX_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "mgpu_worker.py"), preset, str(seconds), "77", out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    sharded = np.load(out)
    win = {} if preset != "tiny" else dict(enc_window=48, dec_window=64)
    with v.Model(model_dir(preset), **win) as m:
        single = m.transcribe(synth_speech(seconds, 77))["tokens"]
    assert len(sharded) == len(single), (len(sharded), len(single))
    assert np.array_equal(sharded, single)


@pytest.mark.parametrize("preset,seconds,world", [("tiny", 12.0, 2)])
def test_one_clip_per_rank_sharded_encoders_parallel_decoders(tmp_path, preset, seconds, world):
    """bench.py --gpus N workload: N clips, each clip's encoder sharded over all ranks and gathered to
    its owner, every rank decodes its own clip.  Each rank's tokens = single-GPU transcription."""
    import voxtral_c_amd as v
    out = str(tmp_path / "toks")
    env = dict(os.environ, VOX_DIST_BACKEND="gloo", VOX_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "mgpu_worker.py"), preset, str(seconds), "91", out, "many"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    win = {} if preset != "tiny" else dict(enc_window=48, dec_window=64)
    with v.Model(model_dir(preset), **win) as m:
        for rank in range(world):
            single = m.transcribe(synth_speech(seconds, 91 + rank))["tokens"]
            got = np.load(f"{out}.{rank}.npy")
            assert np.array_equal(got, single), rank


@pytest.mark.parametrize("preset,seconds,devices", [("small", 40.0, "0,0"), ("small", 70.0, "0,0,0"), ("tiny", 30.0, "0,0,0,0")])
def test_in_library_multi_device_encoder_matches_single_gpu(preset, seconds, devices):
    """libvoxtral's own multi-GPU path (VOX_DEVICES / vox_load_opts_t.devices, host/vox_multi.c): no Python, no torch -
    N engines in one process, the first chunk's encoder positions split over them, K/V tails pushed engine to engine
    behind each layer (peer copy + event, stream-ordered), adapter rows written into the stream engine's buffer, encoder
    state handed back for the incremental tail.  On a 1-GPU box all engines sit on device 0 (the hand-offs are then plain
    D2D copies; events, ordering and bookkeeping are those of N GPUs).  Token ids must equal the single-engine run, for one
    big feed and for a feed pattern whose first chunk is large and whose later chunks are incremental."""
    import voxtral_c_amd as v
    audio = synth_speech(seconds, 55)
    win = {} if preset != "tiny" else dict(enc_window=48, dec_window=64)
    with v.Model(model_dir(preset), **win) as m:
        want = m.transcribe(audio)["tokens"]
        half = len(audio) // 2
        want2 = m.transcribe(audio, feed_sizes=[half, 16000, 16000, len(audio)])["tokens"]
    os.environ["VOX_DEVICES"] = devices
    try:
        with v.Model(model_dir(preset), **win) as mm:
            assert mm.ctx.n_shard_engines == len(devices.split(","))
            got = mm.transcribe(audio)["tokens"]
            got_again = mm.transcribe(audio)["tokens"]
            got2 = mm.transcribe(audio, feed_sizes=[half, 16000, 16000, len(audio)])["tokens"]
    finally:
        del os.environ["VOX_DEVICES"]
    assert len(want) > 100
    assert np.array_equal(got, want) and np.array_equal(got_again, want)
    assert np.array_equal(got2, want2) and np.array_equal(want2, want)


def test_rccl_backend_single_rank_bench_path(tmp_path):
    """The RCCL ("nccl") code path of bench.py --gpus N (GPU-resident staging tensors handed to
    the engine by data_ptr, device-side adapter append, all_reduce of the timing) with the only
    world size a 1-GPU box allows."""
    import json
    env = dict(os.environ, VOX_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--preset", "small", "--seconds", "20"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["decoder_steps_per_pass"] > 100
    assert out["config"]["backend"] == "nccl"
