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


@pytest.mark.parametrize("preset,seconds,devices", [("small", 90.0, "0,0,0"), ("tiny", 60.0, "0,0")])
def test_in_library_later_chunks_are_sharded_too_and_decoding_overlaps(preset, seconds, devices):
    """Round 4: not only a stream's first chunk - every chunk that starts on a token boundary and holds >= 16 tokens per engine goes
    through all engines (three feeds of a third of the clip each, processing interval 2 s: three sharded chunks, the second and
    third continuing from the stream engine's K/V rings and conv history), and the stream engine's decoder no longer waits for
    the whole wavefront: it waits for a shard's adapter rows right in front of the first step that reads them.  Ids = the single
    engine's; VOX_HIP_DISABLE=multi_overlap (the round-3 waits) gives the same ids."""
    import ctypes as C
    import voxtral_c_amd as v
    v.hip.vox_hip_pending_fences.argtypes = [C.c_void_p]
    audio = synth_speech(seconds, 56)
    third = (len(audio) // 3 // 1280) * 1280
    feeds = [third, third, len(audio)]
    win = {} if preset != "tiny" else dict(enc_window=48, dec_window=64)
    with v.Model(model_dir(preset), **win) as m:
        want = m.transcribe(audio, feed_sizes=feeds)["tokens"]
        assert m.ctx.n_sharded_chunks == 0
    for no_overlap in (False, True):
        os.environ["VOX_DEVICES"] = devices
        if no_overlap:
            os.environ["VOX_HIP_DISABLE"] = "multi_overlap"
        try:
            with v.Model(model_dir(preset), **win) as mm:
                got = mm.transcribe(audio, feed_sizes=feeds)["tokens"]
                n_sharded = mm.ctx.n_sharded_chunks
                assert v.hip.vox_hip_pending_fences(mm.engine) == 0
        finally:
            del os.environ["VOX_DEVICES"]
            os.environ.pop("VOX_HIP_DISABLE", None)
        assert len(want) > 300 and np.array_equal(got, want), (no_overlap, len(got), len(want))
        assert n_sharded == 3, n_sharded


@pytest.mark.parametrize("preset,devices", [("small", "0,0,0"), ("tiny", "0,0")])
def test_in_library_larger_chunk_right_behind_a_sharded_one(preset, devices):
    """A sharded chunk leaves the last engine's state push (K/V rings, conv history rows, encoder rows) in flight on THAT engine's
    stream; the next feed is eight times larger, so the stream engine's conv / mel / encoder-row buffers must grow - and a growing
    buffer is freed and re-allocated.  The pending peer copies have to land first (`settle_enc_fences` in `ensure` / `ensure_keep`,
    round-4 advisor finding): otherwise the new buffer keeps stale conv history.  Fresh Model per run, so the buffers start small."""
    import voxtral_c_amd as v
    audio = synth_speech(90.0, 57)
    first = (10 * 16000 // 1280) * 1280
    feeds = [first, len(audio)]
    win = {} if preset != "tiny" else dict(enc_window=48, dec_window=64)
    with v.Model(model_dir(preset), **win) as m:
        want = m.transcribe(audio, feed_sizes=feeds)["tokens"]
    os.environ["VOX_DEVICES"] = devices
    try:
        for _ in range(2):
            with v.Model(model_dir(preset), **win) as mm:
                got = mm.transcribe(audio, feed_sizes=feeds)["tokens"]
                assert mm.ctx.n_sharded_chunks == 2, mm.ctx.n_sharded_chunks
            assert len(want) > 300 and np.array_equal(got, want), (len(got), len(want))
    finally:
        del os.environ["VOX_DEVICES"]


def _bench_json(cmd, env, timeout=1500, want_rc=0):
    import json
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert (r.returncode == 0) == (want_rc == 0), (r.returncode, r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_rccl_backend_single_rank_bench_path(tmp_path):
    """The RCCL ("nccl") code path of bench.py --gpus N with the only world size a 1-GPU box allows: GPU-resident staging
    tensors handed to the engine by data_ptr, the engine's stream as torch's current stream (ExternalStream), a grouped RCCL
    send + recv to self per layer (VOX_DIST_SELF_LOOP: export -> RCCL -> import, all stream-ordered), device-side adapter
    append, all_reduce of the timing - and NO host wait on the engine stream between shard_begin and shard_end."""
    env = dict(os.environ, VOX_FORCE_DIST="1", VOX_DIST_SELF_LOOP="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--preset", "small", "--seconds", "20"]
    out = _bench_json(cmd, env)
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["decoder_steps_per_pass"] > 100
    assert out["config"]["backend"] == "nccl" and out["rccl_ranks"] == 1
    assert out["host_syncs_in_wavefront"] == 0, out
    assert out["replica"]["value"] > 0 and set(out["phases_ms"]) >= {"encode", "gather", "prefill", "decode"}


def test_bench_self_launches_two_ranks_and_matches_the_reference_golden():
    """`python bench.py --gpus 2` exactly as the driver calls it (no torch.distributed.run around it): the script launches
    its own ranks.  Two ranks share this box's one GPU (VOX_SHARE_GPU: gloo, host-staged halo), FULL geometry, the golden
    30 s night1968 clip on every rank: the 2-way sharded encoder at the 4B width must reproduce the reference's 386 ids on
    both ranks, and so must the replica pass."""
    import voxtral_c_amd as v
    if v.device_count() < 1:
        pytest.fail("no HIP device")
    env = dict(os.environ, VOX_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    out = _bench_json(cmd, env)
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["backend"] == "gloo"
    assert out["parity"]["checked"] and out["parity"]["checked_ranks"] == 2 and out["parity"]["mismatches_all_ranks"] == 0, out["parity"]
    assert out["decoder_steps_per_pass"] == 2 * 386
    assert out["replica"]["parity_checked_ranks"] == 2 and out["replica"]["parity_mismatches_all_ranks"] == 0
    assert out["replica"]["value"] > 0 and out["value"] > 0
    # round 5: the same line carries BASELINE config 4 as a second phase - ONE clip of 2 x 75 s, encoder sharded over both ranks,
    # decoder on rank 0 - checked against the first ids of the reference's 600 s run (the 150 s clip is a prefix of that input)
    c4 = out["config4"]
    assert c4["completed"] and c4["audio_seconds"] == 150.0 and c4["value"] > 0 and c4["decoder_steps"] > 1800, c4
    assert c4["parity"]["checked"] and c4["parity"]["mismatches"] == 0 and c4["parity"]["steps"] > 1780, c4["parity"]
    assert set(c4["phases_ms"]) >= {"encode", "prefill", "decode"}


def test_bench_reports_no_value_when_the_sharded_pass_fails():
    """The RCCL point-to-point path cannot run with more than one rank before the first multi-GPU node does: if it raises (or
    hangs past its time limit) the job ends with ONE honest JSON line - value null, the reason, the replica figure only
    under `replica` - and a non-zero exit code: a broken multi-GPU path must not look like a green run."""
    env = dict(os.environ, VOX_FORCE_DIST="1", VOX_DIST_INJECT_FAIL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--preset", "small", "--seconds", "12"]
    out = _bench_json(cmd, env, want_rc=4)
    assert out["sharded_pass"]["completed"] is False and "injected" in out["sharded_pass"]["reason"]
    assert out["value"] is None and out["replica"]["value"] > 0 and "failed" in out["config"]["parallelism"]


# ------------------------------------------------------------------------------------------------------------------------------------
# Round 6: the tests of a node that has more than one GPU.  They are collected everywhere and switch themselves on when a second device
# is visible (the pool's boxes have one): cross-device peer copies (hipMemcpyPeerAsync / hipDeviceEnablePeerAccess) and an RCCL
# communicator with two ranks have only ever seen same-device peers, a 1-rank self-loop and gloo.
# ------------------------------------------------------------------------------------------------------------------------------------
def _need_two_devices():
    import voxtral_c_amd as v
    n = v.device_count()
    if n < 2:
        pytest.skip(f"needs >= 2 visible HIP devices (this box has {n}); runs by itself on a multi-GPU node")


@pytest.mark.parametrize("disable", ["", "peer"])
def test_two_real_devices_in_library_encoder_matches_the_reference_golden(disable):
    """VOX_DEVICES=0,1 (host/vox_multi.c): the encoder of the headline clip sharded over two PHYSICAL GPUs - weights cloned GPU to GPU,
    K/V tails pushed over xGMI behind every layer, adapter rows written into the decoding engine's buffer - must give the reference's
    386 ids.  Second case: VOX_HIP_DISABLE=peer = a node whose GPUs cannot map each other's memory (hipDeviceCanAccessPeer == 0):
    no peer mapping is enabled and the runtime stages the copies through the host - slower, same ids."""
    _need_two_devices()
    import voxtral_c_amd as v
    g = np.load(os.path.join(ROOT, "tests", "golden", "stream_full_batch.npz"), allow_pickle=True)
    audio = g["audio_i16"].astype(np.float32) / 32768.0
    env = {"VOX_DEVICES": "0,1"}
    if disable:
        env["VOX_HIP_DISABLE"] = disable
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        with v.Model(model_dir("full")) as mm:
            assert mm.ctx.n_shard_engines == 2
            got = mm.transcribe(audio)["tokens"]
            half = len(audio) // 2
            got2 = mm.transcribe(audio, feed_sizes=[half, 16000, 16000, len(audio)])["tokens"]
    finally:
        for k, val in saved.items():
            if val is None:
                del os.environ[k]
            else:
                os.environ[k] = val
    assert np.array_equal(np.asarray(got), g["tokens"]) and np.array_equal(np.asarray(got2), g["tokens"])


def test_two_real_devices_rccl_bench_line_matches_the_reference_golden():
    """`python bench.py --gpus 2` on two physical GPUs: one rank per GPU over RCCL (no VOX_SHARE_GPU), per-layer halo isend / irecv on the
    engines' streams, the gather of adapter rows; both ranks' ids = the reference's, and the config-4 phase (one 150 s clip, decoder on
    rank 0) = the first ids of the reference's 600 s run."""
    _need_two_devices()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "VOX_SHARE_GPU"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    out = _bench_json(cmd, env)
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["backend"] == "nccl", out.get("backend")
    assert out["parity"]["checked"] and out["parity"]["checked_ranks"] == 2 and out["parity"]["mismatches_all_ranks"] == 0, out["parity"]
    assert out["host_syncs_in_wavefront"] == 0, out
    c4 = out["config4"]
    assert c4["completed"] and c4["parity"]["checked"] and c4["parity"]["mismatches"] == 0, c4
