"""CPU (gloo, world_size 2 and 3): the multi-GPU orchestration of voxtral_c_amd/multi_gpu.py —
shard planning, mel halo, the per-layer K/V wavefront and the adapter gather — with the numpy
oracle standing in for the GPU behind the same ShardEngine interface.  The sharded result
must equal the single-process oracle run over the whole clip."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, model_dir

sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleShardEngine:
    """ShardEngine on top of oracle/vox_oracle.py primitives (test infrastructure only)."""

    def __init__(self, mdir, dims):
        from oracle import vox_oracle as vo
        self.vo, self.o, self.d = vo, vo.Oracle(mdir, dims), dims
        self.kv_dim = dims.enc_heads * dims.enc_head_dim
        self.n_layers, self.window, self.dec_dim = dims.enc_layers, dims.enc_window, dims.dec_dim
        self.mel = None

    def reset(self):
        self.o.reset_encoder()

    def queue_mel(self, padded, frame0, n_frames):
        self.mel = self.vo.mel_frames(padded[frame0 * 160:], n_frames)

    def begin(self, n_mel, discard, pos0):
        self.o.reset_encoder()
        x = self.o.conv_stem(self.mel[:n_mel])
        self.x = x[discard:].copy()
        self.pos0 = pos0
        self.tail_k = [None] * self.n_layers
        self.tail_v = [None] * self.n_layers
        self.own_k = [None] * self.n_layers
        self.own_v = [None] * self.n_layers
        return self.x.shape[0]

    def kv_import(self, l, pos_first, n, buf):
        a = buf.numpy()
        self.tail_k[l], self.tail_v[l] = a[0, :n].copy(), a[1, :n].copy()
        self.tail_pos = pos_first

    def layer(self, l):
        vo, d, w, x = self.vo, self.d, self.o.w, self.x
        n = x.shape[0]
        hd, heads, W = d.enc_head_dim, d.enc_heads, d.enc_window
        fr = vo.rope_freqs(self.pos0 + np.arange(n), hd, d.rope_theta)
        xn = vo.rms_norm(x, w.f32(w.enc(l, "attention_norm.weight")), d.enc_eps)
        q = vo.linear_bf16(xn, w.bf(w.enc(l, "attention.wq.weight")), w.f32(w.enc(l, "attention.wq.bias")))
        k = vo.linear_bf16(xn, w.bf(w.enc(l, "attention.wk.weight")))
        v = vo.linear_bf16(xn, w.bf(w.enc(l, "attention.wv.weight")), w.f32(w.enc(l, "attention.wv.bias")))
        q, k = vo.apply_rope(q, fr, heads, hd), vo.apply_rope(k, fr, heads, hd)
        self.own_k[l], self.own_v[l] = k, v
        if self.tail_k[l] is not None:
            kk, vv, off = np.concatenate([self.tail_k[l], k]), np.concatenate([self.tail_v[l], v]), self.tail_k[l].shape[0]
        else:
            kk, vv, off = k, v, 0
        a = vo.causal_attention(q, kk, vv, heads, heads, hd, 1.0 / np.sqrt(np.float32(hd)), W, off)
        x = x + vo.linear_bf16(a, w.bf(w.enc(l, "attention.wo.weight")), w.f32(w.enc(l, "attention.wo.bias")))
        xn = vo.rms_norm(x, w.f32(w.enc(l, "ffn_norm.weight")), d.enc_eps)
        g = vo.silu(vo.linear_bf16(xn, w.bf(w.enc(l, "feed_forward.w1.weight"))))
        u = vo.linear_bf16(xn, w.bf(w.enc(l, "feed_forward.w3.weight")))
        self.x = x + vo.linear_bf16(g * u, w.bf(w.enc(l, "feed_forward.w2.weight")), w.f32(w.enc(l, "feed_forward.w2.bias")))

    def kv_export(self, l, pos_first, n, buf):
        kk, vv = self.own_k[l], self.own_v[l]
        if self.tail_k[l] is not None:
            kk, vv = np.concatenate([self.tail_k[l], kk]), np.concatenate([self.tail_v[l], vv])
            base = self.tail_pos
        else:
            base = self.pos0
        a = buf.numpy()
        a[0, :n] = kk[pos_first - base:pos_first - base + n]
        a[1, :n] = vv[pos_first - base:pos_first - base + n]

    def end(self, buf):
        from oracle.vox_oracle import ENC_PFX
        e = self.vo.rms_norm(self.x, self.o.w.f32(f"{ENC_PFX}.transformer.norm.weight"), self.d.enc_eps)
        ad = self.o.adapter(e)
        buf.numpy()[:ad.shape[0]] = ad
        return ad.shape[0]

    def sync(self):
        raise AssertionError("encode_sharded must not wait on the engine: the wavefront is enqueue-only "
                             "(host-staged transports block inside their own recv / wait / Staging hooks)")


def _worker(rank, world, port, mdir, out_path, dst=0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from audio_util import synth_speech
    from oracle import vox_oracle as vo
    from voxtral_c_amd.multi_gpu import Staging, TorchComm, encode_sharded, padded_stream
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dims = vo.PRESETS["tiny"]
    eng = OracleShardEngine(mdir, dims)
    comm = TorchComm()
    audio = synth_speech(12.0, 41)
    padded, n_frames = padded_stream(audio)

    def staging(shape):
        return Staging(comm.empty(shape))

    comm.sync = eng.sync          # any host synchronisation requested by the orchestration itself is an error
    rows, counts = encode_sharded(eng, comm, padded, n_frames, staging, dst=dst)
    assert (rows is not None) == (rank == dst)
    # gloo blocks per message by construction; what is counted here is that the orchestration waited only where a
    # buffer is about to be reused (one wait per receive, one per send of two layers earlier + the final two)
    L = eng.n_layers
    expect = (L if rank > 0 else 0) + (L if rank + 1 < world else 0)
    assert comm.host_waits == expect, (comm.host_waits, expect)
    if rank == dst:
        np.save(out_path, rows.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,dst", [(2, 0), (3, 0), (2, 1)])
def test_sharded_encoder_equals_single_process(tmp_path, world, dst):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    from audio_util import synth_speech
    from oracle import vox_oracle as vo
    from voxtral_c_amd.multi_gpu import padded_stream
    mdir = model_dir("tiny")
    out = str(tmp_path / "rows.npy")
    mp.spawn(_worker, args=(world, _free_port(), mdir, out, dst), nprocs=world, join=True)
    got = np.load(out)
    # single-process reference: whole clip through the oracle's stream encoder
    dims = vo.PRESETS["tiny"]
    o = vo.Oracle(mdir, dims)
    audio = synth_speech(12.0, 41)
    padded, n_frames = padded_stream(audio)
    mel = vo.mel_frames(padded, n_frames)
    ref = o.stream_encode(mel)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert np.abs(got - ref).max() < 2e-4


def test_shard_plan_and_mel_span():
    from voxtral_c_amd.multi_gpu import mel_span, padded_stream, shard_plan
    plan = shard_plan(3392, 8)          # 30 s clip: 1696 positions = 424 tokens
    assert plan[0][0] == 0 and plan[-1][1] == 1696
    assert all(a % 4 == 0 and b % 4 == 0 and b > a for a, b in plan)
    assert all(plan[i][1] == plan[i + 1][0] for i in range(7))
    assert mel_span(0, 212) == (0, 424, 0)
    assert mel_span(212, 424) == (420, 848, 2)
    buf, n = padded_stream(np.zeros(480000, np.float32))
    assert n == 3392                     # 8 mel frames per adapter token, M = 424
    buf, n = padded_stream(np.zeros(176000, np.float32))
    assert n == 1496                     # jfk.wav: 1496 frames


def _pipelined_worker(rank, world, port, mdir, out_dir):
    """Round 4: pipelined delivery of the adapter rows (deliver_rows_pipelined).  The LAST rank is slowed down (it sleeps before
    every layer); rank 0 stamps the moment it is handed its first block - in the product that is where the decoder's prefill
    starts - and every rank stamps the end of its own shard."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import time
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from audio_util import synth_speech
    from oracle import vox_oracle as vo
    from voxtral_c_amd.multi_gpu import Staging, TorchComm, encode_sharded, padded_stream
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dims = vo.PRESETS["tiny"]
    eng = OracleShardEngine(mdir, dims)
    if rank == world - 1:
        plain_layer = eng.layer
        eng.layer = lambda l: (time.sleep(0.4), plain_layer(l))[1]
    comm = TorchComm()
    padded, n_frames = padded_stream(synth_speech(12.0, 41))
    blocks, stamps = [], {}

    def consume(r, rows):
        stamps.setdefault("first_block", time.time())
        blocks.append((r, rows.numpy().copy()))

    def staging(shape):
        return Staging(comm.empty(shape))

    rows, counts = encode_sharded(eng, comm, padded, n_frames, staging, dst=0, on_encoded=lambda: stamps.setdefault("shard_done", time.time()),
                                  consume=consume)
    assert rows is None
    all_stamps = [None] * world
    dist.all_gather_object(all_stamps, stamps)
    if rank == 0:
        assert [r for r, _ in blocks] == list(range(world)) and [b.shape[0] for _, b in blocks] == counts
        np.save(os.path.join(out_dir, "rows.npy"), np.concatenate([b for _, b in blocks]))
        # rank 0 had its first block - and, in the product, its decoder running - before the slowed-down last shard was finished
        assert all_stamps[0]["first_block"] < all_stamps[world - 1]["shard_done"] - 0.3, all_stamps
    dist.barrier()
    dist.destroy_process_group()


def test_rows_are_delivered_block_by_block_and_decoding_starts_before_the_last_shard_is_done(tmp_path):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    from audio_util import synth_speech
    from oracle import vox_oracle as vo
    from voxtral_c_amd.multi_gpu import padded_stream
    mdir = model_dir("tiny")
    mp.spawn(_pipelined_worker, args=(3, _free_port(), mdir, str(tmp_path)), nprocs=3, join=True)
    got = np.load(str(tmp_path / "rows.npy"))
    o = vo.Oracle(mdir, vo.PRESETS["tiny"])
    padded, n_frames = padded_stream(synth_speech(12.0, 41))
    ref = o.stream_encode(vo.mel_frames(padded, n_frames))
    assert got.shape == ref.shape and np.abs(got - ref).max() < 2e-4
