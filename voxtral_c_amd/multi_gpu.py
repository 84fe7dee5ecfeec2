"""Multi-GPU transcription: exact context-parallel encoder + single-stream decoder.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm).  The
reference has nothing distributed; this is the MI355X-native extension BASELINE.json's
north star asks for, designed for a node whose 8 GPUs are point-to-point xGMI peers:

  * The padded audio of the whole job is split into N contiguous ranges of encoder positions
    (token aligned).  Rank r computes mel + conv stem for its own range (with a 4-frame mel
    halo, recomputed locally — 0.3 % redundant work, no communication).
  * The 32 causal window-750 encoder layers are walked in lock step as a wavefront: right
    after finishing layer l a rank sends the layer-l K/V of its last 749 positions to its
    right neighbour (one point-to-point message of 2*749*2048*4 B = 12.3 MB per layer and
    neighbour pair; xGMI is point-to-point, so neighbour send/recv uses exactly one link
    and no ring collective), which imports them into its position-indexed KV ring before
    running its own layer l.  Results are bit-for-bit those of one GPU encoding everything
    (the arithmetic per position is identical) — unlike overlapping chunks, whose leading
    rows would see a truncated context (the dependency cone is 32*749 positions).
  * Adapter rows (12 KB per 80 ms token) are gathered on rank 0 (dist.gather), appended to
    its device adapter buffer and decoded there: the decoder is strictly sequential, so
    decode tokens/s does not scale with N — only the encoder share of the RTF does.

The orchestration below is written against a small ShardEngine interface so that the CPU
test-suite can run it under gloo with the numpy oracle standing in for the GPU
(tests/test_multi_gpu_cpu.py); HipShardEngine is the real thing.
"""
import ctypes as C
import json
import os
import time

import numpy as np

SAMPLES_PER_TOKEN = 1280
LEFT_PAD_TOKENS = 32


# ---------------------------------------------------------------------------------------
# sharding math (pure functions, unit-tested on CPU)
# ---------------------------------------------------------------------------------------
def padded_stream(samples, delay_tokens=6):
    """The sample stream the reference's offline path ends up windowing (voxtral.c:1203,
    1588-1606; voxtral_audio.c:544-555,584-633): zeros(200 + 32 tokens) | audio | zeros(align +
    (delay+1+10) tokens) | 200-sample reflection; frames = all 400-windows at hop 160, minus one."""
    s = np.asarray(samples, np.float32)
    n = len(s)
    align = (SAMPLES_PER_TOKEN - n % SAMPLES_PER_TOKEN) % SAMPLES_PER_TOKEN
    right = align + ((delay_tokens + 1) + 10) * SAMPLES_PER_TOKEN
    buf = np.concatenate([np.zeros(200 + LEFT_PAD_TOKENS * SAMPLES_PER_TOKEN, np.float32), s, np.zeros(right, np.float32)])
    buf = np.concatenate([buf, buf[len(buf) - 2 - np.arange(200)]])
    n_frames = (len(buf) - 400) // 160 + 1 - 1
    return buf, n_frames


def shard_plan(n_frames, world, min_rows=0):
    """Split the encoder positions (= n_frames // 2, grouped in tokens of 4) into `world`
    contiguous token-aligned ranges. Returns a list of (pos0, pos1)."""
    n_pos = n_frames // 2
    n_tok = n_pos // 4
    base, rem = divmod(n_tok, world)
    out, t = [], 0
    for r in range(world):
        cnt = base + (1 if r < rem else 0)
        out.append((t * 4, (t + cnt) * 4))
        t += cnt
    return out


def mel_span(pos0, pos1):
    """Mel frames rank needs and how many leading conv rows to drop: rows p in [pos0,pos1) need
    c0[2p-1..2p+1] and those need mel[2p-3..2p+1]; feeding from frame 2*pos0-4 through a
    zero-history conv stem contaminates exactly the first two rows."""
    if pos0 == 0:
        return 0, 2 * pos1, 0
    return 2 * pos0 - 4, 2 * pos1, 2


# ---------------------------------------------------------------------------------------
# engine adapters
# ---------------------------------------------------------------------------------------
class HipShardEngine:
    """vox_hip_shard_* on one GPU (device buffers are raw pointers owned by the engine)."""

    def __init__(self, model):
        import voxtral_c_amd as v
        self.v, self.m, self.e = v, model, model.engine
        h = v.hip
        h.vox_hip_shard_begin.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        h.vox_hip_shard_layer.argtypes = [C.c_void_p, C.c_int]
        h.vox_hip_shard_kv_export.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        h.vox_hip_shard_kv_import.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        h.vox_hip_shard_end.argtypes = [C.c_void_p, C.c_void_p]
        h.vox_hip_adapter_append_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        h.vox_hip_reset_decoder_kv.argtypes = [C.c_void_p]
        self.d = model.dims
        self.kv_dim = self.d.enc_heads * self.d.enc_head_dim
        self.n_layers = self.d.enc_layers
        self.window = self.d.enc_window
        self.dec_dim = self.d.dec_dim

    def reset(self):
        self.v.hip.vox_hip_reset_encoder(self.e)
        self.v.hip.vox_hip_reset_decoder(self.e)

    def queue_mel(self, padded, frame0, n_frames):
        seg = np.ascontiguousarray(padded[frame0 * 160:(frame0 + n_frames - 1) * 160 + 400], np.float32)
        rc = self.v.hip.vox_hip_mel_frames(self.e, seg.ctypes.data_as(self.v.f32p), n_frames, None, 1)
        assert rc == 0

    def begin(self, n_mel, discard, pos0):
        n = self.v.hip.vox_hip_shard_begin(self.e, n_mel, discard, pos0)
        assert n > 0, self.v.hip.vox_hip_last_error()
        return n

    def layer(self, l):
        assert self.v.hip.vox_hip_shard_layer(self.e, l) == 0

    def kv_export(self, l, pos_first, n, dev_ptr):
        assert self.v.hip.vox_hip_shard_kv_export(self.e, l, pos_first, n, dev_ptr) == 0

    def kv_import(self, l, pos_first, n, dev_ptr):
        assert self.v.hip.vox_hip_shard_kv_import(self.e, l, pos_first, n, dev_ptr) == 0

    def end(self, dev_ptr):
        m = self.v.hip.vox_hip_shard_end(self.e, dev_ptr)
        assert m >= 0, self.v.hip.vox_hip_last_error()
        return m

    def sync(self):
        self.v.hip.vox_hip_sync(self.e)


class TorchComm:
    """torch.distributed plumbing. Tensors live on the GPU for RCCL ("nccl") and on the host
    for gloo (used to exercise the 2-rank path on a single GPU / on CPU)."""

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.on_gpu = dist.get_backend() == "nccl"
        self.device = device if self.on_gpu else "cpu"

    def empty(self, shape):
        return self.torch.empty(shape, dtype=self.torch.float32, device=self.device)

    def send(self, t, dst):
        self.dist.send(t, dst)

    def recv(self, t, src):
        self.dist.recv(t, src)

    def gather_rows(self, t, counts, dst=0):
        """Variable-length gather of [m_r, D] row blocks to rank `dst` (padded to the max count)."""
        mx = max(counts)
        pad = self.empty((mx, t.shape[1]))
        pad[:t.shape[0]] = t
        lst = [self.empty((mx, t.shape[1])) for _ in range(self.world)] if self.rank == dst else None
        self.dist.gather(pad, lst, dst=dst)
        if self.rank != dst:
            return None
        return self.torch.cat([lst[r][:counts[r]] for r in range(self.world)])

    def barrier(self):
        self.dist.barrier()

    def sync(self):
        if self.on_gpu:
            self.torch.cuda.synchronize()


class Staging:
    """A buffer the engine writes/reads (by pointer) and the communicator sends/receives (as a
    torch tensor).  RCCL: one GPU tensor, pointer = data_ptr.  gloo next to the HIP engine: a
    host tensor mirrored by a device buffer (copied around each transfer).  CPU oracle engine:
    the tensor itself."""

    def __init__(self, tensor, ptr=None, to_host=None, to_dev=None):
        self.tensor, self._ptr, self._to_host, self._to_dev = tensor, ptr, to_host, to_dev

    def ptr(self):
        return self.tensor if self._ptr is None else self._ptr

    def after_engine_write(self):
        if self._to_host:
            self._to_host()

    def before_engine_read(self):
        if self._to_dev:
            self._to_dev()


def encode_sharded(eng, comm, padded, n_frames, staging, dst=0):
    """Wavefront context-parallel encode. `staging(shape)` returns a Staging buffer.
    Returns (gathered adapter rows on rank `dst` | None, per-rank row counts)."""
    rank, world = comm.rank, comm.world
    plan = shard_plan(n_frames, world)
    pos0, pos1 = plan[rank]
    f0, f1, discard = mel_span(pos0, pos1)
    eng.queue_mel(padded, f0, f1 - f0)
    n = eng.begin(f1 - f0, discard, pos0)
    assert n == pos1 - pos0, (n, pos0, pos1)
    tail = min(eng.window - 1, pos0)                       # positions we need from the left
    send_tail = min(eng.window - 1, pos1) if rank + 1 < world else 0
    # Two receive buffers, alternating by layer: kv_import only ENQUEUES its copies on the engine stream, so the buffer of
    # layer l - 1 may still be read while layer l's tail arrives.  A buffer is reused two layers later; by then the engine
    # stream has been synchronised by kv_export - except on a rank that never exports (the last one), which syncs itself.
    s_ins = [staging((2, max(tail, 1), eng.kv_dim)) for _ in range(2)]
    s_out = staging((2, max(send_tail, 1), eng.kv_dim))
    for l in range(eng.n_layers):
        if rank > 0 and tail > 0:
            s_in = s_ins[l & 1]
            if send_tail == 0 and l >= 2:
                eng.sync()
            comm.recv(s_in.tensor, rank - 1)
            comm.sync()
            s_in.before_engine_read()
            eng.kv_import(l, pos0 - tail, tail, s_in.ptr())
        eng.layer(l)
        if send_tail > 0:
            comm.sync()            # the previous layer's send has left the staging buffer
            eng.kv_export(l, pos1 - send_tail, send_tail, s_out.ptr())    # synchronises the engine stream
            s_out.after_engine_write()
            comm.send(s_out.tensor, rank + 1)
    s_ad = staging(((pos1 - pos0) // 4, eng.dec_dim))
    m = eng.end(s_ad.ptr())
    s_ad.after_engine_write()
    assert m == (pos1 - pos0) // 4
    counts = [(b - a) // 4 for a, b in plan]
    return comm.gather_rows(s_ad.tensor, counts, dst), counts


# ---------------------------------------------------------------------------------------
# bench driver for N > 1 (called by bench.py)
# ---------------------------------------------------------------------------------------
def run_distributed_bench(args, rank, world, local_rank, mdir, dims):
    import torch
    import torch.distributed as dist
    import voxtral_c_amd as v
    from audio_util import synth_speech

    backend = os.environ.get("VOX_DIST_BACKEND", "nccl")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # one node, no fabric: keep RCCL's bootstrap off interface / InfiniBand probing (seen to stall
    # communicator creation for ~2 minutes on boxes without a network)
    os.environ.setdefault("NCCL_IB_DISABLE", "1")
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    share = os.environ.get("VOX_SHARE_GPU") == "1"          # several ranks on one GPU (gloo only; tests)
    dev = 0 if share else local_rank
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group(backend=backend, device_id=torch.device(f"cuda:{dev}"))
    else:
        dist.init_process_group(backend=backend)
    comm = TorchComm(device=f"cuda:{dev}")
    win = {} if args.preset != "tiny" else dict(enc_window=48, dec_window=64)
    model = v.Model(mdir, device=dev, **win)
    session = DistributedSession(model, comm)
    # one clip per GPU ("owner"); every clip's encoder is sharded over all GPUs, every GPU decodes
    # its own clip (VOX_DIST_MODE=single: one long clip, decoder on rank 0 only — BASELINE config 4)
    single = os.environ.get("VOX_DIST_MODE") == "single"
    if single:
        audio = synth_speech(args.seconds * world, 1234)
    else:
        audios = [synth_speech(args.seconds, 1234 + r) for r in range(world)]

    def one_pass():
        toks = session.transcribe(audio) if single else session.transcribe_many(audios)
        comm.barrier()
        return toks

    for _ in range(args.warmup):
        one_pass()
    comm.barrier(); torch.cuda.synchronize()
    t0 = time.time()
    toks = None
    for _ in range(args.steps):
        toks = one_pass()
    comm.barrier(); torch.cuda.synchronize()
    wall = time.time() - t0
    dev_t = comm.device if comm.on_gpu else "cpu"
    tmax = torch.tensor([wall], dtype=torch.float64, device=dev_t)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    wall = float(tmax.item())
    t = model.timing()                       # of the last pass on this rank
    agg = torch.tensor([float(t["decode_steps"]), float(len(toks)) if toks is not None else 0.0], dtype=torch.float64, device=dev_t)
    dist.all_reduce(agg, op=dist.ReduceOp.SUM)
    dmax = torch.tensor([float(t["decode_ms"])], dtype=torch.float64, device=dev_t)
    dist.all_reduce(dmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        audio_s = args.seconds * world
        dec_ms = float(dmax.item())
        out = {
            "metric": "real-time-factor + decode tokens/sec, Voxtral-4B bf16, 30s audio",
            "value": round(wall / args.steps / audio_s, 5), "unit": "wall s / audio s (RTF)", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(wall * 1e3 / args.steps, 2),
            "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 weights, f32 activations/accumulate", "data": "synthetic",
            "decode_tok_s": round(float(agg[0].item()) / (dec_ms * 1e-3), 1) if dec_ms > 0 else 0.0,
            "decode_tok_s_per_stream": round(t["decode_steps"] / (t["decode_ms"] * 1e-3), 1) if t["decode_ms"] > 0 else 0.0,
            "decoder_steps_per_pass": int(agg[1].item()),
            "config": {"workload": (f"Voxtral-4B ({args.preset} synthetic checkpoint), one {audio_s:g} s clip: encoder positions sharded over "
                                    f"{world} GPUs (wavefront K/V halo over xGMI), adapter rows gathered to rank 0, single-stream greedy "
                                    "decode on rank 0") if single else
                                   (f"Voxtral-4B ({args.preset} synthetic checkpoint), {world} clips of {args.seconds:g} s (one per GPU): each clip's "
                                    f"encoder positions are sharded over all {world} GPUs (wavefront K/V halo over xGMI, RCCL gather of the "
                                    "adapter rows to the clip's GPU), then every GPU runs the single-stream greedy decoder of its own clip"),
                       "audio_seconds": audio_s, "parallelism": f"cp{world} encoder / " + ("1 decoder" if single else f"{world} decoders"),
                       "backend": backend},
        }
        try:        # same live roofline measurement as the 1-GPU line (rank 0's engine)
            from bench import roofline_block
            out["roofline"] = roofline_block(v, model, model.dims, float(len(toks)) if toks is not None else 380.0)
        except Exception as ex:
            out["roofline"] = {"error": str(ex)}
        print(json.dumps(out), flush=True)
    model.close()
    dist.destroy_process_group()


class DistributedSession:
    """One rank's view of a multi-GPU transcription (used by bench.py and the tests)."""

    def __init__(self, model, comm):
        import voxtral_c_amd as v
        self.v, self.model, self.comm = v, model, comm
        self.eng = HipShardEngine(model)
        h = v.hip
        h.vox_hip_device_alloc.restype = C.c_void_p
        h.vox_hip_device_alloc.argtypes = [C.c_void_p, C.c_size_t]
        h.vox_hip_device_free.argtypes = [C.c_void_p, C.c_void_p]
        h.vox_hip_memcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self._dev_bufs = []

    def staging(self, shape):
        comm, h, eng = self.comm, self.v.hip, self.model.engine
        t = comm.empty(shape)
        if comm.on_gpu:
            return Staging(t, C.c_void_p(t.data_ptr()), to_host=comm.sync, to_dev=None)
        nbytes = t.numel() * 4
        dev = h.vox_hip_device_alloc(eng, nbytes)
        self._dev_bufs.append(dev)
        host = C.c_void_p(t.data_ptr())
        return Staging(t, C.c_void_p(dev),
                       to_host=lambda: h.vox_hip_memcpy(eng, host, C.c_void_p(dev), nbytes, 1),
                       to_dev=lambda: h.vox_hip_memcpy(eng, C.c_void_p(dev), host, nbytes, 0))

    def _free_staging(self):
        for d in self._dev_bufs:
            self.v.hip.vox_hip_device_free(self.model.engine, C.c_void_p(d))
        self._dev_bufs = []

    def transcribe_many(self, audios, delay_tokens=6):
        """One clip per rank ("owner").  Every clip's encoder is sharded over ALL ranks (wavefront K/V
        halo, adapter rows gathered to the clip's owner); afterwards every rank decodes its own
        clip, so the strictly sequential decoders of the N streams run side by side.  Returns
        this rank's token ids."""
        assert len(audios) == self.comm.world
        rows_mine = None
        for owner, audio in enumerate(audios):
            padded, n_frames = padded_stream(audio, delay_tokens)
            self.eng.reset()
            rows, _ = encode_sharded(self.eng, self.comm, padded, n_frames, self.staging, dst=owner)
            if self.comm.on_gpu and rows is not None:
                rows = rows.clone()              # outlives the staging buffers of the later clips
            self._free_staging()
            if owner == self.comm.rank:
                rows_mine = rows
        self.eng.reset()
        return self._decode_rows(rows_mine, delay_tokens)

    def transcribe(self, audio, delay_tokens=6):
        """Sharded encode on all ranks, greedy decode on rank 0. Returns token ids on rank 0."""
        padded, n_frames = padded_stream(audio, delay_tokens)
        self.eng.reset()
        rows, counts = encode_sharded(self.eng, self.comm, padded, n_frames, self.staging)
        self._free_staging()
        return self._decode_rows(rows, delay_tokens) if self.comm.rank == 0 else None

    def _decode_rows(self, rows, delay_tokens):
        v, h, comm, model = self.v, self.v.hip, self.comm, self.model
        prompt_len = 1 + LEFT_PAD_TOKENS + delay_tokens
        toks = None
        if rows is not None:
            total = int(rows.shape[0])
            if comm.on_gpu:
                comm.sync()
                h.vox_hip_adapter_append_dev(model.engine, C.c_void_p(rows.data_ptr()), total)
            else:
                arr = np.ascontiguousarray(rows.numpy())
                h.vox_hip_adapter_append(model.engine, arr.ctypes.data_as(v.f32p), total)
            h.vox_hip_reset_decoder_kv(model.engine)
            first = h.vox_hip_decoder_prefill_stream(model.engine, 0, prompt_len, 1, 32, None)
            n_steps = total - prompt_len
            out = np.zeros(max(n_steps, 1), np.int32)
            got = 0
            if n_steps > 0 and first != 2:
                got = h.vox_hip_decoder_run(model.engine, prompt_len, n_steps, first, 2, out.ctypes.data_as(v.i32p), None)
            toks = np.concatenate([[first], out[:got]]).astype(np.int32)
        return toks
