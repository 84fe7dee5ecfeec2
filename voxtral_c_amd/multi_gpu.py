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
    """vox_hip_shard_* on one GPU (device buffers are raw pointers owned by the engine).

    stream_ordered=True (RCCL): only the *_async entry points are used - every call enqueues on the engine's HIP
    stream and returns; the communicator issues its sends / receives on that same stream (TorchComm.stream), so the
    wavefront needs no host wait between begin() and end().  stream_ordered=False (host-staged transports: gloo on a
    shared GPU in the tests): the synchronous variants."""

    def __init__(self, model, stream_ordered=False):
        import voxtral_c_amd as v
        self.v, self.m, self.e = v, model, model.engine
        self.stream_ordered = stream_ordered
        h = v.hip
        h.vox_hip_shard_begin.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        h.vox_hip_shard_layer.argtypes = [C.c_void_p, C.c_int]
        for fn in (h.vox_hip_shard_kv_export, h.vox_hip_shard_kv_export_async, h.vox_hip_shard_kv_import):
            fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        h.vox_hip_shard_end.argtypes = [C.c_void_p, C.c_void_p]
        h.vox_hip_shard_end_async.argtypes = [C.c_void_p, C.c_void_p]
        h.vox_hip_adapter_append_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        h.vox_hip_adapter_append_dev_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        h.vox_hip_reset_decoder_kv.argtypes = [C.c_void_p]
        h.vox_hip_reset_encoder_async.argtypes = [C.c_void_p]
        h.vox_hip_host_syncs.restype = C.c_ulonglong
        h.vox_hip_host_syncs.argtypes = [C.c_void_p]
        self.d = model.dims
        self.kv_dim = self.d.enc_heads * self.d.enc_head_dim
        self.n_layers = self.d.enc_layers
        self.window = self.d.enc_window
        self.dec_dim = self.d.dec_dim
        self.wavefront_syncs = 0          # host waits on the engine stream between begin() and end(), summed over shards
        self._syncs0 = 0

    def host_syncs(self):
        return int(self.v.hip.vox_hip_host_syncs(self.e))

    def reset(self):
        if self.stream_ordered:
            self.v.hip.vox_hip_reset_encoder_async(self.e)
        else:
            self.v.hip.vox_hip_reset_encoder(self.e)
        self.v.hip.vox_hip_reset_decoder(self.e)

    def reset_encoder(self):
        (self.v.hip.vox_hip_reset_encoder_async if self.stream_ordered else self.v.hip.vox_hip_reset_encoder)(self.e)

    def queue_mel(self, padded, frame0, n_frames):
        seg = np.ascontiguousarray(padded[frame0 * 160:(frame0 + n_frames - 1) * 160 + 400], np.float32)
        rc = self.v.hip.vox_hip_mel_frames(self.e, seg.ctypes.data_as(self.v.f32p), n_frames, None, 1)
        assert rc == 0

    def begin(self, n_mel, discard, pos0):
        n = self.v.hip.vox_hip_shard_begin(self.e, n_mel, discard, pos0)
        assert n > 0, self.v.hip.vox_hip_last_error()
        self._syncs0 = self.host_syncs()
        return n

    def layer(self, l):
        assert self.v.hip.vox_hip_shard_layer(self.e, l) == 0

    def kv_export(self, l, pos_first, n, dev_ptr):
        fn = self.v.hip.vox_hip_shard_kv_export_async if self.stream_ordered else self.v.hip.vox_hip_shard_kv_export
        assert fn(self.e, l, pos_first, n, dev_ptr) == 0

    def kv_import(self, l, pos_first, n, dev_ptr):
        assert self.v.hip.vox_hip_shard_kv_import(self.e, l, pos_first, n, dev_ptr) == 0

    def end(self, dev_ptr):
        if self.stream_ordered:
            m = self.v.hip.vox_hip_shard_end_async(self.e, dev_ptr)
            self.wavefront_syncs += self.host_syncs() - self._syncs0
        else:
            m = self.v.hip.vox_hip_shard_end(self.e, dev_ptr)
        assert m >= 0, self.v.hip.vox_hip_last_error()
        return m

    def sync(self):
        self.v.hip.vox_hip_sync(self.e)


class TorchComm:
    """torch.distributed plumbing. Tensors live on the GPU for RCCL ("nccl") and on the host
    for gloo (used to exercise the 2-rank path on a single GPU / on CPU).

    RCCL: `stream` is the engine's own HIP stream wrapped as a torch ExternalStream.  ProcessGroupNCCL orders every
    operation behind the CURRENT stream's tail when it is issued and makes the current stream wait for its completion
    on wait() - both are stream waits, not host waits - so with the engine stream current, a send sits behind the
    kernels that produced its buffer and the engine's next kernels sit behind a receive, with the host only enqueueing
    (ordered() is the context manager that makes the engine stream current)."""

    def __init__(self, device=None, engine_stream=None):
        import contextlib
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.on_gpu = dist.get_backend() == "nccl"
        self.device = device if self.on_gpu else "cpu"
        self.stream = None
        if self.on_gpu and engine_stream:
            self.stream = torch.cuda.ExternalStream(int(engine_stream), device=torch.device(device))
        self._null = contextlib.nullcontext
        self.host_waits = 0               # blocking waits issued by this communicator (gloo: every operation)

    def ordered(self):
        return self.torch.cuda.stream(self.stream) if self.stream is not None else self._null()

    def empty(self, shape):
        return self.torch.empty(shape, dtype=self.torch.float32, device=self.device)

    def isend(self, t, dst):
        return self.dist.isend(t, dst)

    def recv(self, t, src):
        """RCCL: enqueued; the current (engine) stream waits for the data.  gloo: blocks until it has arrived."""
        w = self.dist.irecv(t, src)
        self.wait(w)

    def wait(self, work):
        if work is None:
            return
        if not self.on_gpu:
            self.host_waits += 1
        work.wait()                        # nccl: the current stream waits (no host block); gloo: host block

    def loopback(self, t_out, t_in):
        """Send t_out to ourselves into t_in (one grouped RCCL send + recv): exercises the stream-ordered point-to-point
        path with the only peer a 1-GPU box has."""
        ops = [self.dist.P2POp(self.dist.isend, t_out, self.rank), self.dist.P2POp(self.dist.irecv, t_in, self.rank)]
        for w in self.dist.batch_isend_irecv(ops):
            self.wait(w)

    def gather_rows(self, t, counts, dst=0):
        """Variable-length gather of [m_r, D] row blocks to rank `dst` (padded to the max count)."""
        if self.world == 1:
            return t
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
    torch tensor).  RCCL: one GPU tensor, pointer = data_ptr, no hooks (everything is stream-ordered).  gloo next to
    the HIP engine: a host tensor mirrored by a device buffer (copied - synchronously - around each transfer).  CPU
    oracle engine: the tensor itself."""

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


def deliver_rows_pipelined(comm, local_rows, counts, dst, consume, empty=None):
    """Round 4 ("GPU 0 starts decoding while later shards still encode"): every rank's adapter rows go to rank `dst`
    point to point, in rank order, and `consume(r, rows_r)` runs on `dst` as each block arrives - its own block without any
    communication.  With dst = 0 (one clip, decoder on rank 0) the decoder is already running on the first shard's ~T/N tokens
    while ranks 1 .. N-1 are still in their wavefront; their sends wait (on their own streams) until rank 0 posts the receive.
    No collective: N - 1 point-to-point messages of 12 KB per token over one xGMI link each."""
    rank, world = comm.rank, comm.world
    if rank == dst:
        for r in range(world):
            if r == rank:
                consume(r, local_rows)
            else:
                t = (empty or comm.empty)((counts[r], local_rows.shape[1]))
                comm.recv(t, r)
                consume(r, t)
    else:
        comm.wait(comm.isend(local_rows, dst))


def encode_sharded(eng, comm, padded, n_frames, staging, dst=0, self_loop=False, on_encoded=None, consume=None):
    """Wavefront context-parallel encode. `staging(shape)` returns a Staging buffer.
    Returns (gathered adapter rows on rank `dst` | None, per-rank row counts); with `consume` the rows are not gathered by a
    collective but delivered block by block (deliver_rows_pipelined) and the first element of the result is None.

    This function never waits on the host itself: with a stream-ordered engine + RCCL every call below only enqueues
    (tests assert vox_hip_host_syncs does not move between begin() and end()); host-staged transports block inside
    their own recv / wait / Staging hooks.  Buffers alternate by layer parity: a receive buffer is overwritten two
    layers later, behind the import copies that read it (same stream); a send buffer is rewritten only after wait()
    on the send that read it."""
    rank, world = comm.rank, comm.world
    plan = shard_plan(n_frames, world)
    pos0, pos1 = plan[rank]
    f0, f1, discard = mel_span(pos0, pos1)
    eng.queue_mel(padded, f0, f1 - f0)
    n = eng.begin(f1 - f0, discard, pos0)
    assert n == pos1 - pos0, (n, pos0, pos1)
    tail = min(eng.window - 1, pos0) if rank > 0 else 0     # positions we need from the left
    send_tail = min(eng.window - 1, pos1) if rank + 1 < world else 0
    loop_tail = min(eng.window - 1, pos1) if (self_loop and world == 1) else 0
    s_ins = [staging((2, max(tail, loop_tail, 1), eng.kv_dim)) for _ in range(2)]
    s_outs = [staging((2, max(send_tail, loop_tail, 1), eng.kv_dim)) for _ in range(2)]
    sends = [None, None]
    for l in range(eng.n_layers):
        if tail > 0:
            s_in = s_ins[l & 1]
            comm.recv(s_in.tensor, rank - 1)
            s_in.before_engine_read()
            eng.kv_import(l, pos0 - tail, tail, s_in.ptr())
        eng.layer(l)
        if send_tail > 0:
            s_out = s_outs[l & 1]
            comm.wait(sends[l & 1])            # the send of layer l - 2 has read this buffer
            eng.kv_export(l, pos1 - send_tail, send_tail, s_out.ptr())
            s_out.after_engine_write()
            sends[l & 1] = comm.isend(s_out.tensor, rank + 1)
        if loop_tail > 0:                      # 1-GPU smoke of the RCCL point-to-point path: export, send to self, import
            eng.kv_export(l, pos1 - loop_tail, loop_tail, s_outs[l & 1].ptr())
            comm.loopback(s_outs[l & 1].tensor, s_ins[l & 1].tensor)
            eng.kv_import(l, pos1 - loop_tail, loop_tail, s_ins[l & 1].ptr())
    for w in sends:
        comm.wait(w)
    s_ad = staging(((pos1 - pos0) // 4, eng.dec_dim))
    m = eng.end(s_ad.ptr())
    s_ad.after_engine_write()
    assert m == (pos1 - pos0) // 4
    counts = [(b - a) // 4 for a, b in plan]
    if on_encoded:
        on_encoded()
    if consume is not None:
        deliver_rows_pipelined(comm, s_ad.tensor, counts, dst, consume)
        return None, counts
    return comm.gather_rows(s_ad.tensor, counts, dst), counts


class _IncrementalDecoder:
    """Greedy decoder fed with adapter rows block by block (deliver_rows_pipelined): prefill as soon as the prompt's rows are there,
    then one vox_hip_decoder_run per block over the rows that have arrived.  Same calls, same order of steps as decoding after a
    complete gather, so the ids are those of the single-GPU run."""

    def __init__(self, session, delay_tokens):
        self.s, self.prompt_len = session, 1 + LEFT_PAD_TOKENS + delay_tokens
        self.total, self.pos, self.first, self.out, self.stopped = 0, 0, None, [], False
        session.v.hip.vox_hip_reset_decoder_kv(session.model.engine)

    def append(self, rows):
        s = self.s
        v, h, comm, model = s.v, s.v.hip, s.comm, s.model
        n = int(rows.shape[0])
        if n == 0:
            return
        if comm.on_gpu:
            s._keep.append(rows)
            assert h.vox_hip_adapter_append_dev_async(model.engine, C.c_void_p(rows.data_ptr()), n) == 0
        else:
            arr = np.ascontiguousarray(rows.numpy())
            h.vox_hip_adapter_append(model.engine, arr.ctypes.data_as(v.f32p), n)
        self.total += n
        if self.stopped:
            return
        if self.first is None:
            if self.total < self.prompt_len:
                return
            self.first = h.vox_hip_decoder_prefill_stream(model.engine, 0, self.prompt_len, 1, 32, None)
            self.pos = self.prompt_len
            self.out.append(np.array([self.first], np.int32))
            if self.first == 2:
                self.stopped = True
                return
        n_steps = self.total - self.pos
        if n_steps > 0:
            out = np.zeros(n_steps, np.int32)
            prev = int(self.out[-1][-1])
            got = h.vox_hip_decoder_run(model.engine, self.pos, n_steps, prev, 2, out.ctypes.data_as(v.i32p), None)
            assert got >= 0, h.vox_hip_last_error()
            self.out.append(out[:got])
            self.pos += got
            if got < n_steps:
                self.stopped = True        # eos

    def tokens(self):
        return np.concatenate(self.out).astype(np.int32) if self.out else np.zeros(0, np.int32)


# ---------------------------------------------------------------------------------------
# one rank's view of a multi-GPU transcription (bench.py, tests)
# ---------------------------------------------------------------------------------------
class DistributedSession:
    def __init__(self, model, comm):
        import voxtral_c_amd as v
        self.v, self.model, self.comm = v, model, comm
        self.eng = HipShardEngine(model, stream_ordered=comm.on_gpu)
        h = v.hip
        h.vox_hip_device_alloc.restype = C.c_void_p
        h.vox_hip_device_alloc.argtypes = [C.c_void_p, C.c_size_t]
        h.vox_hip_device_free.argtypes = [C.c_void_p, C.c_void_p]
        h.vox_hip_memcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self._dev_bufs = []
        self._keep = []                    # RCCL staging tensors of the pass in flight (freed after the pass's final sync)
        self.self_loop = os.environ.get("VOX_DIST_SELF_LOOP") == "1"
        self.phase_ms = {}                 # of the last pass: encode (mel + wavefront), gather, prefill, decode
        self._events = []

    # ---- staging buffers ---------------------------------------------------------------------------------------
    def staging(self, shape):
        comm, h, eng = self.comm, self.v.hip, self.model.engine
        t = comm.empty(shape)
        if comm.on_gpu:
            self._keep.append(t)
            return Staging(t, C.c_void_p(t.data_ptr()))
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
        self._keep = []

    # ---- phase timing: events on the engine stream (RCCL) / host clock (host-staged transports) -------------------
    def _mark(self, name):
        if self.comm.on_gpu and self.comm.stream is not None:
            ev = self.comm.torch.cuda.Event(enable_timing=True)
            ev.record(self.comm.stream)
            self._events.append((name, ev))
        else:
            self._events.append((name, time.time()))

    def _phases_done(self):
        acc = {}
        for (n0, e0), (n1, e1) in zip(self._events, self._events[1:]):
            if n1 == "start":
                continue
            dt = e0.elapsed_time(e1) if self.comm.on_gpu and self.comm.stream is not None else (e1 - e0) * 1e3
            acc[n1] = acc.get(n1, 0.0) + dt
        self._events = []
        return acc

    # ---- transcription ------------------------------------------------------------------------------------------------
    def transcribe_many(self, audios, delay_tokens=6):
        """One clip per rank ("owner").  Every clip's encoder is sharded over ALL ranks (wavefront K/V
        halo, adapter rows gathered to the clip's owner); afterwards every rank decodes its own
        clip, so the strictly sequential decoders of the N streams run side by side.  Returns
        this rank's token ids.  With RCCL the N wavefronts are enqueued back to back (rank 0 starts clip c + 1 while
        the later ranks still work on clip c)."""
        assert len(audios) == self.comm.world
        rows_mine = None
        self.eng.wavefront_syncs = 0
        self.eng.reset()
        with self.comm.ordered():
            for owner, audio in enumerate(audios):
                padded, n_frames = padded_stream(audio, delay_tokens)
                self.eng.reset_encoder()
                self._mark("start")
                rows, _ = encode_sharded(self.eng, self.comm, padded, n_frames, self.staging, dst=owner, self_loop=self.self_loop,
                                         on_encoded=lambda: self._mark("encode"))
                if owner == self.comm.rank:
                    rows_mine = rows
                self._mark("gather")
            toks = self._decode_rows(rows_mine, delay_tokens)
        self._finish_pass()
        return toks

    def transcribe(self, audio, delay_tokens=6):
        """Sharded encode on all ranks, greedy decode on rank 0. Returns token ids on rank 0."""
        padded, n_frames = padded_stream(audio, delay_tokens)
        self.eng.wavefront_syncs = 0
        self.eng.reset()
        pipelined = os.environ.get("VOX_DIST_NO_PIPELINE") != "1"      # A/B: the round-3 collective gather, decode after it
        with self.comm.ordered():
            self._mark("start")
            if pipelined:
                self._pipelined_pass = True
                self._t_before = self.model.timing()
                dec = _IncrementalDecoder(self, delay_tokens) if self.comm.rank == 0 else None
                encode_sharded(self.eng, self.comm, padded, n_frames, self.staging, self_loop=self.self_loop,
                               on_encoded=lambda: self._mark("encode"),
                               consume=(lambda r, rows: dec.append(rows)) if dec else (lambda r, rows: None))
                self._mark("gather")
                toks = dec.tokens() if dec else None
            else:
                rows, counts = encode_sharded(self.eng, self.comm, padded, n_frames, self.staging, self_loop=self.self_loop,
                                              on_encoded=lambda: self._mark("encode"))
                self._mark("gather")
                toks = self._decode_rows(rows, delay_tokens) if self.comm.rank == 0 else None
        self._finish_pass()
        return toks

    def _finish_pass(self):
        self.eng.sync()
        self.comm.sync()
        ph = self._phases_done()
        self.phase_ms = {"encode": ph.get("encode", 0.0), "gather": ph.get("gather", 0.0)}
        if getattr(self, "_pipelined_pass", False):
            # the rows were delivered block by block and each block was decoded as it arrived: the interval called "gather" holds the
            # decoder's work of rank 0 as well - report what is left of it beside the engine's own prefill + decode timers
            t, t0 = self.model.timing(), self._t_before
            self.phase_ms["gather_and_decode"] = self.phase_ms["gather"]
            self.phase_ms["gather"] = max(0.0, self.phase_ms["gather"] - (t["prefill_ms"] - t0["prefill_ms"]) - (t["decode_ms"] - t0["decode_ms"]))
            self._pipelined_pass = False
        self._free_staging()

    def _decode_rows(self, rows, delay_tokens):
        v, h, comm, model = self.v, self.v.hip, self.comm, self.model
        prompt_len = 1 + LEFT_PAD_TOKENS + delay_tokens
        toks = None
        if rows is not None:
            total = int(rows.shape[0])
            if comm.on_gpu:
                self._keep.append(rows)
                assert h.vox_hip_adapter_append_dev_async(model.engine, C.c_void_p(rows.data_ptr()), total) == 0
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


# ---------------------------------------------------------------------------------------
# bench driver for N > 1 (called by bench.py)
# ---------------------------------------------------------------------------------------
def run_distributed_bench(args, rank, world, local_rank, mdir, dims):
    """`python bench.py --gpus N` (one rank per GPU, RCCL).  Weak scaling: N clips of `seconds` s, one per GPU.

    Two ways of running the same job are timed in the same process, K steps each, barrier + device synchronisation
    on both sides, MAX over ranks:
      sharded  (the north star's path, = `value`): every clip's encoder positions are split over all N GPUs (exact
               context parallelism: per-layer K/V halo to the right neighbour over xGMI, RCCL gather of the adapter
               rows to the clip's GPU), then every GPU runs the strictly sequential decoder of its own clip;
      replica  (`replica` in the JSON): every GPU transcribes its own clip alone through the ordinary voxtral.h stream
               API - no communication.  The decoder is 95 % of a transcription and does not shard, and an
               N-way shard of a 30 s encoder (212 rows at N = 8) is launch-bound, so the replica figure is expected to
               be the better one; both are reported so that the driver's scaling curve can be read.
    Every rank's clip is the golden 30 s night1968 input, and both passes' token ids are compared with the reference's
    (`parity`)."""
    import torch
    import torch.distributed as dist
    import voxtral_c_amd as v
    from bench import headline_audio, parity_block

    share = os.environ.get("VOX_SHARE_GPU") == "1"          # several ranks on one GPU (tests): RCCL refuses that, so gloo
    backend = os.environ.get("VOX_DIST_BACKEND", "gloo" if share else "nccl")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # one node, no fabric: keep RCCL's bootstrap off interface / InfiniBand probing (seen to stall
    # communicator creation for ~2 minutes on boxes without a network)
    os.environ.setdefault("NCCL_IB_DISABLE", "1")
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    dev = 0 if share else local_rank
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group(backend=backend, device_id=torch.device(f"cuda:{dev}"))
    else:
        dist.init_process_group(backend=backend)
    win = {} if args.preset != "tiny" else dict(enc_window=48, dec_window=64)
    model = v.Model(mdir, device=dev, **win)
    v.hip.vox_hip_stream_handle.restype = C.c_void_p
    v.hip.vox_hip_stream_handle.argtypes = [C.c_void_p]
    comm = TorchComm(device=f"cuda:{dev}", engine_stream=v.hip.vox_hip_stream_handle(model.engine))
    session = DistributedSession(model, comm)
    audio, golden, golden_name, audio_desc = headline_audio(args.seconds)
    # VOX_DIST_MODE=single: one clip of N x seconds, decoder on rank 0 only (BASELINE config 4).  With a golden for that length
    # (8 x 75 s = the 600 s fixture stream_full_batch600.npz) rank 0's ids are checked against the reference's run.
    single = os.environ.get("VOX_DIST_MODE") == "single"
    golden_all = golden_all_name = None
    if single:
        audio_all, golden_all, golden_all_name, audio_desc = headline_audio(args.seconds * world)
    if args.preset != "full":
        golden = golden_all = None
    dev_t = comm.device if comm.on_gpu else "cpu"

    def allmax(x):
        t = torch.tensor([float(x)], dtype=torch.float64, device=dev_t)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def allsum(x):
        t = torch.tensor([float(x)], dtype=torch.float64, device=dev_t)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def timed(fn):
        for _ in range(args.warmup):
            fn()
        comm.barrier(); torch.cuda.synchronize()
        v.hip.vox_hip_reset_timing(model.engine)
        t0 = time.time()
        r = None
        for _ in range(args.steps):
            r = fn()
            comm.barrier()
        torch.cuda.synchronize()
        return allmax(time.time() - t0), r, model.timing()

    # ---- replica pass FIRST: every GPU alone on its own clip, ordinary stream API (its only collectives are the barrier and
    # the all-reduces of the timing: if RCCL works at all, this line exists) ------------------------------------------------
    rep = None
    if not single:
        rwall, rres, rt = timed(lambda: model.transcribe(audio))
        rpar = parity_block(rres["tokens"], golden, golden_name)
        rep = {"value": round(rwall / args.steps / (args.seconds * world), 5), "ms_per_step": round(rwall * 1e3 / args.steps, 2),
               # (vox_stream_init resets the engine's phase timers: these are the LAST pass's)
               "decode_tok_s": round(allsum(rt["decode_steps"]) / (allmax(rt["decode_ms"]) * 1e-3), 1) if rt["decode_ms"] > 0 else 0.0,
               "encode_ms": round(allmax(rt["encode_ms"]), 2), "prefill_ms": round(allmax(rt["prefill_ms"]), 2),
               "parity_mismatches_all_ranks": int(allsum(rpar.get("mismatches", 0) if rpar.get("checked") else 0)),
               "parity_checked_ranks": int(allsum(1 if rpar.get("checked") else 0)),
               "what": "every GPU transcribes its own clip alone (vox_stream_feed + finish), no communication"}
    audio_s = args.seconds * world

    def base_line():
        return {
            "metric": "real-time-factor + decode tokens/sec, Voxtral-4B bf16, 30s audio",
            "unit": "wall s / audio s (RTF)", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 weights, f32 activations/accumulate", "data": "synthetic",
            "data_note": "weights: seeded synthetic checkpoint of the exact architecture; audio (every rank): " + audio_desc,
            "replica": rep, "rccl_ranks": world, "backend": backend,
            "rccl_version": ".".join(str(x) for x in torch.cuda.nccl.version()) if backend == "nccl" else None,
        }

    # The stream-ordered point-to-point path (RCCL send / recv issued on the engine's own stream) cannot be exercised with more
    # than one rank before the first multi-GPU node runs it.  If it raises or does not come back in time, the job still ends
    # with one honest JSON line: the replica figure as `value`, and what happened to the sharded pass next to it.
    import threading
    limit_s = float(os.environ.get("VOX_DIST_SHARD_TIMEOUT", 60.0 + 30.0 * (args.steps + args.warmup)))
    state = {"done": False}

    def fallback(reason):
        # The sharded pass is what `value` means on this line.  If it raised or hung there is NO value: the line says so
        # (value null, the reason, the replica figure only under `replica`) and every rank exits non-zero - a broken or
        # deadlocked multi-GPU path must not look like a green run whose number means something else.
        if state["done"]:
            return
        state["done"] = True
        if rank == 0:
            out = base_line()
            out.update({"value": None, "ms_per_step": None,
                        "sharded_pass": {"completed": False, "reason": reason},
                        "config": {"workload": f"Voxtral-4B ({args.preset} synthetic checkpoint), {world} clips of {args.seconds:g} s, one per GPU; "
                                               "the sharded-encoder pass did not complete (see sharded_pass.reason); `replica` holds the "
                                               "no-communication figure measured before it",
                                   "audio_seconds": audio_s, "parallelism": f"cp{world} encoder (failed)", "backend": backend}})
            print(json.dumps(out), flush=True)
        os._exit(4)

    timer = threading.Timer(limit_s, fallback, args=(f"no result after {limit_s:.0f} s (hang in the RCCL point-to-point path?)",))
    timer.daemon = True
    timer.start()
    try:
        # ---- sharded pass ---------------------------------------------------------------------------------------------
        if os.environ.get("VOX_DIST_INJECT_FAIL") == "1":        # test hook for the fallback line
            raise RuntimeError("injected failure of the sharded pass (VOX_DIST_INJECT_FAIL)")
        wall, toks, t = timed((lambda: session.transcribe(audio_all)) if single else (lambda: session.transcribe_many([audio] * world)))
        phase = {k: allmax(val) for k, val in sorted(session.phase_ms.items())}
        phase["prefill"] = allmax(t["prefill_ms"] / max(args.steps, 1))
        phase["decode"] = allmax(t["decode_ms"] / max(args.steps, 1))
        dec_steps, dec_ms = allsum(t["decode_steps"]), allmax(t["decode_ms"])
        if single:      # only rank 0 decodes
            par = parity_block(toks, golden_all, golden_all_name) if toks is not None else {"checked": False, "reason": "this rank does not decode"}
        else:
            par = parity_block(toks, golden, golden_name) if toks is not None else {"checked": False, "reason": "no tokens on this rank"}
        mism = allsum(par.get("mismatches", 0) if par.get("checked") else 0)
        checked = allsum(1 if par.get("checked") else 0)
        wf_syncs = allmax(session.eng.wavefront_syncs)
        n_tok = allsum(len(toks) if toks is not None else 0)
    except Exception as ex:            # noqa: BLE001 - whatever it was, report it instead of losing the whole line
        timer.cancel()
        fallback(f"{type(ex).__name__}: {ex}")
    timer.cancel()
    if state["done"]:
        return
    state["done"] = True

    if rank == 0:
        out = base_line()
        out.update({
            "value": round(wall / args.steps / audio_s, 5), "ms_per_step": round(wall * 1e3 / args.steps, 2),
            "decode_tok_s": round(dec_steps / (dec_ms * 1e-3), 1) if dec_ms > 0 else 0.0,
            "decode_tok_s_per_stream": round(t["decode_steps"] / (t["decode_ms"] * 1e-3), 1) if t["decode_ms"] > 0 else 0.0,
            "decoder_steps_per_pass": int(n_tok),
            "phases_ms": {k: round(val, 2) for k, val in phase.items()},
            "parity": dict(par, mismatches_all_ranks=int(mism), checked_ranks=int(checked)),
            "sharded_pass": {"completed": True},
            "host_syncs_in_wavefront": int(wf_syncs),
            "config": {"workload": (f"Voxtral-4B ({args.preset} synthetic checkpoint), one {audio_s:g} s clip: encoder positions sharded over "
                                    f"{world} GPUs (wavefront K/V halo over xGMI), adapter rows gathered to rank 0, single-stream greedy "
                                    "decode on rank 0") if single else
                                   (f"Voxtral-4B ({args.preset} synthetic checkpoint), {world} clips of {args.seconds:g} s (one per GPU): each clip's "
                                    f"encoder positions are sharded over all {world} GPUs (wavefront K/V halo over xGMI, RCCL gather of the "
                                    "adapter rows to the clip's GPU), then every GPU runs the single-stream greedy decoder of its own clip"),
                       "audio_seconds": audio_s,
                       "parallelism": f"cp{world} encoder (N-way shards of a 30 s encoder are launch-bound: expected slower per clip than the "
                                      f"replica figure reported beside it) / " + ("1 decoder" if single else f"{world} decoders (replicas)"),
                       "backend": backend},
        })
        try:        # same live roofline measurement as the 1-GPU line (rank 0's engine)
            from bench import roofline_block
            out["roofline"] = roofline_block(v, model, model.dims, n_tok / world if n_tok else 380.0, pmc=False)
        except Exception as ex:
            out["roofline"] = {"error": str(ex)}
    else:
        out = None

    # ---- second phase: BASELINE config 4 itself - ONE clip of world x 75 s (600 s at 8 GPUs), its encoder positions sharded over all
    # GPUs, adapter rows delivered to rank 0 block by block, single-stream greedy decode on rank 0 - reported under `config4` of the
    # same line (`value` stays the weak-scaling figure above).  It runs under its own watchdog: if it raises or hangs, the line
    # still goes out, with config4.completed = false and the reason.
    c4 = None
    if not single and world > 1 and args.preset == "full" and os.environ.get("VOX_BENCH_CONFIG4", "1") != "0":
        c4_seconds = float(os.environ.get("VOX_BENCH_CONFIG4_SECONDS", 75.0 * world))
        c4_state = {"done": False}
        c4_lock = threading.Lock()           # the watchdog thread and the main thread both "claim" the line: exactly one of them prints it

        def c4_claim():
            with c4_lock:
                if c4_state["done"]:
                    return False
                c4_state["done"] = True
                return True

        def c4_fallback(reason):
            if not c4_claim():
                return
            if rank == 0:
                out["config4"] = {"completed": False, "reason": reason, "audio_seconds": c4_seconds}
                print(json.dumps(out), flush=True)
            os._exit(0)          # the line's `value` is valid: only the second phase is missing

        c4_timer = threading.Timer(float(os.environ.get("VOX_DIST_CONFIG4_TIMEOUT", 120.0 + 0.2 * c4_seconds)), c4_fallback,
                                   args=("no result in time (hang in the single-clip pass?)",))
        c4_timer.daemon = True
        c4_timer.start()
        try:
            from bench import prefix_parity_block
            audio4, golden4, golden4_name, desc4 = headline_audio(c4_seconds)
            if args.warmup > 0:
                session.transcribe(audio4)
            comm.barrier(); torch.cuda.synchronize()
            v.hip.vox_hip_reset_timing(model.engine)
            t0 = time.time()
            toks4 = session.transcribe(audio4)
            comm.barrier(); torch.cuda.synchronize()
            wall4 = allmax(time.time() - t0)
            t4 = model.timing()
            phase4 = {k: allmax(val) for k, val in sorted(session.phase_ms.items())}
            phase4["prefill"] = allmax(t4["prefill_ms"]); phase4["decode"] = allmax(t4["decode_ms"])
            steps4, dec_ms4 = allsum(t4["decode_steps"]), allmax(t4["decode_ms"])
            syncs4 = allmax(session.eng.wavefront_syncs)
            if toks4 is not None:
                par4 = parity_block(toks4, golden4, golden4_name) if golden4 is not None else prefix_parity_block(toks4, c4_seconds)
            else:
                par4 = {"checked": False, "reason": "this rank does not decode"}
            mism4 = allsum(par4.get("mismatches", 0) if par4.get("checked") else 0)
            c4 = {"completed": True, "value": round(wall4 / c4_seconds, 5), "unit": "wall s / audio s (RTF), one pass",
                  "audio_seconds": c4_seconds, "wall_ms": round(wall4 * 1e3, 2),
                  "decode_tok_s": round(steps4 / (dec_ms4 * 1e-3), 1) if dec_ms4 > 0 else 0.0, "decoder_steps": int(steps4),
                  "phases_ms": {k: round(val, 2) for k, val in phase4.items()}, "host_syncs_in_wavefront": int(syncs4),
                  "parity": dict(par4, mismatches_all_ranks=int(mism4)),
                  "workload": (f"BASELINE config 4 at {world} GPUs: one {c4_seconds:g} s clip ({desc4}), encoder positions sharded over the {world} GPUs "
                               "(exact context parallelism, per-layer K/V halo to the right neighbour), adapter rows delivered to rank 0 block by "
                               "block and decoded as they arrive, single-stream greedy decode on rank 0")}
        except Exception as ex:            # noqa: BLE001
            c4_timer.cancel()
            c4_fallback(f"{type(ex).__name__}: {ex}")
        c4_timer.cancel()
        if not c4_claim():
            return
    if rank == 0:
        if c4 is not None:
            out["config4"] = c4
        print(json.dumps(out), flush=True)
    comm.barrier()
    model.close()
    dist.destroy_process_group()
