"""Worker for tests/test_gpu_multi.py: one rank of a multi-process transcription.
Launched by torch.distributed.run; backend and GPU sharing come from the environment."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    preset, seconds, seed, out_path = sys.argv[1], float(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import torch
    import torch.distributed as dist
    import voxtral_c_amd as v
    from audio_util import synth_speech
    from conftest import model_dir
    from voxtral_c_amd.multi_gpu import DistributedSession, TorchComm

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev = 0 if os.environ.get("VOX_SHARE_GPU") == "1" else local_rank
    torch.cuda.set_device(dev)
    dist.init_process_group(backend=os.environ.get("VOX_DIST_BACKEND", "nccl"))
    win = {} if preset != "tiny" else dict(enc_window=48, dec_window=64)
    model = v.Model(model_dir(preset), device=dev, **win)
    import ctypes as C
    v.hip.vox_hip_stream_handle.restype = C.c_void_p
    v.hip.vox_hip_stream_handle.argtypes = [C.c_void_p]
    comm = TorchComm(device=f"cuda:{dev}", engine_stream=v.hip.vox_hip_stream_handle(model.engine))
    sess = DistributedSession(model, comm)
    if len(sys.argv) > 5 and sys.argv[5] == "many":     # one clip per rank, every encoder sharded over all ranks
        toks = sess.transcribe_many([synth_speech(seconds, seed + r) for r in range(comm.world)])
        np.save(f"{out_path}.{comm.rank}.npy", toks)
    else:
        toks = sess.transcribe(synth_speech(seconds, seed))
        if comm.rank == 0:
            np.save(out_path, toks)
    comm.barrier()
    model.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
