"""ggpm_amd -- MI355X (gfx950) native hierarchical message passing for ggpm's HierMPNEncoder path.

Host side: Python mirror of the reference's classes (rnn.GRU/LSTM, encoder.MPNEncoder/HierMPNEncoder);
device side: hand-written HIP kernels behind the C ABI in include/ggpm_hip.h (ggpm_amd/libggpm_hip.so).
"""
import os as _os

# One process uses up to five HIP streams (main, the high-priority atom-level stream, the second stream for transposes and
# weight gradients, a copy stream, the collective's own) and RCCL brings its own; with the default of 4 hardware queues
# streams share a queue and the encoder's second stream ends up behind the main one (measured: 5.94 instead of
# 5.45 ms/step).  Read by the HIP runtime when it starts, so it is set HERE -- importing the package comes before the
# first HIP call of any program that uses it -- and never overrides a value the user chose.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# multi-process GPU work (RCCL, tensors shared between rank processes): this pool's host driver supports dmabuf IPC only; with
# the legacy mode hipIpcGetMemHandle fails with "invalid argument".  Same rule: before the runtime starts, never overriding.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

__all__ = ["GRU", "LSTM", "MPNEncoder", "HierMPNEncoder", "MotifEncoder", "IncMPNEncoder", "IncHierMPNEncoder",
           "IncEncoder", "HierEncoderVAE", "rsample", "make_cuda", "DevicePrefetcher"]


def __getattr__(name):
    # lazy: importing the package (e.g. for ggpm_amd.synth on a CPU-only host) must not need torch.cuda
    if name in ("GRU", "LSTM"):
        from . import rnn
        return getattr(rnn, name)
    if name in ("MPNEncoder", "HierMPNEncoder", "MotifEncoder", "PreparedBatch"):
        from . import encoder
        return getattr(encoder, name)
    if name in ("IncMPNEncoder", "IncHierMPNEncoder", "IncEncoder", "HTuple"):
        from . import inc_encoder
        return getattr(inc_encoder, name)
    if name in ("HierEncoderVAE", "rsample"):
        from . import property_vae
        return getattr(property_vae, name)
    if name == "DevicePrefetcher":
        from .dataloader import DevicePrefetcher
        return DevicePrefetcher
    if name == "make_cuda":
        from .nnutils import make_cuda
        return make_cuda
    raise AttributeError(name)
