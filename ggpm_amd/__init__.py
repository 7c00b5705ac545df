"""ggpm_amd -- MI355X (gfx950) native hierarchical message passing for ggpm's HierMPNEncoder path.

Host side: Python mirror of the reference's classes (rnn.GRU/LSTM, encoder.MPNEncoder/HierMPNEncoder);
device side: hand-written HIP kernels behind the C ABI in include/ggpm_hip.h (ggpm_amd/libggpm_hip.so).
"""
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
