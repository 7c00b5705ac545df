"""Host enqueue time of each part of the bench step (dev tool; no profiler, no synchronisation inside the step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
from ggpm_amd import _dev
from ggpm_amd.nnutils import make_cuda
from ggpm_amd.parallel import FlatGradSync
from ggpm_amd.property_vae import HierEncoderVAE, rsample

rnn = os.environ.get("RNN", "GRU")      # CONFIG=<BASELINE config index> picks the workload
dev = torch.device("cuda:0")
CFG = bench.CONFIGS[int(os.environ.get("CONFIG", "1"))]
pool = bench.make_batches(8, CFG["batch"], seed0=1000, gen=CFG["gen"], n_motif=CFG["vocab"][0], n_attach=CFG["vocab"][1])
dev_batches = [make_cuda(b) for b in pool]
torch.manual_seed(0)
model = HierEncoderVAE(bench.make_args(rnn, CFG["hidden"], CFG["depth"], CFG["latent"], *CFG["vocab"])).to(dev)
if _dev.HIP_ADAM:      # (False: torch.optim.Adam, for A/B runs)
    from ggpm_amd.optim import FlatAdam
    sync = FlatGradSync(model.parameters(), encoder=model.encoder, keep_flat=True)
    opt = FlatAdam(sync, lr=1e-3)
else:
    sync = FlatGradSync(model.parameters(), encoder=model.encoder)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
acc = {}

# time spent inside the two C driver calls themselves (the rest of "encoder forward" / "backward" is Python + autograd)
from ggpm_amd import _lib as _L
_lib_obj = _L.load()
_c_time = {"ggpm_encoder_forward": 0.0, "ggpm_encoder_backward": 0.0}


class _Timed:
    def __init__(self, name):
        self.name, self.fn = name, getattr(_lib_obj, name)

    def __call__(self, *a):
        t0 = time.perf_counter()
        r = self.fn(*a)
        _c_time[self.name] += time.perf_counter() - t0
        return r


class _Proxy:
    def __getattr__(self, k):
        return _Timed(k) if k in _c_time else getattr(_lib_obj, k)


_L.load = lambda *a, **k: _Proxy()

# finer: the pinned-ring upload and the autograd Function call of the encoder forward
from ggpm_amd import encoder as _E, fused as _Fz
_big = []
_sub = {"ring upload": 0.0, "_HierEncoder.apply": 0.0, "torch.empty(saved)": 0.0}
_up = _E._RING.upload


def _upload(values, device):
    t0 = time.perf_counter()
    r = _up(values, device)
    _sub["ring upload"] += time.perf_counter() - t0
    return r


_E._RING.upload = _upload
_apply = _Fz._HierEncoder.apply


def _timed_apply(*a):
    t0 = time.perf_counter()
    r = _apply(*a)
    _sub["_HierEncoder.apply"] += time.perf_counter() - t0
    return r


_Fz._HierEncoder.apply = staticmethod(_timed_apply)
_empty = torch.empty


def _timed_empty(*a, **k):
    big = len(a) == 1 and isinstance(a[0], int) and a[0] > (1 << 24)
    if not big:
        return _empty(*a, **k)
    t0 = time.perf_counter()
    r = _empty(*a, **k)
    dt = time.perf_counter() - t0
    _sub["torch.empty(saved)"] += dt
    _big.append((a[0], dt, torch.cuda.memory_reserved()))
    return r


torch.empty = _timed_empty


def lap(name, t0):
    t1 = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t1 - t0)
    return t1


def step(i, record):
    tree, graph = dev_batches[i % len(dev_batches)]
    t = time.perf_counter()
    sync.zero_grad(); t = lap("zero_grad", t) if record else time.perf_counter()
    outs = model.encoder.forward_padded(tree, graph); t = lap("encoder forward", t) if record else time.perf_counter()
    _, kl = rsample(outs[0], model.R_mean, model.R_var, perturb=False); t = lap("rsample", t) if record else time.perf_counter()
    loss = 0.1 * kl + 1e-3 * (outs[0].sum() + outs[1].sum() + outs[2].sum() + outs[3].sum())
    t = lap("loss", t) if record else time.perf_counter()
    loss.backward(); t = lap("backward", t) if record else time.perf_counter()
    sync.all_reduce(); t = lap("all_reduce", t) if record else time.perf_counter()
    opt.step(); t = lap("adam", t) if record else time.perf_counter()


for i in range(40):
    step(i, False)
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for i in range(N):
    step(i, True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.3f ms/step, total %.3f ms/step" % (1e3 * (t1 - t0) / N, 1e3 * (t2 - t0) / N))
for k, v in acc.items():
    print("  %-16s %.3f ms" % (k, 1e3 * v / N))
for k, v in _sub.items():
    print("  sub %-27s %.3f ms (/ %d)" % (k, 1e3 * v / (N + 40), N + 40))
for n, dt, res in _big[-12:]:
    print("    big alloc %6.1f MB took %8.1f us, reserved %.2f GB" % (n / 1e6, dt * 1e6, res / 1e9))
for k, v in _c_time.items():
    print("  inside %-24s %.3f ms (all %d steps incl. warm-up: / %d)" % (k, 1e3 * v / (N + 40), N + 40, N + 40))
