"""Dev diagnostic: run-to-run reproducibility of the full VAE step's gradients (graphs= vs schedule=, repeated)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_gpu_training_loop import _Configs, _init_like_vae_train
from ggpm_amd import synth
from ggpm_amd.decoder import DecodeSchedule
from ggpm_amd.property_vae import HierPropertyVAE
from ggpm_amd.vocab import IndexPairVocab

vocab = IndexPairVocab(40, 120)
configs = _Configs(vocab, rnn_type=os.environ.get("RNN", "GRU"), hidden_size=64, embed_size=64, latent_size=16, depthT=6, depthG=6, dropout=0.0)
torch.manual_seed(5)
model = HierPropertyVAE(configs).to("cuda:0")
_init_like_vae_train(model)
specs = synth.random_batch(9, 6, motifs=(2, 7), n_motif_vocab=40, n_attach_vocab=120)
batch = synth.train_batch(specs)


def run(**kw):
    model.zero_grad()
    loss, m = model(*batch, beta=0.3, perturb_z=False, **kw)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss.detach()), {k: p.grad.clone() for k, p in model.named_parameters()}


runs = {"A graphs": run(), "B schedule": run(schedule=DecodeSchedule.from_specs(specs, batch[2])),
        "C schedule": run(schedule=DecodeSchedule.from_specs(specs, batch[2])), "D graphs": run()}
names = list(runs)
for i in range(len(names)):
    for j in range(i + 1, len(names)):
        (la, ga), (lb, gb) = runs[names[i]], runs[names[j]]
        bad = {k: float((ga[k] - gb[k]).abs().max()) / (float(ga[k].abs().max()) + 1e-30) for k in ga if not torch.equal(ga[k], gb[k])}
        print("%s vs %s: loss equal %s, %d tensors differ; worst %s" % (names[i], names[j], la == lb, len(bad),
              sorted(bad.items(), key=lambda kv: -kv[1])[:4]))
