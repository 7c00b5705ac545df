"""Development settings: alternative forms of the host path that exist for A/B measurements and for the parity tests.

None of these is an environment switch.  Every default is the measured-fastest form (DESIGN.md sections 6, 8, 9 carry the
figures); a test or a tool under ``tools/`` changes an attribute here (``monkeypatch.setattr(_dev, "ATOM_COMPACT", False)``)
to run the other form against the same fixtures; every module reads these attributes at CALL time (none copies one
into a module global at import), so a change made after the package was imported takes effect.  The runtime switches a user of the package sees are the few in
README.md; the compile-time tuning switches of the kernels live in the ``dev`` variant of the library
(``python -m ggpm_amd.build --variant dev -DGGPM_DEV_SWITCHES``, csrc/common.h: ggpm_dev_env).
"""

# ---- the teacher-forced decoder (decoder.py, atom_decode.py, tree_decode.py; reference ggpm/decoder.py:166-284)
DECODER_BATCHED = True     # tree-side levels as ONE level call each over the decode-time DAG (False: the reference's step loop)
ATOM_DECODE = True         # atom level as one autograd node on host-built index tables (False: through IncMPNEncoder per step)
ATOM_COMPACT = True        # ... on the compact row set of every decode step (False: all rows of the level, frozen mask)
ATOM_AHEAD = True          # ... issued on its own stream BEFORE the encoder, joined where the attachment level needs it
ATOM_PRIORITY = True       # ... on a high-priority stream
DECODE_DRIVER = True       # the two step loops as one C call each (csrc/decode.hip; False: launches issued from Python)
ATOM_ASYNC = True          # ... issued by a worker thread of the library (ggpm_decode_steps_*_async)
PACK_ONCE = True           # the decode steps share one packed weight set (False: every step packs again)
TREE_COMPOSITE = True      # each tree-side decoder level as one autograd node (False: op by op)
TREE_DRIVER = True         # ... whose two directions are one C call each (csrc/tree_level.hip; False: ~30 ctypes calls)
HEADS_COMPOSITE = True     # the four score heads + losses + accuracies as ONE autograd node (heads_fused.py; False: ~30 nodes)
TREE_WGRADS_ASIDE = True   # ... and whose parameter gradients are formed on the second stream, beside the rest of the backward
ENC_NARROW = True          # the encoder's levels take two row tiles per workgroup while they run beside the atom-level chain
                           # ("fwd" / "bwd": in that direction only)

# ---- gradients (functional.py, parallel.py, optim.py)
DEFER_EARLY = True         # deferred weight-gradient contractions start beside the atom level's backward (second stream)
INDEX_MEMO = True          # CSRs / transposes / masks derived from RESIDENT index tensors are remembered on them
RECORD_GRADS = False       # published gradients are marked with record_stream(main) (an event record each when released)
GRAD_SINK = True           # the encoder's backward writes its gradients straight into FlatGradSync's flat buffer
HIP_ADAM = True            # FlatAdam's update as one ggpm_adam_step launch (False: torch's fused Adam)

# ---- metrics (property_vae.py)
METRICS_ASYNC = True       # the metrics' copy to pinned memory is enqueued behind the forward and read through its event
