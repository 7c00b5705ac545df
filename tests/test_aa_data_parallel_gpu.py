"""Data parallelism through the real encoder and the gradient sink: two gloo ranks on the GPU box's one MI355X.

The two rank processes (tests/dp_rank_worker.py) are started by ``conftest.pytest_collection_finish`` BEFORE this
pytest process touches the GPU, so that the ranks' first HIP calls do not compete with a GPU-initialised parent for the
box's limit of six GPU-using processes (starting a fresh child from a GPU-initialised process is fine -- golden_utils.
OracleRuns does it; what this pool forbids is REPLACING such a process, os.exec*); this module only waits for them and
checks what they reported.  (Named ``test_aa_`` so that it is collected first.)
"""
import pytest

pytestmark = pytest.mark.gpu


def test_two_ranks_stay_bit_identical_through_the_gradient_sink(dp_rank_processes):
    procs = dp_rank_processes
    assert procs, "conftest did not start the rank processes"
    outs = []
    for p, log in procs:
        try:
            rc = p.wait(timeout=600)
        except Exception:
            p.kill()
            raise
        with open(log) as f:
            outs.append((rc, f.read()))
    for rank, (rc, text) in enumerate(outs):
        assert rc == 0 and "DP-RANK-OK" in text, "rank %d failed (exit %s):\n%s" % (rank, rc, text[-4000:])
    for cell in ("GRU", "LSTM"):
        assert "%s bucketed=1" % cell in outs[0][1] and "%s bucketed=0" % cell in outs[0][1]
        assert "%s two backwards per step" % cell in outs[0][1]
        # the full model with tied embeddings (HierPropertyVAE, vae_train.py:78-83 order)
        assert "VAE %s tie_embedding: ranks bit-identical" % cell in outs[0][1]
        assert "VAE %s 2-rank all-reduced gradient vs 1-rank" % cell in outs[0][1]
