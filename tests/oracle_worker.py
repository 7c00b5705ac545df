"""One oracle evaluation in its own process (test infrastructure; started by golden_utils.oracle_runs_parallel).

    python tests/oracle_worker.py JOB.pkl OUT.pkl

JOB.pkl: dict(rnn, depth, sd, tree, graph, dtype="f32"|"f64", hoisted, reverse, threads) -- the arguments of
golden_utils.oracle_encoder_result; OUT.pkl: its result.  Never touches the GPU (CPU torch only)."""
import os
import pickle
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import time
    t0 = time.time()
    import torch
    import golden_utils as G
    t1 = time.time()
    with open(sys.argv[1], "rb") as f:
        job = pickle.load(f)
    tree, graph = job["tree"], job["graph"]
    if job.get("reverse"):
        tree, graph = G.reversed_batch(tree, graph)
    if job.get("threads"):
        torch.set_num_threads(int(job["threads"]))
    res = G.oracle_encoder_result(job["rnn"], job["depth"], job["sd"], tree, graph,
                                  dtype=torch.float64 if job.get("dtype") == "f64" else torch.float32,
                                  hoisted=bool(job.get("hoisted")))
    t2 = time.time()
    with open(sys.argv[2] + ".tmp", "wb") as f:
        pickle.dump(res, f, protocol=pickle.HIGHEST_PROTOCOL)
    os.replace(sys.argv[2] + ".tmp", sys.argv[2])
    print("[oracle worker] imports %.1f s, evaluation %.1f s (%s threads), result written %.1f s"
          % (t1 - t0, t2 - t1, torch.get_num_threads(), time.time() - t2), flush=True)


if __name__ == "__main__":
    main()
