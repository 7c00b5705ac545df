"""profiles/r03_vae_launches.json from rocprofv3 kernel traces of `bench.py --only-vae` (dev tool):

    python tools/vae_launches.py GRU=<trace dir> LSTM=<trace dir> > profiles/r03_vae_launches.json

bench.py quotes `launches_per_step` and the per-kernel-class table of the full VAE step from this file."""
import json, os, subprocess, sys, tempfile
out = {"source": "rocprofv3 --kernel-trace of `python bench.py --only-vae [--rnn LSTM]`, the last 20 resident steps "
                 "(tools/prof_summary.py --steps 20); figures per step"}
for arg in sys.argv[1:]:
    cell, d = arg.split("=", 1)
    tmp = tempfile.mktemp(suffix=".json")
    subprocess.check_call([sys.executable, os.path.join(os.path.dirname(__file__), "prof_summary.py"), d, "--steps", "20",
                           "--json", tmp], stdout=subprocess.DEVNULL)
    out[cell] = json.load(open(tmp))
print(json.dumps(out, indent=1))
