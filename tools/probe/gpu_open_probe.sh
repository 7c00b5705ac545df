# Does a CPU-only torch process open the GPU device nodes?  (the box allows 6 processes with the GPU open: the oracle
# workers of the parity tests must not count)  ->  lists /dev/kfd and /dev/dri descriptors after import, CPU work, backward
for mode in plain hidden; do
  if [ $mode = hidden ]; then export ROCR_VISIBLE_DEVICES=""; export HIP_VISIBLE_DEVICES=""; fi
  python -c "
import os, sys, torch
def gpu_fds():
    out = []
    for f in os.listdir('/proc/self/fd'):
        try:
            t = os.readlink('/proc/self/fd/%s' % f)
        except OSError:
            continue
        if 'kfd' in t or 'dri' in t:
            out.append(t)
    return out
print('$mode after import torch:', gpu_fds())
x = torch.randn(100, 100, requires_grad=True)
y = (x @ x).sum()
print('$mode after cpu forward:', gpu_fds())
y.backward()
print('$mode after cpu backward:', gpu_fds())
"
done
