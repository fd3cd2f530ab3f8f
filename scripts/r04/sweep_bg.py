#!/usr/bin/env python3
"""A/B of the background-context settings inside ONE process (GPU box): for every (NS_BG_PRIO, NS_UCOOP_SHIFT) the headline step of
bench.py (configs[1], two engine contexts) timed over a few steps.  python scripts/r04/sweep_bg.py [steps]"""
import os
import sys
import tempfile
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
a = types.SimpleNamespace(dist_backend="nccl", unaligned_delay_ms=0.0)
tmp = tempfile.mkdtemp(prefix="nssweep_")
n = 1_000_000
for prio, shift in ((1, 3), (0, 3), (1, 2), (1, 4), (1, 5), (0, 4), (1, 3)):
    os.environ["NS_BG_PRIO"] = str(prio)
    os.environ["NS_UCOOP_SHIFT"] = str(shift)
    w = bench.Workload(a, "ecoli", False, 0, 0, 0, 1, None, False, False, tmp)
    n_al, n_un = w.split(n, False)
    for i in range(2):
        w.step(i, n, n_al, n_un)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    infos = [w.step(2 + i, n, n_al, n_un) for i in range(steps)]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    al = np.mean([st[0].ms_total for st in infos]); un = np.mean([st[1].ms_total for st in infos])
    ch_al = np.mean([st[0].ms_kernel[1] for st in infos]); ch_un = np.mean([st[1].ms_kernel[1] for st in infos])
    print("bg_prio=%d ucoop_shift=%d: %.2f ms/step = %.1f M reads/s | aligned %.2f (chain %.2f) unaligned %.2f (chain %.2f)"
          % (prio, shift, dt * 1e3, n / dt / 1e6, al, ch_al, un, ch_un), flush=True)
    w.close()
