#!/usr/bin/env python3
"""Development probe: the fused echo-synthesis + range kernel at the bench shape -- kernel time from the library's own HIP events, and a digest
of everything the CPI produced (echo grid + estimate), so that two builds (or two settings of a development switch) can be compared for
bit-identity across processes.  ER_TARGETS = number of LoS targets (1: digest 99744bff7fea44d1, 2: 4272ba4af33d73b2 since round 3)."""
import ctypes as C, hashlib, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import numpy as np
pkg = importlib.import_module(bench.PKG)
n_targets = int(os.environ.get("ER_TARGETS", "1"))
cell = bench.Cell(pkg, 0, 0, 64, 16, n_targets, inflight=1)
cell.ctx.check(cell.ctx.lib.isac_profile_enable(cell.ctx.handle, 1))
cell.profile_sink = []
for i in range(40):
    cell.n_sub = 0                      # same seed, same buffers every step
    est = cell.step()
ms = np.array(cell.profile_sink[15:])
h = hashlib.sha256(cell.echo[0].numpy().tobytes())
for k in sorted(vars(est)) if est is not None and hasattr(est, "__dict__") else []:
    v = getattr(est, k)
    if isinstance(v, np.ndarray):
        h.update(np.ascontiguousarray(v).tobytes())
print(f"targets {n_targets}: fused kernel {ms.mean():.4f} ms (min {ms.min():.4f}, max {ms.max():.4f}) digest {h.hexdigest()[:16]}")
