#!/usr/bin/env python3
"""Development probe: phases of eigh_tridiag_small_kernel / music_subspace_kernel at n <= 64 (ISAC_DEBUG prints thread 0's cycle counters / 64) and the
kernel's bits (digest of d, e through the eigenvalues + the top vectors).   ISAC_DEBUG=1 python tools/tridiag_small_probe.py [n]"""
import hashlib, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module(bench.PKG)
ctx = pkg.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(n)
m = rng.standard_normal((n, 400)) + 1j * rng.standard_normal((n, 400))
sv = np.exp(-1j * np.pi * np.arange(n) * 0.3)[:, None] * (rng.standard_normal((1, 400)) + 1j * rng.standard_normal((1, 400))) * 6.0
h = np.asfortranarray((m + sv) @ (m + sv).conj().T / 400)
for _ in range(4):
    w, u = ctx.eigh_top(h, 1)
wr = np.linalg.eigvalsh(h)
print("max |w - numpy|", np.abs(w - wr).max() / np.abs(wr).max(), "residual", np.abs(h @ u - u * wr[-1]).max() / np.abs(wr).max())
print("digest", hashlib.sha256(w.tobytes() + u.tobytes()).hexdigest()[:16])
ctx.sync(); ctx.timer_start()
for _ in range(20):
    w, u = ctx.eigh_top(h, 1)
print("host wall per eigh_top call incl. copies: %.1f us" % (1e3 * ctx.timer_stop_ms() / 20))
