#!/usr/bin/env python3
"""Development probe: where does the host spend its time per CPI at the bench shape (un-paced pipeline, 8 contexts)?  cProfile over 300
submissions + wall time per submit; ISAC_TRACE_CALLS=1 in the library is not needed -- the two C entry points show up as the ctypes calls
inside monoStaticSensing / fft2D_submit."""
import cProfile, importlib, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg = importlib.import_module(bench.PKG)
n_ctx = int(os.environ.get("N_CTX", "8"))
pool = bench.SlotPool(pkg, 0, n_ctx)
cell = bench.Cell(pkg, 0, 0, 64, 16, 1, pool=pool, n_buf=n_ctx)
for _ in range(400): pool.submit(cell)
pool.drain(); pool.sync()
t0 = time.perf_counter()
n = 300
for _ in range(n): pool.submit(cell)
pool.drain(); pool.sync()
print("wall per CPI: %.3f ms" % (1e3 * (time.perf_counter() - t0) / n))
pr = cProfile.Profile(); pr.enable()
for _ in range(n): pool.submit(cell)
pool.drain(); pool.sync()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
