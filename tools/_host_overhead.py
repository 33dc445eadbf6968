import importlib, sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg = importlib.import_module(bench.PKG)
pool = bench.SlotPool(pkg, 0, 8)
cell = bench.Cell(pkg, 0, 0, 4, 2, 1, pool=pool, n_buf=8)     # 4 antennas, 2 slots: negligible GPU work
for _ in range(20): pool.submit(cell)
pool.drain(); pool.sync()
t0 = time.perf_counter()
n = 400
for _ in range(n): pool.submit(cell)
pool.drain(); pool.sync()
print("host + launch overhead per CPI: %.3f ms" % (1e3 * (time.perf_counter() - t0) / n))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200): pool.submit(cell)
pool.drain(); pool.sync()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
