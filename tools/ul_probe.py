import importlib, os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import bench
pkg = importlib.import_module(bench.PKG)
CM = pkg.communication.channelModels
ctx = pkg.Context(0)
T = 61440 + 400
rng = np.random.default_rng(1)
x = ctx.to_device(np.asfortranarray(rng.standard_normal((T, 2)) + 1j * rng.standard_normal((T, 2))))
for prof in ("CDL-D", "CDL-A"):
    chans = [CM.CDLChannel(DelayProfile=prof, TransmitAntennaArraySize=(1, 1, 2, 1, 1), ReceiveAntennaArraySize=(4, 8, 2, 1, 1)) for _ in range(20)]
    outs = [ctx.empty((T, 64)) for _ in chans]
    gains = ctx.empty((20 * 4 * 23 * 2 * 64,))
    ctx.check(ctx.lib.isac_profile_enable(ctx.handle, 1))
    import ctypes as C
    ms = []
    for i in range(6):
        CM.applyCDLBatch(chans, [x] * 20, ctx=ctx, outs=outs, gains=gains); ctx.sync()
        v = C.c_double(0.0); ctx.check(ctx.lib.isac_profile_last_kernel_ms(ctx.handle, C.byref(v))); ms.append(v.value)
    print(prof, "UL fused kernel, 20 jobs: %.3f ms = %.1f us per job" % (np.mean(ms[1:]), 1e3 * np.mean(ms[1:]) / 20))
