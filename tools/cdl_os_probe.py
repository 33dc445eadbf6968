#!/usr/bin/env python3
"""Development probe: one config-5 downlink batch (8 slots x 5 UEs on 64 antennas, T = 61 909, CDL-A or CDL-D) timed with the device to itself -- the overlap-save path
(default) or the time-domain kernels (ISAC_CDL_TIME_DOMAIN=1); under rocprofv3 --kernel-trace the per-kernel split.   python tools/cdl_os_probe.py [CDL-A|CDL-D] [reps]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
pkg = importlib.import_module(bench.PKG)
ctx = pkg.Context(0)
CM = pkg.communication.channelModels
prof = sys.argv[1] if len(sys.argv) > 1 else "CDL-A"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
T, nt, n_slots, n_ue = 61909, 64, 8, 5
rng = np.random.default_rng(0)
waves = [ctx.to_device(np.asfortranarray(rng.standard_normal((T, nt)) + 1j * rng.standard_normal((T, nt)))) for _ in range(n_slots)]
chans = [CM.CDLChannel(DelayProfile=prof, TransmitAntennaArraySize=(4, 8, 2, 1, 1), Seed=73) for _ in range(n_ue)]
outs = [ctx.empty((T, 2)) for _ in range(n_slots * n_ue)]
st = chans[0]._static()
gains = ctx.empty((n_slots * n_ue * 4 * st.base.shape[0] * st.base.shape[2] * st.base.shape[3],))
def call():
    CM.applyCDLBatch([chans[u] for s_ in range(n_slots) for u in range(n_ue)], [waves[s_] for s_ in range(n_slots) for u in range(n_ue)], ctx=ctx, outs=outs, gains=gains)
call(); ctx.sync()
ctx.timer_start()
for _ in range(reps):
    call()
ms = ctx.timer_stop_ms() / reps
print(f"{prof}: {n_slots * n_ue}-job downlink batch, {'time-domain kernels' if os.environ.get('ISAC_CDL_TIME_DOMAIN') else 'overlap-save'}: {ms:.3f} ms per call = {1e3 * ms / (n_slots * n_ue):.1f} us per job")
