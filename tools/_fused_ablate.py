#!/usr/bin/env python3
"""Development probe: the fused echo-synthesis + range kernel at the bench shape with its generator (Philox, NZ = 1) against the same kernel
fed an injected spectral noise field from HBM (NZ = 2: no generator VALU, +0.75 GB of reads) -- how much of the launch is the generator?"""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import numpy as np
pkg = importlib.import_module(bench.PKG)
cell = bench.Cell(pkg, 0, 0, 64, 16, 1, inflight=1)
c = cell.ctx
d_w = c.empty((cell.K, cell.Lsym, cell.A))
c.check(c.lib.isac_synth_qpsk_grid_dev(c.handle, C.c_void_p(d_w.ptr), cell.K, cell.Lsym, cell.A, C.c_uint64(77), 0))    # any unit-scale field
c.check(c.lib.isac_profile_enable(c.handle, 1))
for name, kw in (("Philox generator (NZ=1)", dict(seed=5, noise_domain="spectral")), ("injected field (NZ=2)", dict(spectral_noise=d_w))):
    out = []
    for i in range(40):
        pkg.sensing.monoStaticSensing(cell.tx_wave, (cell.K, cell.Lsym, cell.A), cell.carrier, cell.rp, cell.los, nfft=4096, out=cell.echo[0], ctx=c,
                                      fuse_fft2d=(cell.rp, cell.cfar, cell.tx_grid), **kw)
        ms = C.c_double(0.0)
        c.check(c.lib.isac_profile_last_kernel_ms(c.handle, C.byref(ms)))
        out.append(ms.value)
    c.sync()
    print(f"{name:28s} {np.mean(out[20:]):.4f} ms")
