"""host-resident minibatch stream alone (no training): gathered bytes per second over PCIe, per stored type"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lamp_amd._capi import lib; lib.load()
from lamp_amd import sten as S
from lamp_amd.data import BatchStream
N, B = 50000, 2048
pix = (np.arange(N * 3072, dtype=np.int64) % 251).astype(np.uint8).reshape(N, 3, 32, 32)
lab = S.STen.from_numpy(np.arange(N, dtype=np.int64), S.CPU)
order = np.random.default_rng(1).permutation(N)
for name, host, odt, bpp in (("f32 -> bf16", pix.astype(np.float32), S.BF16, 12288), ("f32 -> f32", pix.astype(np.float32), -1, 12288), ("u8 -> bf16", pix, S.BF16, 3072)):
    st = BatchStream.minibatchesFromFull(B, False, S.STen.from_numpy(host, S.CPU), lab, order=order, hostResident=True, outDtype=odt)
    for _ in st: pass
    st.reset(); lib.lamp_device_synchronize()
    t = time.perf_counter()
    for _ in st: pass
    lib.lamp_device_synchronize()
    dt = time.perf_counter() - t
    print(f"{name}: {N / dt:10.0f} records/s, {N * bpp / dt / 1e9:6.1f} GB/s over the bus")
