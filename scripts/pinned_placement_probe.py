"""Does the PCIe gather rate depend on WHERE a pinned buffer landed?  Alternate allocations of the f32 (600 MB) and u8 (150 MB) record sets,
as bench.py --workload epoch does between its variants, and time one epoch of the bare host-resident stream on each."""
import sys, os, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lamp_amd._capi import lib; lib.load()
from lamp_amd import sten as S
from lamp_amd.data import BatchStream
N, B = 50000, 2048
pix = (np.arange(N * 3072, dtype=np.int64) % 251).astype(np.uint8).reshape(N, 3, 32, 32)
f32 = pix.astype(np.float32)
lab = S.STen.from_numpy(np.arange(N, dtype=np.int64), S.CPU)
order = np.random.default_rng(1).permutation(N)
hosts = {"u8": (S.STen.from_numpy(pix, S.CPU), 3072), "f32": (S.STen.from_numpy(f32, S.CPU), 12288)}
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    for name in ("f32", "u8"):
        host, bpp = hosts[name]
        st = BatchStream.minibatchesFromFull(B, False, host, lab, order=order, hostResident=True, outDtype=S.BF16)
        for _ in st: pass
        st.reset(); lib.lamp_device_synchronize()
        t = time.perf_counter()
        for _ in st: pass
        lib.lamp_device_synchronize()
        dt = time.perf_counter() - t
        print(f"round {rnd} {name}: {N * bpp / dt / 1e9:6.1f} GB/s over the bus", flush=True)
        del st; gc.collect()
