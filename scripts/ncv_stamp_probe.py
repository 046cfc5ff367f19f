"""in-kernel phase timing of ncv_fwd2 (diagnostic build with EXTRA=-DNCV_STAMP only): cycles per phase of the first image of every workgroup"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lamp_amd._capi import lib, i64_array; lib.load()
from lamp_amd import sten as S
N = 2048
rng = np.random.default_rng(0)
LAYERS = [("stem 3->6 5x5 32x32", 3, 6, 5, 32, 1, 2), ("res1.c1 6->6 3x3 s2 32x32", 6, 6, 3, 32, 2, 1), ("res1.c2 6->6 3x3 16x16", 6, 6, 3, 16, 1, 1),
          ("res1.left 6->6 1x1 s2 32x32", 6, 6, 1, 32, 2, 0), ("res2.c1 6->16 3x3 s2 16x16", 6, 16, 3, 16, 2, 1), ("res2.left 6->16 1x1 s2 16x16", 6, 16, 1, 16, 2, 0)]
names = ["start", "weights in regs", "LDS zeroed+barrier", "image staged", "computed+stores issued", "all images done", "stores drained"]
for name, Cin, Cout, k, H, s, p in LAYERS:
    x = S.STen.from_numpy(rng.standard_normal((N, Cin, H, H), dtype=np.float32), 0, S.BF16)
    w = S.STen.from_numpy(rng.standard_normal((Cout, Cin, k, k), dtype=np.float32) * 0.05, 0, S.BF16)
    b = S.STen.from_numpy(np.zeros(Cout, dtype=np.float32), 0, S.BF16)
    for it in range(4):
        o = C.c_void_p()
        lib.lamp_convolution(C.byref(o), x, w, b, i64_array([s, s]), i64_array([p, p]), i64_array([1, 1]), 2, 0, i64_array([0, 0]), 1)
        S.STen(o).release()
    lib.lamp_device_synchronize()
    buf = (C.c_uint64 * (8 * 1024))()
    lib.lamp_debug_ncv_stamps(buf)
    a = np.array(buf[:], dtype=np.uint64).reshape(1024, 8).astype(np.int64)
    t0 = a[:, 0].min()
    print(f"== {name}: in {N*Cin*H*H*2/1e6:.1f} MB out {N*Cout*(H//s)**2*2/1e6:.1f} MB")
    for i in range(1, 7):
        d = a[:, i] - a[:, i - 1]
        print(f"   {names[i]:24s} median {np.median(d):8.0f}  min {d.min():8.0f}  max {d.max():8.0f}")
    print(f"   start spread {a[:, 0].max() - t0}  total median {np.median(a[:, 6] - a[:, 0]):.0f}  last end - first start {a[:, 6].max() - t0} cycles")
