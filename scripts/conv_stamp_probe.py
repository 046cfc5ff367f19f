"""in-kernel phase timing of ig_conv8d (diagnostic build with -DIG8D_STAMP only): prologue / main loop / epilogue cycles per workgroup"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lamp_amd._capi import lib, i64_array; lib.load()
from lamp_amd import sten as S
N, Cin, Cout, k = 2048, int(os.environ.get("CIN", "128")), int(os.environ.get("COUT", "128")), 3
rng = np.random.default_rng(0)
x = S.STen.from_numpy(rng.standard_normal((N, Cin, 8, 8), dtype=np.float32), 0, S.BF16)
w = S.STen.from_numpy(rng.standard_normal((Cout, Cin, k, k), dtype=np.float32) * 0.05, 0, S.BF16)
b = S.STen.from_numpy(np.zeros(Cout, dtype=np.float32), 0, S.BF16)
p = (k - 1) // 2
for it in range(5):
    o = C.c_void_p()
    lib.lamp_convolution(C.byref(o), x, w, b, i64_array([1, 1]), i64_array([p, p]), i64_array([1, 1]), 2, 0, i64_array([0, 0]), 1)
    S.STen(o).release()
lib.lamp_device_synchronize()
buf = (C.c_uint64 * (8 * 512))()
lib.lamp_debug_ig8d_stamps(buf)
a = np.array(buf[:], dtype=np.uint64).reshape(512, 8)[:256].astype(np.int64)
t0 = a[:, 0].min()
names = ["start", "dma issued", "images in LDS", "main loop done", "LDS staged", "stores issued", "end"]
for i in range(1, 7):
    d = a[:, i] - a[:, i - 1]
    print(f"{names[i]:16s} median {np.median(d):9.0f}  min {d.min():9.0f}  max {d.max():9.0f} cycles (100 MHz realtime ticks x ? - see s_memtime)")
print("workgroup start spread", a[:, 0].max() - t0, " total median", np.median(a[:, 6] - a[:, 0]), "last end - first start", a[:, 6].max() - t0)
