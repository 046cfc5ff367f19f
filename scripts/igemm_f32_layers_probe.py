"""Per-layer timing of the f32 matrix-core convolutions (kernels/conv_igemm_f32.hip) at the ResNet's shapes, B = 2048: forward, input
gradient and weight gradient back to back, TFLOP/s against the 157.3 TF f32 MFMA peak."""
import ctypes as C, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lamp_amd._capi import lib, i64_array
lib.load()
from lamp_amd import sten as S
import numpy as np
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
SHAPES = [(16, 128, 3), (128, 128, 3), (128, 100, 3), (100, 100, 3), (16, 128, 1), (128, 100, 1)]
rng = np.random.default_rng(0)
tot = 0.0
def mask3(a, b, c): return (C.c_uint8 * 3)(a, b, c)
for Cin, Cout, k in SHAPES:
    x = S.STen.from_numpy(rng.standard_normal((N, Cin, 8, 8)).astype(np.float32), 0, S.F32)
    gy = S.STen.from_numpy(rng.standard_normal((N, Cout, 8, 8)).astype(np.float32), 0, S.F32)
    w = S.STen.from_numpy((rng.standard_normal((Cout, Cin, k, k)) * 0.05).astype(np.float32), 0, S.F32)
    b = S.STen.zeros([Cout], S.F32, 0)
    p = (k - 1) // 2
    geom = (i64_array([1, 1]), i64_array([p, p]), i64_array([1, 1]), 2, 0, i64_array([0, 0]), 1)
    def fwd():
        o = C.c_void_p()
        lib.lamp_convolution(C.byref(o), x, w, b, *geom)
        S.STen(o).release()
    def bwd(m):
        def f():
            out = (C.c_void_p * 3)()
            lib.lamp_convolution_backward(out, gy, x, w, *geom, m)
            for h in out:
                if h: S.STen(C.c_void_p(h)).release()
            lib.lamp_flush_deferred()
        return f
    fl = 2.0 * N * 64 * Cin * Cout * k * k
    line = f"{Cin:4d}->{Cout:4d} k{k}:"
    for name, fn in (("fprop", fwd), ("dgrad", bwd(mask3(1, 0, 0))), ("wgrad", bwd(mask3(0, 1, 0)))):
        for _ in range(3): fn()
        lib.lamp_device_synchronize()
        R = 20
        t = time.perf_counter()
        for _ in range(R): fn()
        lib.lamp_device_synchronize()
        dt = (time.perf_counter() - t) / R
        tot += dt
        line += f"  {name} {dt*1e6:7.1f} us {fl/dt/1e12:6.1f} TF ({fl/dt/1e12/157.3*100:4.1f} %)"
    print(line)
print(f"sum {tot*1e6:.1f} us")
