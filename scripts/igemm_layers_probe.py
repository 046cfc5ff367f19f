"""Back-to-back timing of the implicit-GEMM layer shapes of the CIFAR ResNet step (fprop; dgrad = the same kernel with the channel
counts swapped).  Run once per LAMP_IG_VARIANT to compare kernel variants on one device."""
import ctypes as C, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lamp_amd._capi import lib, i64_array
lib.load()
from lamp_amd import sten as S
import numpy as np
N = 2048
SHAPES = [(16, 128, 3), (128, 128, 3), (128, 100, 3), (100, 100, 3), (100, 128, 3), (16, 16, 3), (128, 16, 3), (16, 128, 1), (128, 16, 1), (128, 100, 1), (100, 128, 1)]
rng = np.random.default_rng(0)
tot = 0.0
for Cin, Cout, k in SHAPES:
    x = S.STen.from_numpy(rng.standard_normal((N, Cin, 8, 8)).astype(np.float32), 0, S.BF16)
    w = S.STen.from_numpy((rng.standard_normal((Cout, Cin, k, k)) * 0.05).astype(np.float32), 0, S.BF16)
    b = S.STen.zeros([Cout], S.BF16, 0)
    p = (k - 1) // 2
    def run():
        o = C.c_void_p()
        lib.lamp_convolution(C.byref(o), x, w, b, i64_array([1, 1]), i64_array([p, p]), i64_array([1, 1]), 2, 0, i64_array([0, 0]), 1)
        S.STen(o).release()
    for _ in range(5): run()
    lib.lamp_device_synchronize()
    R = 100
    t = time.perf_counter()
    for _ in range(R): run()
    lib.lamp_device_synchronize()
    dt = (time.perf_counter() - t) / R
    fl = 2.0 * N * 64 * Cin * Cout * k * k
    by = N * 64 * (Cin + Cout) * 2
    tot += dt
    print(f"variant={os.environ.get('LAMP_IG_VARIANT','default'):8s} {Cin:4d}->{Cout:4d} k{k}: {dt*1e6:6.1f} us  {fl/dt/1e12:6.0f} TF/s  {by/dt/1e12:5.2f} TB/s")
print(f"sum {tot*1e6:.1f} us")
