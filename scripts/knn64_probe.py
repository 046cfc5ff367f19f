import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from lamp_amd import sten as S
from lamp_amd._capi import lib
n, q, d, k = 1000000, 32768, 128, 10
rng = np.random.default_rng(0)
pts = rng.random((n, d)) + (np.arange(n) % 16)[:, None]
X = S.STen.from_numpy(pts, 0, S.F64)
Qs = X.slice(0, 0, q)
def run():
    i = C.c_void_p()
    lib.lamp_knn_squared_euclidean(C.byref(i), None, X, Qs, k)
    lib.lamp_device_synchronize()
run()
t = time.perf_counter(); run(); dt = time.perf_counter() - t
print(f"f64 knn n={n} q={q}: {dt*1e3:.1f} ms  {2.0*n*q*d/dt/1e12:.1f} TFLOP/s -> 1M x 1M in {dt*n/q:.1f} s")
