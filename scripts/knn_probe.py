"""kNN timing: fused top-k-in-GEMM kernel vs the chunked GEMM + top-k path (LAMP_KNN_FUSED=0). usage: knn_probe.py [n] [q] [d] [k]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lamp_amd import sten as S
from lamp_amd._capi import lib
n, q, d, k = (int(a) for a in (sys.argv[1:5] + ["1000000", "131072", "128", "10"][len(sys.argv) - 1:]))
rng = np.random.default_rng(0)
pts = rng.random((n, d), dtype=np.float32) + (np.arange(n) % 16)[:, None].astype(np.float32)
X = S.STen.from_numpy(pts, 0)
Qs = S.STen.from_numpy(pts[:q], 0)
def run():
    i, dd = C.c_void_p(), C.c_void_p()
    lib.lamp_knn_squared_euclidean(C.byref(i), C.byref(dd), X, Qs, k)
    lib.lamp_device_synchronize()
    return S.STen(i), S.STen(dd)
run()
t = time.perf_counter(); I, D = run(); dt = time.perf_counter() - t
fl = 2.0 * n * q * d
print(f"knn n={n} q={q} d={d} k={k} fused={os.environ.get('LAMP_KNN_FUSED', '1')}: {dt * 1e3:.1f} ms  {fl / dt / 1e12:.1f} TFLOP/s  checksum {int(I.to_numpy().sum())} {float(D.to_numpy().sum()):.3f}")
