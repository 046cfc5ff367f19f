"""End-to-end lamp_amd.umap timing (BASELINE config 5 shape: n x 128 f32 kNN, k = 10, 2-D f64 layout), stage by stage.
usage: python scripts/umap_full_probe.py [n] [iterations] [f32|f64]   (kNN precision; lamp's default is DoublePrecision = f64)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lamp_amd._capi import lib
lib.load()
from lamp_amd import umap as U, sten as S
import numpy as np
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
prec = sys.argv[3] if len(sys.argv) > 3 else "f32"
rng = np.random.default_rng(0)
data = rng.random((n, 128)) + (np.arange(n) % 16)[:, None]


def stage(f):
    lib.lamp_device_synchronize(); t = time.perf_counter(); r = f(); lib.lamp_device_synchronize(); return r, time.perf_counter() - t


X, t_up = stage(lambda: S.STen.from_numpy(data.astype(np.float32), 0, S.F32) if prec == "f32" else S.STen.from_numpy(data, 0, S.F64))
U.knn_search(X, X.slice(0, 0, min(n, 4096)), 10)                    # warm-up (code objects, allocator)
knn, t_knn = stage(lambda: U.knn_search(X, X, 10))
X64, t_up64 = stage(lambda: S.STen.from_numpy(data, 0, S.F64))
def _dist():
    d = C.c_void_p(); lib.lamp_knn_row_distances(C.byref(d), X64, knn); return S.STen(d)
dist, t_dist = stage(_dist)
ew, t_ew = stage(lambda: U.edge_weights(dist, knn))
U.optimize(ew, n, 0.1, 2, 0.0, 5, 42, True, 1.0, 0, 2)
(layout, loss), t_opt = stage(lambda: U.optimize(ew, n, 0.1, iters, 0.0, 5, 42, True, 1.0, 0, 2))
it_ms = t_opt / iters * 1e3
total = t_up + t_knn + t_up64 + t_dist + t_ew + 0.5 * it_ms
print(f"umap n={n} kNN in {prec}: upload {t_up:.2f} s, kNN {t_knn:.2f} s ({2.0 * n * n * 128 / t_knn / 1e12:.1f} TFLOP/s), upload f64 {t_up64:.2f} s, "
      f"row distances {t_dist * 1e3:.0f} ms, edge weights {t_ew * 1e3:.0f} ms ({ew.shape[0]} edges), layout {it_ms:.2f} ms/iteration "
      f"-> 500 iterations {total:.1f} s total; loss after {iters}: {loss:.4f}")
