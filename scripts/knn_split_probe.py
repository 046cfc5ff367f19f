import sys, os, time, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from lamp_amd._capi import lib; lib.load()
from lamp_amd import sten as S
n = int(sys.argv[1]); nq = int(sys.argv[2]); kind = sys.argv[3] if len(sys.argv) > 3 else "normal"
prec = sys.argv[4] if len(sys.argv) > 4 else "f32"
DT = S.F64 if prec == "f64" else S.F32
g = torch.Generator().manual_seed(1)
if kind == "normal": data = torch.randn(n, 128, generator=g)
else: data = torch.rand(n, 128, generator=g) + (torch.arange(n) % 16).float().reshape(n, 1)      # bench.py's kNN points: 16 clusters, far from the origin
D = S.STen.from_numpy(data.numpy(), 0, DT)
Qt = D if nq == n else S.STen.from_numpy(data[:nq].numpy().copy(), 0, DT)
def run(mode):
    lib.lamp_knn_split_mode(mode)
    i, d = C.c_void_p(), C.c_void_p()
    lib.lamp_knn_squared_euclidean(C.byref(i), C.byref(d), D, Qt, 10)
    lib.lamp_device_synchronize(0) if hasattr(lib, "lamp_device_synchronize") else None
    t = time.time()
    i, d = C.c_void_p(), C.c_void_p()
    lib.lamp_knn_squared_euclidean(C.byref(i), C.byref(d), D, Qt, 10)
    I = S.STen(i).to_numpy(); Dd = S.STen(d).to_numpy()
    return time.time() - t, I, Dd
t0, i0, d0 = run(0)
t2, i2, d2 = run(2)
f = C.c_int64(); lib.lamp_knn_split_last_failed(C.byref(f))
pl = C.c_int(); lib.lamp_knn_split_last_planes(C.byref(pl))
same = (np.sort(i0, 1) == np.sort(i2, 1)).all(1)
print(f"{prec} {kind} planes {pl.value}  n {n} nq {nq}: exact {t0:.3f} s  split {t2:.3f} s  failed {f.value}  rows with the same set {same.mean():.6f}  max |dd| {np.abs(d0 - d2).max():.3e}")
