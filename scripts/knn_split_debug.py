import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
from lamp_amd._capi import lib; lib.load()
from lamp_amd import sten as S
n, d, k, nq = int(sys.argv[1]), 128, 10, int(sys.argv[2])
rng = np.random.default_rng(5)
pts = rng.random((n, d), dtype=np.float32) + (np.arange(n) % 16)[:, None].astype(np.float32)
data = S.STen.from_numpy(pts, 0, S.F32)
rows = np.sort(rng.choice(n, nq, replace=False))
query = S.STen.from_numpy(pts[rows], 0, S.F32)
def run(mode):
    lib.lamp_knn_split_mode(mode)
    i, dd = C.c_void_p(), C.c_void_p()
    lib.lamp_knn_squared_euclidean(C.byref(i), C.byref(dd), data, query, k)
    return S.STen(i).to_numpy(), S.STen(dd).to_numpy()
ei, ed = run(0)
si, sd = run(2)
f = C.c_int64(); lib.lamp_knn_split_last_failed(C.byref(f)); pl = C.c_int(); lib.lamp_knn_split_last_planes(C.byref(pl))
self_e = (ei == rows[:, None]).any(1); self_s = (si == rows[:, None]).any(1)
print("planes", pl.value, "failed", f.value, "self present exact", self_e.mean(), "split", self_s.mean())
bad = np.where(~self_s)[0]
print("rows without self:", len(bad), bad[:10])
for b in bad[:3]:
    print(" row", b, "query id", rows[b]); print("  exact", ei[b], ed[b]); print("  split", si[b], sd[b])
same = (np.sort(ei, 1) == np.sort(si, 1)).all(1)
print("same sets", same.mean(), " idx range", si.min(), si.max())
