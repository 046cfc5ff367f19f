"""Throughput probe for BASELINE config 5 (lamp-umap): kNN graph on n x 128 f32 points (k = 10) and the fused UMAP layout
loss+gradient, at a size that runs in seconds.  usage: python scripts/umap_probe.py [n_points] [n_queries]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lamp_amd._capi import lib
lib.load()
from lamp_amd import sten as S
import numpy as np
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
d, k = 128, 10
rng = np.random.default_rng(0)
pts = (rng.random((n, d), dtype=np.float32) + (np.arange(n) % 16)[:, None].astype(np.float32))
data = S.STen.from_numpy(pts, 0, S.F32)
query = S.STen.from_numpy(pts[:nq].copy(), 0, S.F32)
def knn():
    i, dd = C.c_void_p(), C.c_void_p()
    lib.lamp_knn_squared_euclidean(C.byref(i), C.byref(dd), data, query, k)
    return S.STen(i), S.STen(dd)
I, D = knn(); lib.lamp_device_synchronize()
t0 = time.perf_counter(); R = 3
for _ in range(R): I, D = knn()
lib.lamp_device_synchronize()
dt = (time.perf_counter() - t0) / R
idx = I.to_numpy()
assert (idx[:, :] == np.arange(nq)[:, None]).any(1).all(), "self must be among the neighbours"
print(f"kNN n={n} q={nq} d={d} k={k} f32: {dt*1e3:.1f} ms  {nq/dt:,.0f} queries/s  {2.0*nq*n*d/dt/1e12:.1f} TFLOP/s  (1M x 1M would take {dt*(1e6/nq)*(1e6/n):.0f} s)")

# UMAP layout: E+ = n*k attractive pairs, 5 negatives each, 2-D fp64 locations
ne = nq * k
loc = S.STen.from_numpy(rng.random((nq, 2)), 0, S.F64)
i1 = S.STen.from_numpy(np.repeat(np.arange(nq), k).astype(np.int64), 0)
i2 = S.STen.from_numpy(idx.reshape(-1).astype(np.int64) % nq, 0)
b = S.STen.from_numpy(rng.random(ne), 0, S.F64)
i3 = S.STen.from_numpy(rng.integers(0, nq, ne * 5).astype(np.int64), 0)
i4 = S.STen.from_numpy(rng.integers(0, nq, ne * 5).astype(np.int64), 0)
grad = S.STen.zeros([nq, 2], S.F64, 0)
w = (C.c_double * 4)(1.0, 2.0, 4.0, 8.0)
def it():
    o = C.c_void_p()
    lib.lamp_umap_loss_grad(C.byref(o), grad, loc, i1, i2, b, i3, i4, 0.0, 1, 1.0, w)
    return S.STen(o)
it(); lib.lamp_device_synchronize()
t0 = time.perf_counter(); R = 20
for _ in range(R): it()
lib.lamp_device_synchronize()
dt = (time.perf_counter() - t0) / R
pairs = ne * 6
print(f"UMAP loss+grad points={nq} pairs={pairs}: {dt*1e3:.3f} ms/iteration  {pairs/dt/1e9:.2f} Gpairs/s  {pairs*(2*2*8+2*2*8)/dt/1e9:.0f} GB/s gather+scatter")

# edge weights (umap.scala:50-113) on a 1M x 10 neighbour table
ne_pts = 1_000_000
kd = np.sort(rng.random((ne_pts, k)), axis=1); kd[:, 0] = 0.0
ki = rng.integers(0, ne_pts, (ne_pts, k)).astype(np.int64); ki[:, 0] = np.arange(ne_pts)
KD, KI = S.STen.from_numpy(kd, 0, S.F64), S.STen.from_numpy(ki, 0)
def ew():
    o = C.c_void_p(); lib.lamp_umap_edge_weights(C.byref(o), KD, KI); return S.STen(o)
ew(); lib.lamp_device_synchronize()
t0 = time.perf_counter(); R = 5
for _ in range(R): out = ew()
lib.lamp_device_synchronize()
dt = (time.perf_counter() - t0) / R
print(f"UMAP edge weights n={ne_pts} k={k}: {dt*1e3:.2f} ms ({out.shape[0]} edges)")
