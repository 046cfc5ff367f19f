"""End-to-end lamp_amd.umap.umap timing (BASELINE config 5 shape: n x 128 f32 kNN, k = 10, 2-D f64 layout).
usage: python scripts/umap_full_probe.py [n] [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lamp_amd._capi import lib
lib.load()
from lamp_amd import umap as U
import numpy as np
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = np.random.default_rng(0)
data = rng.random((n, 128)) + (np.arange(n) % 16)[:, None]
t0 = time.perf_counter()
layout, b, loss = U.umap(data, precision="f32", k=10, iterations=0)
lib.lamp_device_synchronize()
t1 = time.perf_counter()
layout, b, loss = U.umap(data, precision="f32", k=10, iterations=iters)
lib.lamp_device_synchronize()
t2 = time.perf_counter()
graph_s = t1 - t0
it_ms = ((t2 - t1) - graph_s) / iters * 1e3
print(f"umap n={n}: host->device + kNN graph + distances + edge weights {graph_s:.2f} s ({b.shape[0]} edges); layout {it_ms:.1f} ms/iteration "
      f"-> 500 iterations {graph_s + 0.5 * it_ms:.1f} s total; final loss {loss:.4f}")

# per-kernel-class breakdown of the layout iterations
import ctypes as C
knn = U.knn_search(*(lambda X: (X, X))(__import__("lamp_amd").sten.STen.from_numpy(data.astype(np.float32), 0)), 10)
X64 = __import__("lamp_amd").sten.STen.from_numpy(data, 0)
d = C.c_void_p(); lib.lamp_knn_row_distances(C.byref(d), X64, knn)
ew = U.edge_weights(__import__("lamp_amd").sten.STen(d), knn)
lib.lamp_kernel_timer_enable(1)
U.optimize(ew, n, 0.1, 10, 0.0, 5, 42, True, 1.0, 0, 2)
buf = C.create_string_buffer(1 << 16)
lib.lamp_kernel_timer_report(buf, len(buf))
lib.lamp_kernel_timer_enable(0)
rows = [l.split() for l in buf.value.decode().splitlines()]
rows.sort(key=lambda r: -float(r[2]))
for tag, cnt, ms, fl, by in rows[:12]:
    print(f"   {tag:28s} launches/iter {int(cnt) / 10:6.1f}  ms/iter {float(ms) / 10:8.3f}")
