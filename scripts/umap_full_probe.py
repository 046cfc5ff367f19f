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
