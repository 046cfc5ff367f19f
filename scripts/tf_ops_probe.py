"""Bandwidth check of the transformer-path elementwise / normalisation ops (bf16): achieved GB/s vs algorithmic bytes."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lamp_amd import sten as S
from lamp_amd._capi import lib, i64_array

rng = np.random.default_rng(0)
R, D = 16384, 4096
x = S.STen.from_numpy(rng.standard_normal((R, D), dtype=np.float32), 0, S.BF16)
g = S.STen.from_numpy(rng.standard_normal((R, D), dtype=np.float32), 0, S.BF16)
w = S.STen.ones([D], S.BF16, 0); b = S.STen.zeros([D], S.BF16, 0)


def timeit(fn, n=20):
    fn(); lib.lamp_device_synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    lib.lamp_device_synchronize()
    return (time.perf_counter() - t) / n


def ln_fwd():
    o = (C.c_void_p * 3)(); lib.lamp_native_layer_norm(o, x, i64_array([D]), 1, w, b, 1e-5); return [S.STen(h) for h in o]
y, mean, rstd = ln_fwd()
def ln_bwd():
    o = (C.c_void_p * 3)(); lib.lamp_native_layer_norm_backward(o, g, x, i64_array([D]), 1, mean, rstd, w, b, (C.c_uint8 * 3)(1, 1, 1)); return [S.STen(h) for h in o if h]
def gelu():
    o = C.c_void_p(); lib.lamp_gelu(C.byref(o), x); return S.STen(o)
def gelu_b():
    o = C.c_void_p(); lib.lamp_gelu_backward(C.byref(o), g, x); return S.STen(o)
def lsm():
    o = C.c_void_p(); lib.lamp_log_softmax(C.byref(o), x, 1); return S.STen(o)
nb = R * D * 2
V, E, T = 50304, 768, 65536
emb = S.STen.from_numpy(rng.standard_normal((V, E), dtype=np.float32), 0, S.BF16)
idx = S.STen.from_numpy(rng.integers(0, V, T).astype(np.int64), 0)
def embed():
    o = C.c_void_p(); lib.lamp_embedding(C.byref(o), emb, idx); return S.STen(o)

Vb = S.STen.from_numpy(rng.standard_normal((4096, V), dtype=np.float32), 0, S.BF16)
def lsm_vocab():
    o = C.c_void_p(); lib.lamp_log_softmax(C.byref(o), Vb, 1); return S.STen(o)
LSo = lsm(); 
def lsm_b():
    o = C.c_void_p(); lib.lamp_log_softmax_backward_data(C.byref(o), g, LSo, 1); return S.STen(o)
for name, fn, byts in (("log_softmax fwd (4096 x 50304)", lsm_vocab, 2 * 4096 * V * 2), ("log_softmax bwd (rows of 4096)", lsm_b, 3 * nb), ("layer_norm fwd", ln_fwd, 2 * nb), ("layer_norm bwd", ln_bwd, 3 * nb), ("gelu fwd", gelu, 2 * nb), ("gelu bwd", gelu_b, 3 * nb),
                       ("log_softmax fwd (rows of 4096)", lsm, 2 * nb), ("embedding 65536 x 768", embed, 2 * T * E * 2)):
    dt = timeit(fn)
    print(f"{name:34s} {dt * 1e6:9.1f} us  {byts / dt / 1e9:8.0f} GB/s")
