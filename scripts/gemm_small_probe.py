"""GEMM shapes of the language-model step at a small token batch (T = batch * 384): time per call under the current dispatch.
Run with LAMP_GEMM_SPLITK=0 to see the no-split choice (one process per setting: the switch is read once)."""
import ctypes as C, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lamp_amd._capi import lib
lib.load()
from lamp_amd import sten as S
import numpy as np
T = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
rng = np.random.default_rng(0)
def mk(r, c): return S.STen.from_numpy(rng.standard_normal((r, c)).astype(np.float32), 0, S.BF16)
def bench(name, f, flops):
    for _ in range(5): f()
    lib.lamp_device_synchronize()
    R = 200
    t = time.perf_counter()
    for _ in range(R): f()
    lib.lamp_device_synchronize()
    dt = (time.perf_counter() - t) / R
    print(f"{name:44s} {dt*1e6:7.1f} us  {flops/dt/1e12:6.0f} TF/s")
for (n, k) in [(768, 768), (3072, 768), (768, 3072), (256, 768)]:
    x, w, p = mk(T, k), mk(k, n), mk(T, n)
    dW, dX = S.STen.zeros([k, n], S.BF16, 0), S.STen.zeros([T, k], S.BF16, 0)
    fl = 2.0 * T * n * k
    def fwd():
        o = C.c_void_p(); lib.lamp_mm(C.byref(o), x, w); S.STen(o).release()
    bench(f"fwd  x[{T},{k}] . W[{k},{n}]", fwd, fl)
    bench(f"dX   p[{T},{n}] . W^T", lambda: S.STen.addmm_out_transposed2(dX, dX, p, w, 0.0, 1.0), fl)
    bench(f"dW   x^T[{k},{T}] . p[{T},{n}]", lambda: S.STen.addmm_out_transposed1(dW, dW, x, p, 0.0, 1.0), fl)
