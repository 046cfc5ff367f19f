"""Wall-clock TF/s of the bf16 GEMM on uniform random [-1, 1) operands, n^3, the three layouts of lamp's Linear fwd/bwd,
interleaved rounds in ONE process (the harness used for the A/B of kernel variants: set LAMP_GEMM_KERNEL per variant)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lamp_amd._capi import lib
lib.load()
from lamp_amd import sten as S
import numpy as np
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(0)
mk = lambda: S.STen.from_numpy(rng.uniform(-1, 1, (n, n)).astype(np.float32), 0, S.BF16)
A_, W_, P_ = mk(), mk(), mk()
dW, dX = S.STen.zeros([n, n], S.BF16, 0), S.STen.zeros([n, n], S.BF16, 0)
def fwd():
    o = C.c_void_p(); lib.lamp_mm(C.byref(o), A_, W_); return S.STen(o)
ops = {"mm (A kc, B ks)": fwd,
       "A^T.p (ks, ks)": lambda: S.STen.addmm_out_transposed1(dW, dW, A_, P_, 0.0, 1.0),
       "p.W^T (kc, kc)": lambda: S.STen.addmm_out_transposed2(dX, dX, P_, W_, 0.0, 1.0)}
variants = sys.argv[2:] or ["default"]
res = {}
for rnd in range(3):
    for v in variants:
        os.environ["LAMP_GEMM_KERNEL"] = v
        for name, f in ops.items():
            for _ in range(3): f()
            lib.lamp_device_synchronize()
            t0 = time.perf_counter(); R = 20
            for _ in range(R): f()
            lib.lamp_device_synchronize()
            dt = (time.perf_counter() - t0) / R
            res.setdefault((v, name), []).append(2.0 * n ** 3 / dt / 1e12)
for k, v in res.items():
    print(f"{k[0]:5s} {k[1]:18s} TF/s rounds: " + " ".join(f"{x:7.1f}" for x in v))
