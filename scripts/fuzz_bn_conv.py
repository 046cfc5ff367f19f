"""Randomised sweep of convolution(relu(batch_norm(x))) with the batch norm folded into the convolution (lamp_batch_norm_affine +
lamp_convolution_bn_relu_input + _backward) against the chain it replaces (native_batch_norm_relu -> convolution and convolution_backward):
everything bitwise - output, saved and running statistics, the gradient w.r.t. the activation, dweight, dbias.  Half of the cases are
geometries whose kernels fold the table (bf16, N >= 1024, Cin > 32, Cout > 64, 3x3 on 8x8 maps), with ragged batches and channel counts.
The first convolution's statistics hand-off is exercised too: x is the output of a convolution in a third of the cases.
usage: fuzz_bn_conv.py [seed] [iterations]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lamp_amd._capi import lib; lib.load()
from lamp_amd import sten as S
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rng = np.random.default_rng(seed)
bad = folded = 0


def T(a, dt): return S.STen.from_numpy(a.astype(np.float32), 0, dt)
def i64(v): return (C.c_int64 * len(v))(*v)


for it in range(iters):
    fold_case = it % 2 == 0
    if fold_case:
        dt, N, Ci, Co, H, k = S.BF16, int(rng.integers(1024, 2100)), int(rng.integers(33, 129)), int(rng.integers(65, 129)), 8, 3
    else:
        dt = [S.BF16, S.F32][int(rng.integers(0, 2))]
        N, Ci, Co = int(rng.choice([1, 3, 9, 40, 300, 1025])), int(rng.choice([3, 8, 16, 31, 64, 100])), int(rng.choice([2, 16, 40, 64, 100]))
        H, k = int(rng.choice([4, 8, 8, 16])), int(rng.choice([1, 3, 3, 5]))
        if N * max(Ci, Co) * H * H > 30_000_000: N = max(1, 30_000_000 // (max(Ci, Co) * H * H))
    p = k // 2
    one, pad, zero = i64([1, 1]), i64([p, p]), i64([0, 0])
    x = rng.standard_normal((N, Ci, H, H)).astype(np.float32) * 2.0 + 0.3
    X = T(x, dt)
    if it % 3 == 0 and H == 8:                                  # x as a convolution's output: its epilogue hands the statistics over
        w0 = rng.standard_normal((Ci, Ci, 3, 3)).astype(np.float32) * 0.1
        o = C.c_void_p(); lib.lamp_convolution(C.byref(o), X, T(w0, dt), None, one, i64([1, 1]), one, 2, 0, zero, 1)
        X = S.STen(o)
        o = C.c_void_p(); lib.lamp_convolution(C.byref(o), T(x, dt), T(w0, dt), None, one, i64([1, 1]), one, 2, 0, zero, 1)
        X2 = S.STen(o)                                          # a second, identical output for the chain (each hand-off is consumed by its reader)
    else:
        X2 = X
    g, b = rng.standard_normal(Ci).astype(np.float32) * 0.5 + 1.0, rng.standard_normal(Ci).astype(np.float32) * 0.3
    w, cb = rng.standard_normal((Co, Ci, k, k)).astype(np.float32) * 0.2, rng.standard_normal(Co).astype(np.float32)
    gy = rng.standard_normal((N, Co, H, H)).astype(np.float32)
    G, B, Wt, CB, GY = (T(a, dt) for a in (g, b, w, cb, gy))
    z, o1 = np.zeros(Ci, np.float32), np.ones(Ci, np.float32)
    RMc, RVc, RMf, RVf = (T(a, dt) for a in (z, o1, z, o1))
    o3 = (C.c_void_p * 3)(); lib.lamp_native_batch_norm_relu(o3, X2, G, B, RMc, RVc, 1, 0.1, 1e-5)
    act, smc, sic = (S.STen(h) for h in o3)
    o = C.c_void_p(); lib.lamp_convolution(C.byref(o), act, Wt, CB, one, pad, one, 2, 0, zero, 1)
    yc = S.STen(o)
    fl = C.c_int(-1); lib.lamp_convolution_bn_relu_input_folds(C.byref(fl), X, Wt, one, pad, one, 2, 1)
    folded += fl.value
    if fold_case and fl.value != 1:
        bad += 1; print("EXPECTED A FOLDING GEOMETRY", (N, Ci, Co, H, k))
    a3 = (C.c_void_p * 3)(); lib.lamp_batch_norm_affine(a3, X, G, B, RMf, RVf, 0.1, 1e-5)
    aff, smf, sif = (S.STen(h) for h in a3)
    o = C.c_void_p(); lib.lamp_convolution_bn_relu_input(C.byref(o), X, aff, Wt, CB, one, pad, one, 2, 1)
    yf = S.STen(o)
    c3 = (C.c_void_p * 3)(); lib.lamp_convolution_backward(c3, GY, act, Wt, one, pad, one, 2, 0, zero, 1, (C.c_uint8 * 3)(1, 1, 1))
    f3 = (C.c_void_p * 3)(); lib.lamp_convolution_bn_relu_input_backward(f3, GY, X, aff, Wt, one, pad, one, 2, 1, (C.c_uint8 * 3)(1, 1, 1))
    pairs = [(yf, yc, "y"), (smf, smc, "save_mean"), (sif, sic, "save_invstd"), (RMf, RMc, "running_mean"), (RVf, RVc, "running_var")]
    pairs += [(S.STen(f3[i]), S.STen(c3[i]), n_) for i, n_ in enumerate(("d activation", "dweight", "dbias"))]
    for a_, c_, what in pairs:
        if not np.array_equal(a_.to_numpy(), c_.to_numpy(), equal_nan=True):
            d = np.abs(a_.to_numpy().astype(np.float64) - c_.to_numpy().astype(np.float64)).max()
            bad += 1; print("MISMATCH", what, (N, Ci, Co, H, k), dt, "folds" if fl.value else "materialises", float(d))
print(f"seed {seed}: {iters} cases ({folded} folding), {bad} problems")
