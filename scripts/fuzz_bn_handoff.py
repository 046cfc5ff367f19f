"""Randomised check of the convolution -> batch-norm statistics hand-off: a bf16 8x8 convolution followed by BatchNorm2D(+relu) on its
output against the same batch norm on a COPY of the output (which has no published statistics), for batches on both sides of the
eight-image kernel's threshold and not divisible by 8."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lamp_amd._capi import lib, i64_array; lib.load()
from lamp_amd import sten as S
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(seed)
bad = 0
def T(a): return S.STen.from_numpy(np.ascontiguousarray(a.astype(np.float32)), 0, S.BF16)
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    Ci, Co = int(rng.choice([16, 64, 100, 128])), int(rng.choice([16, 48, 64, 100, 128])); k = int(rng.choice([1, 3])); p = (k - 1) // 2
    N = int(rng.choice([2, 7, 64, 1024, 1030, 1032, 2048]))
    x, w, b = rng.standard_normal((N, Ci, 8, 8)), rng.standard_normal((Co, Ci, k, k)) * 0.2, rng.standard_normal(Co)
    g, be = rng.standard_normal(Co) + 1.5, rng.standard_normal(Co)
    o = C.c_void_p(); lib.lamp_convolution(C.byref(o), T(x), T(w), T(b), i64_array([1, 1]), i64_array([p, p]), i64_array([1, 1]), 2, 0, i64_array([0, 0]), 1)
    Y = S.STen(o); Yc = Y.clone()
    res = []
    for t in (Y, Yc):
        out3 = (C.c_void_p * 3)()
        lib.lamp_native_batch_norm_relu(out3, t, T(g), T(be), T(np.zeros(Co)), T(np.ones(Co)), 1, 0.1, 1e-5)
        res.append([S.STen(h).castToFloat().to_numpy() for h in out3])
    for nm, a, c in zip(("y", "mean", "invstd"), res[0], res[1]):
        err = np.abs(a.astype(np.float64) - c); lim = 2e-2 * (np.abs(c) + np.abs(c).mean() + 1e-30)
        if not (err <= lim).all():
            bad += 1; print("MISMATCH", nm, f"N{N} {Ci}->{Co} k{k}", float(err.max()))
print(f"seed {seed}: {bad} problems")
