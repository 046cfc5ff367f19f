"""Randomised parity sweep of the bf16 convolution paths (narrow MFMA kernels and the 8x8 implicit-GEMM kernels) against ATen-CPU in
f32 on the same bf16 inputs: random batch sizes (ragged for the persistent grids), channel counts, filter sizes, strides, paddings.
  python scripts/fuzz_conv_bf16.py [seed] [iterations]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lamp_amd._capi import lib, i64_array; lib.load()
from lamp_amd import sten as S
import torch
aten = torch.ops.aten
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rng = np.random.default_rng(seed)
bad = 0
def T(t): return S.STen.from_numpy(t.float().numpy(), 0, S.BF16)
def back(t): return torch.from_numpy(t.castToFloat().to_numpy())
def check(name, got, ref, tol):
    global bad
    err = (got.double() - ref.double()).abs(); lim = tol * (ref.double().abs() + ref.double().abs().mean() + 1e-30)
    if not bool((err <= lim).all()):
        bad += 1; i = int((err > lim).flatten().nonzero()[0]); print("MISMATCH", name, "max err", float(err.max()), "first at", i, float(got.flatten()[i]), float(ref.flatten()[i]))
for it in range(iters):
    if it % 2 == 0:   # narrow path
        Ci, Co = int(rng.integers(1, 17)), int(rng.integers(1, 17)); H = int(rng.choice([8, 16, 24, 32])); W = int(rng.choice([8, 16, 24, 32]))
        k = int(rng.choice([1, 3, 5])); st = int(rng.choice([1, 2])); p = int(rng.integers(0, k // 2 + 1)); N = int(rng.choice([1, 2, 3, 5, 17, 300, 1100]))
    else:             # 8x8 implicit GEMM
        Ci, Co = int(rng.choice([8, 16, 24, 64, 100, 128])), int(rng.choice([8, 16, 48, 64, 100, 128])); H = W = 8
        k = int(rng.choice([1, 3])); st = 1; p = (k - 1) // 2; N = int(rng.choice([1, 3, 8, 13, 64, 70, 1030]))
    if (H + 2 * p - k) // st + 1 < 1: continue
    x = (torch.from_numpy(rng.standard_normal((N, Ci, H, W)).astype(np.float32))).bfloat16()
    w = (torch.from_numpy(rng.standard_normal((Co, Ci, k, k)).astype(np.float32)) * 0.3).bfloat16()
    b = (torch.from_numpy(rng.standard_normal(Co).astype(np.float32))).bfloat16()
    args = ([st, st], [p, p], [1, 1], False, [0, 0], 1)
    try:
        ref = aten.convolution(x.float(), w.float(), b.float(), *args)
        o = C.c_void_p(); lib.lamp_convolution(C.byref(o), T(x), T(w), T(b), i64_array([st, st]), i64_array([p, p]), i64_array([1, 1]), 2, 0, i64_array([0, 0]), 1)
        tag = f"N{N} {Ci}->{Co} k{k} s{st} p{p} {H}x{W}"
        check("fwd " + tag, back(S.STen(o)), ref, 1.6e-2)
        gy = (torch.from_numpy(rng.standard_normal(tuple(ref.shape)).astype(np.float32))).bfloat16()
        rb = aten.convolution_backward(gy.float(), x.float(), w.float(), [Co], *args, [True, True, True])
        out3 = (C.c_void_p * 3)(); lib.lamp_convolution_backward(out3, T(gy), T(x), T(w), i64_array([st, st]), i64_array([p, p]), i64_array([1, 1]), 2, 0, i64_array([0, 0]), 1, (C.c_uint8 * 3)(1, 1, 1))
        for nm, h, r in zip(("dx", "dw", "db"), out3, rb):
            check(nm + " " + tag, back(S.STen(h)), r, 3e-2)
    except Exception as e:
        bad += 1; print("EXCEPTION", repr(e)[:300])
print(f"seed {seed}: {iters} iterations, {bad} problems")
