"""Randomised check of the batch-norm backward (plain, +relu, +add+relu; bf16) against f32 arithmetic on the same bf16 inputs (ATen CPU),
over sizes on both sides of the one-pass kernel's threshold (bn_bwd_fused_kernel: activations of ~10 MB and more), odd batch sizes and
maps whose packets per row are not a power of two; and of the convolution input gradient accumulated in the dgrad epilogue
(lamp_convolution_backward_input_add) against the dgrad + add chain, bitwise.  usage: fuzz_bn_backward.py [seed] [iterations]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lamp_amd._capi import lib, i64_array; lib.load()
from lamp_amd import sten as S
aten = torch.ops.aten
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = np.random.default_rng(seed)
gen = torch.Generator().manual_seed(seed)
bad = 0


def T(t):
    return S.STen.from_numpy(np.ascontiguousarray(t.float().numpy()), 0, S.BF16)


def back(t):                                                 # t: an STen (a raw handle must be wrapped exactly once: the wrapper owns it)
    return torch.from_numpy(np.ascontiguousarray(t.castToFloat().to_numpy()))


def close(got, want, what, tag, rtol=2 ** -6):
    global bad
    err = (got.double() - want.double()).abs()
    lim = rtol * (want.double().abs() + want.double().abs().mean() + 1e-30)
    if not bool((err <= lim).all()):
        bad += 1
        print("MISMATCH", what, tag, float(err.max()), flush=True)


for it in range(iters):
    variant = int(rng.integers(0, 3))
    C_, H = [(128, 8), (100, 8), (64, 8), (16, 16), (6, 32), (5, 20), (37, 12), (130, 8)][int(rng.integers(0, 8))]
    N = int(rng.choice([3, 64, 500, 1024, 2048, 2051, 2100]))
    if N * C_ * H * H > 2100 * 128 * 64:
        N = 2048
    shape = (N, C_, H, H)
    bf = torch.bfloat16
    x = (torch.randn(shape, generator=gen) * 2 + 0.3).to(bf)
    ad = torch.randn(shape, generator=gen).to(bf)
    gy = torch.randn(shape, generator=gen).to(bf)
    w = (torch.randn(C_, generator=gen) + 1.5).to(bf); b = torch.randn(C_, generator=gen).to(bf)
    X, AD, GY, Wt, Bt = T(x), T(ad), T(gy), T(w), T(b)
    RM, RV = T(torch.zeros(C_)), T(torch.ones(C_))
    fwd = (C.c_void_p * 3)()
    lib.lamp_native_batch_norm(fwd, X, Wt, Bt, RM, RV, 1, 0.1, 1e-5)
    sm, si = S.STen(fwd[1]), S.STen(fwd[2]); _y = S.STen(fwd[0])
    if variant == 2:
        out = (C.c_void_p * 4)()
        lib.lamp_native_batch_norm_add_relu_backward(out, GY, X, AD, Wt, Bt, RM, RV, sm, si, 1, 1e-5, (C.c_uint8 * 4)(1, 1, 1, 1))
    else:
        out = (C.c_void_p * 3)()
        if variant == 1:
            lib.lamp_native_batch_norm_relu_backward(out, GY, X, Wt, Bt, RM, RV, sm, si, 1, 1e-5, (C.c_uint8 * 3)(1, 1, 1))
        else:
            lib.lamp_native_batch_norm_backward(out, GY, X, Wt, RM, RV, sm, si, 1, 1e-5, (C.c_uint8 * 3)(1, 1, 1))
    got = [back(S.STen(out[i])) for i in range(len(out))]
    mean, invstd = back(sm), back(si)
    xf, g = x.float(), gy.float()
    if variant >= 1:
        pre = ((xf - mean.view(1, -1, 1, 1)) * (invstd * w.float()).view(1, -1, 1, 1) + b.float().view(1, -1, 1, 1)).to(bf)
        if variant == 2:
            pre = (pre.float() + ad.float()).to(bf)
        g = torch.where(pre.float() < 0, torch.zeros_like(g), g)
    ref = aten.native_batch_norm_backward(g, xf, w.float(), None, None, mean, invstd, True, 1e-5, [True, True, True])
    tag = f"N{N} C{C_} H{H} variant{variant}"
    for a, r, nm in zip(got, ref, ("dx", "dweight", "dbias")):
        close(a, r, nm, tag)
    if variant == 2 and not torch.equal(got[3], g):
        bad += 1; print("MISMATCH daddend", tag, flush=True)

for it in range(iters):
    Ci, Co = int(rng.choice([16, 64, 100, 128])), int(rng.choice([16, 48, 100, 128])); k = int(rng.choice([1, 3])); p = (k - 1) // 2
    N = int(rng.choice([7, 64, 1024, 1029, 2048]))
    x = torch.randn((N, Ci, 8, 8), generator=gen).to(torch.bfloat16)
    w = (torch.randn((Co, Ci, k, k), generator=gen) * 0.2).to(torch.bfloat16)
    gy = torch.randn((N, Co, 8, 8), generator=gen).to(torch.bfloat16)
    ad = torch.randn((N, Ci, 8, 8), generator=gen).to(torch.bfloat16)
    X, Wt, GY, AD = T(x), T(w), T(gy), T(ad)
    geom = (i64_array([1, 1]), i64_array([p, p]), i64_array([1, 1]), 2)
    out3 = (C.c_void_p * 3)()
    lib.lamp_convolution_backward(out3, GY, X, Wt, *geom, 0, i64_array([0, 0]), 1, (C.c_uint8 * 3)(1, 0, 0))
    dx = S.STen(out3[0])
    chain = C.c_void_p(); lib.lamp_add(C.byref(chain), AD, dx, 1.0)
    o = C.c_void_p(); lib.lamp_convolution_backward_input_add(C.byref(o), GY, X, Wt, *geom, i64_array([0, 0]), 1, AD)
    if not torch.equal(back(S.STen(o)), back(S.STen(chain))):
        bad += 1; print("MISMATCH dgrad+add", f"N{N} {Ci}->{Co} k{k}", flush=True)
print(f"seed {seed}: {bad} problems")
