"""Randomised parity sweep (not part of the test suite): element-wise ops with broadcasting and strided views, reductions over random
dimension sets, index ops, small convolutions and batch norms against ATen-CPU in f32 / f64 on random shapes.  Prints every mismatch.
  python scripts/fuzz_parity.py [seed] [iterations]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lamp_amd._capi import lib, i64_array; lib.load()
from lamp_amd import sten as S
import torch
aten = torch.ops.aten
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.default_rng(seed)
bad = 0

def T(a, dt=None):
    return S.STen.from_numpy(np.ascontiguousarray(a), 0, dt)
def back(t):
    return torch.from_numpy(t.to_numpy())
def check(name, got, ref, tol):
    global bad
    g, r = got.double(), ref.double()
    if g.shape != r.shape:
        bad += 1; print("SHAPE", name, tuple(g.shape), tuple(r.shape)); return
    if r.numel() == 0: return
    err = (g - r).abs()
    lim = tol * (r.abs() + r.abs().mean() + 1e-30)
    both_nan = torch.isnan(g) & torch.isnan(r)
    ok = (err <= lim) | both_nan | ((g == r))
    if not bool(ok.all()):
        bad += 1; i = int((~ok).flatten().nonzero()[0]); print("MISMATCH", name, "max err", float(err[~both_nan].max()), "at", i, float(g.flatten()[i]), float(r.flatten()[i]))

def rshape(maxd=4, maxn=9):
    return [int(rng.integers(1, maxn)) for _ in range(int(rng.integers(1, maxd + 1)))]
def bshape(s):     # a shape broadcastable to s
    k = int(rng.integers(0, len(s) + 1))
    return [1 if rng.random() < 0.4 else d for d in s[len(s) - k:]] if k else []

for it in range(iters):
    f64 = rng.random() < 0.5
    npdt, tdt, sdt, tol = (np.float64, torch.float64, S.F64, 1e-12) if f64 else (np.float32, torch.float32, S.F32, 1e-5)
    kind = it % 6
    try:
        if kind == 0:      # binary with broadcasting
            s = rshape(); a = rng.standard_normal(s).astype(npdt); b = np.asarray(rng.standard_normal(bshape(s)) + 2.5).astype(npdt)
            A_, B_ = T(a, sdt), T(b, sdt); ta, tb = torch.from_numpy(a), torch.from_numpy(b)
            for nm, f, tf in (("add", lambda: A_ + B_, ta + tb), ("sub", lambda: A_ - B_, ta - tb), ("mul", lambda: A_ * B_, ta * tb), ("div", lambda: A_ / B_, ta / tb)):
                check(f"{nm} {s} {list(b.shape)}", back(f()), tf, tol)
        elif kind == 1:    # unary on a transposed / sliced view
            s = rshape(3, 12); s = s if len(s) >= 2 else s + [5]
            a = rng.standard_normal(s).astype(npdt); A_ = T(a, sdt).transpose(0, len(s) - 1); ta = torch.from_numpy(a).transpose(0, len(s) - 1)
            for nm in ("exp", "tanh", "sigmoid", "relu", "gelu", "sin", "cos"):
                ref = {"gelu": lambda t: torch.nn.functional.gelu(t), "relu": torch.relu, "sigmoid": torch.sigmoid}.get(nm, getattr(torch, nm, None))(ta)
                check(f"{nm} view {s}", back(getattr(A_, nm)()), ref, tol * 10)
        elif kind == 2:    # reductions
            s = rshape(4, 10); a = rng.standard_normal(s).astype(npdt); A_ = T(a, sdt); ta = torch.from_numpy(a)
            dims = sorted(set(int(x) for x in rng.integers(0, len(s), int(rng.integers(1, len(s) + 1))))); keep = bool(rng.random() < 0.5)
            check(f"sum {s} {dims} {keep}", back(A_.sum(dims, keep)), ta.sum(dims, keep), tol * 10)
            check(f"mean {s} {dims} {keep}", back(A_.mean(dims, keep)), ta.mean(dims, keep), tol * 10)
        elif kind == 3:    # index_select / cat
            s = rshape(3, 9); a = rng.standard_normal(s).astype(npdt); d = int(rng.integers(0, len(s)))
            idx = rng.integers(0, s[d], int(rng.integers(1, 12))).astype(np.int64)
            A_ = T(a, sdt); ta = torch.from_numpy(a)
            check(f"index_select {s} {d}", back(A_.indexSelect(d, T(idx))), ta.index_select(d, torch.from_numpy(idx)), 0.0)
            check(f"cat {s} {d}", back(S.STen.cat([A_, A_ * 2.0], d)), torch.cat([ta, ta * 2], d), tol)
        elif kind == 4:    # small convolution forward + backward
            N, Ci, Co = int(rng.integers(1, 4)), int(rng.integers(1, 7)), int(rng.integers(1, 7)); H, W = int(rng.integers(3, 11)), int(rng.integers(3, 11))
            k = int(rng.choice([1, 2, 3])); st = int(rng.choice([1, 2])); p = int(rng.integers(0, k)); 
            x = rng.standard_normal((N, Ci, H, W)).astype(npdt); w = rng.standard_normal((Co, Ci, k, k)).astype(npdt); b = rng.standard_normal(Co).astype(npdt)
            tx, tw, tb = map(torch.from_numpy, (x, w, b))
            ref = aten.convolution(tx, tw, tb, [st, st], [p, p], [1, 1], False, [0, 0], 1)
            o = C.c_void_p(); lib.lamp_convolution(C.byref(o), T(x, sdt), T(w, sdt), T(b, sdt), i64_array([st, st]), i64_array([p, p]), i64_array([1, 1]), 2, 0, i64_array([0, 0]), 1)
            check(f"conv {x.shape} {w.shape} s{st} p{p}", back(S.STen(o)), ref, tol * 20)
            gy = rng.standard_normal(tuple(ref.shape)).astype(npdt)
            rb = aten.convolution_backward(torch.from_numpy(gy), tx, tw, [Co], [st, st], [p, p], [1, 1], False, [0, 0], 1, [True, True, True])
            out3 = (C.c_void_p * 3)(); lib.lamp_convolution_backward(out3, T(gy, sdt), T(x, sdt), T(w, sdt), i64_array([st, st]), i64_array([p, p]), i64_array([1, 1]), 2, 0, i64_array([0, 0]), 1, (C.c_uint8 * 3)(1, 1, 1))
            for nm, h, r in zip(("dx", "dw", "db"), out3, rb):
                check(f"conv {nm} {x.shape} {w.shape} s{st} p{p}", back(S.STen(h)), r, tol * 50)
        else:              # batch norm training forward
            N, Cc, H = int(rng.integers(2, 6)), int(rng.integers(1, 6)), int(rng.integers(1, 7))
            x = rng.standard_normal((N, Cc, H, H)).astype(npdt); g = (rng.standard_normal(Cc) + 1.5).astype(npdt); b = rng.standard_normal(Cc).astype(npdt)
            rm, rv = np.zeros(Cc, npdt), np.ones(Cc, npdt)
            ref = aten.native_batch_norm(torch.from_numpy(x), torch.from_numpy(g), torch.from_numpy(b), torch.from_numpy(rm.copy()), torch.from_numpy(rv.copy()), True, 0.1, 1e-5)
            out3 = (C.c_void_p * 3)(); lib.lamp_native_batch_norm(out3, T(x, sdt), T(g, sdt), T(b, sdt), T(rm, sdt), T(rv, sdt), 1, 0.1, 1e-5)
            for nm, h, r in zip(("y", "mean", "invstd"), out3, ref):
                check(f"bn {nm} {x.shape}", back(S.STen(h)), r, tol * 50)
    except Exception as e:
        bad += 1; print("EXCEPTION", kind, repr(e)[:300])
print(f"seed {seed}: {iters} iterations, {bad} problems")
