"""Randomised parity sweep of the row-wise / pooling / kNN kernels against ATen-CPU (f32 and bf16): log_softmax (+backward) over random
dims and odd widths, layer norm (+backward) with / without affine, nll_loss with ignore_index and weights, max / avg pooling with
indices, kNN index sets on random integer-valued points (exact distances)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lamp_amd._capi import lib, i64_array; lib.load()
from lamp_amd import sten as S
import torch
aten = torch.ops.aten
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(seed)
bad = 0
def back(t): return torch.from_numpy(t.castToDouble().to_numpy() if t.dtype not in (S.I64,) else t.to_numpy())
def chk(name, got, ref, tol):
    global bad
    g, r = got.double(), ref.double()
    if g.shape != r.shape: bad += 1; print("SHAPE", name, tuple(g.shape), tuple(r.shape)); return
    if r.numel() == 0: return
    err = (g - r).abs(); lim = tol * (r.abs() + r.abs().mean() + 1e-30)
    if not bool(((err <= lim) | (g == r)).all()):
        bad += 1; print("MISMATCH", name, float(err.max()))
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 200):
    bf = rng.random() < 0.4
    tdt, sdt, tol = (torch.bfloat16, S.BF16, 2.0 ** -6) if bf else (torch.float32, S.F32, 2e-5)
    def mk(shape, scale=1.0):
        t = (torch.from_numpy(rng.standard_normal(shape).astype(np.float32)) * scale).to(tdt)
        return t, S.STen.from_numpy(t.float().numpy(), 0, sdt)
    kind = it % 5
    try:
        if kind == 0:
            shape = [int(rng.integers(1, 40)), int(rng.choice([1, 3, 10, 100, 257, 1000, 4097]))]
            if rng.random() < 0.3: shape = [int(rng.integers(1, 5))] + shape
            dim = int(rng.integers(0, len(shape)))
            x, X = mk(shape, 3.0)
            ref = torch.log_softmax(x.float(), dim).to(tdt)
            o = C.c_void_p(); lib.lamp_log_softmax(C.byref(o), X, dim); Y = S.STen(o)
            chk(f"log_softmax {shape} d{dim} {tdt}", back(Y), ref, tol)
            g, G = mk(shape)
            rb = aten._log_softmax_backward_data(g.float(), ref.float(), dim, torch.float32)
            o2 = C.c_void_p(); lib.lamp_log_softmax_backward_data(C.byref(o2), G, Y, dim)
            chk(f"log_softmax bwd {shape} d{dim} {tdt}", back(S.STen(o2)), rb, tol * 8)
        elif kind == 1:
            rows, width = int(rng.integers(1, 70)), int(rng.choice([5, 33, 64, 96, 768, 1001]))   # widths 1 and 2: x^ is 0 or +-1, every gradient is rounding noise times rstd (up to 316) around 0
            x, X = mk([rows, width], 2.0); affine = rng.random() < 0.6
            w, W = mk([width]); b, Bb = mk([width])
            ref = aten.native_layer_norm(x.float(), [width], w.float() if affine else None, b.float() if affine else None, 1e-5)
            out3 = (C.c_void_p * 3)(); lib.lamp_native_layer_norm(out3, X, i64_array([width]), 1, W if affine else None, Bb if affine else None, 1e-5)
            Y, Mn, Rs = (S.STen(h) for h in out3)
            chk(f"layer_norm {rows}x{width} aff{int(affine)} {tdt}", back(Y), ref[0], tol * 4)
            g, G = mk([rows, width])
            rb = aten.native_layer_norm_backward(g.float(), x.float(), [width], ref[1], ref[2], w.float() if affine else None, b.float() if affine else None, [True, affine, affine])
            o3 = (C.c_void_p * 3)(); lib.lamp_native_layer_norm_backward(o3, G, X, i64_array([width]), 1, Mn, Rs, W if affine else None, Bb if affine else None, (C.c_uint8 * 3)(1, int(affine), int(affine)))
            chk(f"layer_norm dx {rows}x{width} {tdt}", back(S.STen(o3[0])), rb[0], tol * 16)
            if affine:
                chk(f"layer_norm dw {rows}x{width} {tdt}", back(S.STen(o3[1])), rb[1], tol * 16)
                chk(f"layer_norm db {rows}x{width} {tdt}", back(S.STen(o3[2])), rb[2], tol * 16)
        elif kind == 2:
            n, c = int(rng.integers(1, 300)), int(rng.choice([2, 10, 100, 257]))
            x, X = mk([n, c]); lp = torch.log_softmax(x.float(), 1).to(tdt); LP = S.STen.from_numpy(lp.float().numpy(), 0, sdt)
            tg = rng.integers(0, c, n).astype(np.int64); ign = int(rng.integers(0, c)) if rng.random() < 0.5 else -100
            wts, Wt = mk([c]); wts = wts.abs() + 0.1; Wt = S.STen.from_numpy(wts.float().numpy(), 0, sdt)
            red = int(rng.integers(0, 3))
            ref = aten.nll_loss_forward(lp.float(), torch.from_numpy(tg), wts.float(), red, ign)
            o, tw = C.c_void_p(), C.c_void_p(); lib.lamp_nll_loss_forward(C.byref(o), C.byref(tw), LP, S.STen.from_numpy(tg, 0), Wt, red, ign)
            chk(f"nll {n}x{c} red{red} ign{ign} {tdt}", back(S.STen(o)), ref[0], tol * 8)
        elif kind == 3:
            N, Cc, H, W = int(rng.integers(1, 4)), int(rng.integers(1, 5)), int(rng.integers(2, 13)), int(rng.integers(2, 13))
            k = int(rng.integers(1, 4)); st = int(rng.integers(1, 3)); p = int(rng.integers(0, k // 2 + 1))
            if (H + 2 * p - k) // st + 1 < 1 or (W + 2 * p - k) // st + 1 < 1: continue
            x, X = mk([N, Cc, H, W])
            ref = aten.max_pool2d_with_indices(x.float(), [k, k], [st, st], [p, p], [1, 1], False)
            o, ix = C.c_void_p(), C.c_void_p(); lib.lamp_max_pool2d_with_indices(C.byref(o), C.byref(ix), X, k, st, p, 1, 0)
            chk(f"maxpool {x.shape} k{k} s{st} p{p} {tdt}", back(S.STen(o)), ref[0], 0.0)
            if not bool((torch.from_numpy(S.STen(ix).to_numpy()) == ref[1]).all()):
                # ties may pick another position of the same value: compare the values at the indices
                gi = torch.from_numpy(S.STen(ix).to_numpy()); flat = x.float().reshape(N, Cc, -1)
                if not bool((flat.gather(2, gi.reshape(N, Cc, -1)) == flat.gather(2, ref[1].reshape(N, Cc, -1))).all()):
                    bad += 1; print("MISMATCH maxpool indices", x.shape, k, st, p)
            ra = aten.avg_pool2d(x.float(), [k, k], [st, st], [p, p], False, True)
            o2 = C.c_void_p(); lib.lamp_avg_pool2d(C.byref(o2), X, k, st, p, 0, 1)
            chk(f"avgpool {x.shape} k{k} s{st} p{p} {tdt}", back(S.STen(o2)), ra.to(tdt), tol)
        else:
            n, q, d, k = int(rng.integers(20, 400)), int(rng.integers(1, 60)), int(rng.choice([3, 16, 64, 128, 200])), int(rng.integers(1, 11))
            pts = rng.integers(-8, 9, (n, d)).astype(np.float32); qs = rng.integers(-8, 9, (q, d)).astype(np.float32)
            P, Q = S.STen.from_numpy(pts, 0, S.F32), S.STen.from_numpy(qs, 0, S.F32)
            oi, od = C.c_void_p(), C.c_void_p(); lib.lamp_knn_squared_euclidean(C.byref(oi), C.byref(od), P, Q, k)
            got_d = S.STen(od).to_numpy().astype(np.float64)
            dist = ((qs[:, None, :].astype(np.float64) - pts[None, :, :]) ** 2).sum(-1)
            ref_d = np.sort(dist, 1)[:, :k]
            if not np.array_equal(got_d, ref_d):
                bad += 1; print("MISMATCH knn distances", n, q, d, k, np.abs(got_d - ref_d).max())
            gi = S.STen(oi).to_numpy()
            if not np.array_equal(np.take_along_axis(dist, gi, 1), ref_d):
                bad += 1; print("MISMATCH knn indices", n, q, d, k)
    except Exception as e:
        bad += 1; print("EXCEPTION", kind, repr(e)[:300])
print(f"seed {seed}: {bad} problems")
