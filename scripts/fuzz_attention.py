"""Randomised parity sweep of ScaledDotProductAttention (flash kernels for bf16 d = 64 / 128 incl. strided (B, S, H, d) -> (B, H, S, d)
views, causal masks, ragged sequence lengths; composed path otherwise) forward and backward against an f32 softmax(QK^T)V reference."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lamp_amd._capi import lib; lib.load()
from lamp_amd import sten as S
import torch
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(seed)
bad = 0
def back(t): return torch.from_numpy(t.castToFloat().to_numpy())
def chk(name, got, ref, tol):
    global bad
    err = (got.double() - ref.double()).abs(); lim = tol * (ref.double().abs() + ref.double().abs().max())
    if not bool((err <= lim).all()):
        bad += 1; print("MISMATCH", name, float(err.max()), float(ref.abs().max()))
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    B, H = int(rng.integers(1, 4)), int(rng.integers(1, 5)); d = int(rng.choice([64, 128, 32])); Sq = int(rng.choice([7, 64, 100, 128, 384, 513]))
    causal = int(rng.random() < 0.5); strided = rng.random() < 0.5
    Sk = Sq if causal else int(rng.choice([Sq, 96, 200]))
    def mk(Sn):
        t = torch.from_numpy(rng.standard_normal((B, H, Sn, d)).astype(np.float32)).bfloat16()
        if strided:   # stored as (B, S, H, d), handed over as the (B, H, S, d) view - what MultiheadAttention passes
            st = S.STen.from_numpy(t.permute(0, 2, 1, 3).contiguous().float().numpy(), 0, S.BF16).transpose(1, 2)
        else:
            st = S.STen.from_numpy(t.float().numpy(), 0, S.BF16)
        return t, st
    (q, Q), (k, K), (v, V) = mk(Sq), mk(Sk), mk(Sk)
    (go, GO) = mk(Sq)
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    sc = qf @ kf.transpose(-1, -2) / np.sqrt(d)
    if causal: sc = sc.masked_fill(torch.ones(Sq, Sk, dtype=torch.bool).triu(1), float("-inf"))
    ref = torch.softmax(sc, -1) @ vf
    ref.backward(go.float())
    o, l = C.c_void_p(), C.c_void_p()
    lib.lamp_scaled_dot_product_attention(C.byref(o), C.byref(l), Q, K, V, causal, 0.0)
    O, L = S.STen(o), S.STen(l)
    tag = f"B{B} H{H} Sq{Sq} Sk{Sk} d{d} causal{causal} strided{int(strided)}"
    chk("out " + tag, back(O), ref.detach(), 2.0 ** -6)
    out3 = (C.c_void_p * 3)()
    lib.lamp_scaled_dot_product_attention_backward(out3, GO, Q, K, V, O, L, causal, 0.0)
    for nm, h, r in zip(("dq", "dk", "dv"), out3, (qf.grad, kf.grad, vf.grad)):
        chk(nm + " " + tag, back(S.STen(h)), r, 2.0 ** -5)
print(f"seed {seed}: {bad} problems")
