"""Debug aid: errors of the fused attention forward / backward against an f64 numpy reference. usage: attn_check.py B H Sq Sk D causal"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lamp_amd import sten as S
from lamp_amd._capi import lib

B, H, Sq, Sk, D, causal = (int(a) for a in sys.argv[1:7])
rng = np.random.default_rng(0)
bf = lambda a: S.bf16_bits_to_f32(S.f32_to_bf16_bits(a.astype(np.float32))).astype(np.float64)
q, go = bf(rng.standard_normal((B, H, Sq, D))), bf(rng.standard_normal((B, H, Sq, D)))
k, v = bf(rng.standard_normal((B, H, Sk, D))), bf(rng.standard_normal((B, H, Sk, D)))
s = q @ k.transpose(0, 1, 3, 2) / np.sqrt(D)
if causal:
    s = np.where(np.triu(np.ones((Sq, Sk), bool), 1), -np.inf, s)
m = s.max(-1, keepdims=True)
p = np.exp(s - m); l = p.sum(-1, keepdims=True); p /= l
o = p @ v
dv = p.transpose(0, 1, 3, 2) @ go
dp = go @ v.transpose(0, 1, 3, 2)
ds = p * (dp - (dp * p).sum(-1, keepdims=True)) / np.sqrt(D)
dq, dk = ds @ k, ds.transpose(0, 1, 3, 2) @ q
T = lambda a: S.STen.from_numpy(a.astype(np.float32), 0, S.BF16)
tq, tk, tv, tg = T(q), T(k), T(v), T(go)
oo, ll = C.c_void_p(), C.c_void_p()
lib.lamp_scaled_dot_product_attention(C.byref(oo), C.byref(ll), tq, tk, tv, causal, 0.0)
O, L = S.STen(oo), S.STen(ll)
out3 = (C.c_void_p * 3)()
lib.lamp_scaled_dot_product_attention_backward(out3, tg, tq, tk, tv, O, L, causal, 0.0)
err = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
print("out", err(O.to_numpy(), o), "lse", err(L.to_numpy(), (m + np.log(l))[..., 0]))
for name, h, r in zip(("dq", "dk", "dv"), out3, (dq, dk, dv)):
    g = S.STen(h).to_numpy().astype(np.float64)
    e = np.abs(g - r)
    print(name, err(g, r), "nan:", np.isnan(g).sum(), "worst index", np.unravel_index(np.nanargmax(e), e.shape))
