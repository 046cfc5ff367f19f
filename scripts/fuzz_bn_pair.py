"""Randomised sweep of relu(bn(x) + bn2(x2)) as one op (lamp_native_batch_norm2_add_relu + _backward) against the chain it replaces
(native_batch_norm(x2) -> native_batch_norm_add_relu(x, .) and their two backward calls): forward bitwise; backward bitwise where the entry
point runs the chain, within bf16 rounding of the channel sums where the one-pass kernel serves it.  usage: fuzz_bn_pair.py [seed] [iterations]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lamp_amd._capi import lib; lib.load()
from lamp_amd import sten as S
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.default_rng(seed)
bad = 0


def T(a, dt): return S.STen.from_numpy(a.astype(np.float32), 0, dt)


for it in range(iters):
    dt = [S.BF16, S.BF16, S.F32][it % 3]
    N = int(rng.choice([1, 3, 8, 37, 256, 1000, 2048, 2049, 4100])); Cc = int(rng.choice([1, 5, 6, 16, 100, 128, 130]))
    H = int(rng.choice([8, 9, 12, 16, 20])); W_ = int(rng.choice([8, 16, H]))
    if N * Cc * H * W_ > 40_000_000: N = max(1, 40_000_000 // (Cc * H * W_))
    shape = (N, Cc, H, W_)
    x, x2, gy = (rng.standard_normal(shape).astype(np.float32) * s + o for s, o in ((2.0, 0.3), (1.5, -0.2), (1.0, 0.0)))
    w, b, w2, b2 = (rng.standard_normal(Cc).astype(np.float32) * 0.5 + o for o in (1.0, 0.0, 0.8, 0.1))
    X, X2, GY, Wt, Bt, W2, B2 = (T(a, dt) for a in (x, x2, gy, w, b, w2, b2))
    z, o = np.zeros(Cc, np.float32), np.ones(Cc, np.float32)
    RMc, RVc, RM2c, RV2c, RMf, RVf, RM2f, RV2f = (T(a, dt) for a in (z, o, z, o, z, o, z, o))
    ol = (C.c_void_p * 3)(); lib.lamp_native_batch_norm(ol, X2, W2, B2, RM2c, RV2c, 1, 0.1, 1e-5)
    l, sm2c, si2c = (S.STen(h) for h in ol)
    oc = (C.c_void_p * 3)(); lib.lamp_native_batch_norm_add_relu(oc, X, l, Wt, Bt, RMc, RVc, 1, 0.1, 1e-5)
    yc, smc, sic = (S.STen(h) for h in oc)
    o5 = (C.c_void_p * 5)(); lib.lamp_native_batch_norm2_add_relu(o5, X, Wt, Bt, RMf, RVf, X2, W2, B2, RM2f, RV2f, 0.1, 0.1, 1e-5, 1e-5)
    yf, smf, sif, sm2f, si2f = (S.STen(h) for h in o5)
    for a_, c_, what in ((yf, yc, "y"), (smf, smc, "mean"), (sif, sic, "invstd"), (sm2f, sm2c, "mean2"), (si2f, si2c, "invstd2"), (RMf, RMc, "rm"), (RVf, RVc, "rv"),
                         (RM2f, RM2c, "rm2"), (RV2f, RV2c, "rv2")):
        if not np.array_equal(a_.to_numpy(), c_.to_numpy(), equal_nan=True):
            bad += 1; print("FORWARD MISMATCH", what, shape, dt)
    o4 = (C.c_void_p * 4)(); lib.lamp_native_batch_norm_add_relu_backward(o4, GY, X, l, Wt, Bt, RMc, RVc, smc, sic, 1, 1e-5, (C.c_uint8 * 4)(1, 1, 1, 1))
    dxc, dwc, dbc, dlc = (S.STen(h) for h in o4)
    o3 = (C.c_void_p * 3)(); lib.lamp_native_batch_norm_backward(o3, dlc, X2, W2, RM2c, RV2c, sm2c, si2c, 1, 1e-5, (C.c_uint8 * 3)(1, 1, 1))
    chain = [dxc, dwc, dbc] + [S.STen(h) for h in o3]
    lib.lamp_kernel_timer_enable(1)
    o6 = (C.c_void_p * 6)(); lib.lamp_native_batch_norm2_add_relu_backward(o6, GY, X, Wt, Bt, smf, sif, X2, W2, B2, sm2f, si2f, 1e-5, 1e-5, (C.c_uint8 * 6)(1, 1, 1, 1, 1, 1))
    buf = C.create_string_buffer(1 << 16); lib.lamp_kernel_timer_report(buf, len(buf)); lib.lamp_kernel_timer_enable(0)
    one_pass = b"bn_bwd_fused" in buf.value
    for h, c_, what in zip(o6, chain, ("dx", "dw", "db", "dx2", "dw2", "db2")):
        f = S.STen(h).to_numpy().astype(np.float64); r = c_.to_numpy().astype(np.float64)
        if one_pass:
            lim = 4e-2 * (np.abs(r) + max(np.abs(r).max(), 1e-30))
            ok = bool((np.abs(f - r) <= lim).all())
        else:
            ok = np.array_equal(f, r, equal_nan=True)
        if not ok:
            bad += 1; print("BACKWARD MISMATCH", what, shape, dt, "one-pass" if one_pass else "chain", float(np.abs(f - r).max()))
print(f"seed {seed}: {iters} cases, {bad} problems")
