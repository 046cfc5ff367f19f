"""Randomised parity sweep of the bf16 GEMM dispatch (256 x 256 / 256 x 128 / 128 x 128 tiles, split-K, tail split, transposed
operands, beta accumulation) against f32 products of the same bf16 operands."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lamp_amd._capi import lib; lib.load()
from lamp_amd import sten as S
import torch
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(seed)
bad = 0
def T(t): return S.STen.from_numpy(t.float().numpy(), 0, S.BF16)
def back(t): return torch.from_numpy(t.castToFloat().to_numpy())
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    M = int(rng.choice([64, 128, 200, 256, 512, 768, 1024, 3072, 4096, 6400])); N = int(rng.choice([64, 100, 128, 256, 768, 1024, 3072]))
    K = int(rng.choice([64, 96, 128, 512, 768, 2048, 3072]))
    a = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).bfloat16(); b = torch.from_numpy(rng.standard_normal((K, N)).astype(np.float32)).bfloat16()
    mode = it % 3
    if mode == 0:
        got = back(T(a).mm(T(b))); ref = a.float() @ b.float(); tag = "mm"
    elif mode == 1:     # dW += x^T p   (A^T . B with beta = 1)
        c0 = torch.from_numpy(rng.standard_normal((K, N)).astype(np.float32)).bfloat16(); p = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).bfloat16()
        out = T(c0); S.STen.addmm_out_transposed1(out, out, T(a), T(p), 1.0, 1.0); got = back(out); ref = c0.float() + a.float().t() @ p.float(); tag = "addmm_t1"
    else:               # dX = p W^T
        p = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).bfloat16(); out = S.STen.zeros([M, K], S.BF16, 0)
        S.STen.addmm_out_transposed2(out, out, T(p), T(b), 0.0, 1.0); got = back(out); ref = p.float() @ b.float().t(); tag = "addmm_t2"
    err = (got.double() - ref.double()).abs(); lim = 2.0 ** -7 * (ref.double().abs() + ref.double().abs().mean())
    if not bool((err <= lim).all()):
        bad += 1; print("MISMATCH", tag, M, N, K, float(err.max()), float(lim.min()))
print(f"seed {seed}: {bad} problems")
