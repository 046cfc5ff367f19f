"""helpers shared by the GPU parity tests: torch (oracle side) <-> STen (HIP side)."""
import numpy as np
import torch

from lamp_amd import sten as S

TORCH2LAMP = {torch.float32: S.F32, torch.float64: S.F64, torch.bfloat16: S.BF16, torch.int64: S.I64, torch.bool: S.BOOL,
              torch.int32: S.I32, torch.uint8: S.U8}
DTYPES = [torch.float64, torch.float32, torch.bfloat16]
# forward tolerance per dtype (relative to max |ref|): f32 <= 1e-5 is BASELINE.json's bar; bf16 has 8 bits of mantissa
FWD_TOL = {torch.float64: 1e-12, torch.float32: 1e-5, torch.bfloat16: 1.6e-2}
BWD_TOL = {torch.float64: 1e-10, torch.float32: 1e-3, torch.bfloat16: 3e-2}


def to_sten(t: torch.Tensor, device=0) -> S.STen:
    if t.dtype == torch.bfloat16:
        return S.STen.from_numpy(t.float().numpy(), device=device, dtype=S.BF16)
    return S.STen.from_numpy(t.contiguous().numpy(), device=device, dtype=TORCH2LAMP[t.dtype])


def to_torch(s: S.STen) -> torch.Tensor:
    shape = s.shape
    a = s.to_numpy()
    return torch.from_numpy(np.ascontiguousarray(a)).reshape(shape)


def closed_form(shape, salt=0, scale=1.0, dtype=torch.float32):
    n = int(np.prod(shape)) if len(shape) else 1
    i = torch.arange(n, dtype=torch.int64) + salt
    v = (((i * 7919) % 1009).to(torch.float64) / 1009.0 - 0.5) * scale
    return v.reshape(shape).to(dtype)


def rel_err(got, ref):
    got = got.double() if isinstance(got, torch.Tensor) else torch.as_tensor(got).double()
    ref = ref.double()
    assert list(got.shape) == list(ref.shape), f"shape {list(got.shape)} vs {list(ref.shape)}"
    if ref.numel() == 0:
        return 0.0
    den = ref.abs().max().item()
    return (got - ref).abs().max().item() / (den if den > 0 else 1.0)


def assert_close(got, ref, tol, what=""):
    e = rel_err(got, ref)
    assert e <= tol, f"{what}: relative error {e:.3e} > {tol:.1e}"
