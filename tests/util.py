"""helpers shared by the GPU parity tests: torch (oracle side) <-> STen (HIP side)."""
import numpy as np
import torch

from lamp_amd import sten as S

TORCH2LAMP = {torch.float32: S.F32, torch.float64: S.F64, torch.bfloat16: S.BF16, torch.int64: S.I64, torch.bool: S.BOOL,
              torch.int32: S.I32, torch.uint8: S.U8, torch.float16: S.F16}
DTYPES = [torch.float64, torch.float32, torch.bfloat16]
# forward tolerance per dtype: f32 <= 1e-5 is BASELINE.json's bar; bf16 has 8 bits of mantissa: rtol 2^-7 per element
FWD_TOL = {torch.float64: 1e-12, torch.float32: 1e-5, torch.bfloat16: 2.0 ** -7}
BWD_TOL = {torch.float64: 1e-10, torch.float32: 1e-3, torch.bfloat16: 2.0 ** -6}


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


def assert_close(got, ref, tol, what="", scale="mean"):
    """per element: |got - ref| <= tol * (|ref| + s) with s = mean|ref| (default) or max|ref| (scale="max").  The relative part is the
    stated tolerance; the absolute part covers elements that are small through cancellation - a sum of K products carries the
    rounding of its terms whatever the size of the result.  For ONE rounding of an exactly accumulated sum the mean magnitude is the
    right yardstick (every single-operator test uses it).  scale="max" is for CHAINS of reduced-precision stages (bf16 attention
    gradients, whole networks in bf16): there an element inherits the rounding of intermediates - probabilities, score gradients,
    activations - whose magnitude is unrelated to the element itself and is bounded by the tensor's largest values."""
    got = got.double() if isinstance(got, torch.Tensor) else torch.as_tensor(got).double()
    ref = ref.double()
    assert list(got.shape) == list(ref.shape), f"{what}: shape {list(got.shape)} vs {list(ref.shape)}"
    if ref.numel() == 0:
        return
    finite = torch.isfinite(ref)
    assert torch.equal(torch.isfinite(got), finite), f"{what}: non-finite pattern differs"
    assert torch.equal(got[~finite].nan_to_num(0.0, 1.0, -1.0), ref[~finite].nan_to_num(0.0, 1.0, -1.0)), f"{what}: infinities differ"
    g, r = got[finite], ref[finite]
    if r.numel() == 0:
        return
    scale = (r.abs().max() if scale == "max" else r.abs().mean()).item()
    d = (g - r).abs()
    bound = tol * (r.abs() + (scale if scale > 0 else 1.0))
    bad = d > bound
    if bool(bad.any()):
        k = int(torch.argmax(d - bound))
        raise AssertionError(f"{what}: {int(bad.sum())} of {r.numel()} elements outside |d| <= {tol:.1e} * (|ref| + {scale:.3e}); worst: got "
                             f"{g[k].item():.9g} ref {r[k].item():.9g} (|d| = {d[k].item():.3e}, bound {bound[k].item():.3e}); "
                             f"max-norm relative error {rel_err(got.nan_to_num(0, 0, 0), ref.nan_to_num(0, 0, 0)):.3e}")
