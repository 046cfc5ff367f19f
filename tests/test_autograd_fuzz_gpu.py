"""Random expression graphs through both autograd implementations (the oracle on CPU, the HIP path through the C ABI) in float64:
values and the gradients of every leaf must agree.  The graphs reuse variables (several consumers: the accumulate-into-gradient rule,
ops.scala "out += ..."), broadcast, transpose, reshape, slice and reduce - the plumbing the fixed KATs touch one operator at a time."""
import numpy as np
import pytest

from tests.backends import HipBackend, OracleBackend


def _build(B, rng, leaves_np):
    """the same pseudo-random program on backend B: returns (loss variable, leaf variables)"""
    leaves = [B.param(a) for a in leaves_np]
    pool = list(leaves)                      # all [4, 6] matrices
    for step in range(int(rng.integers(6, 14))):
        k = int(rng.integers(0, 12))
        a = pool[int(rng.integers(0, len(pool)))]
        b = pool[int(rng.integers(0, len(pool)))]
        if k == 0: v = a + b
        elif k == 1: v = a - b
        elif k == 2: v = a * b
        elif k == 3: v = a / (b * b + 1.5)
        elif k == 4: v = (a * 0.3).tanh()
        elif k == 5: v = (a * 0.2).sigmoid() * b
        elif k == 6: v = a.relu() + (b * 0.1).exp()
        elif k == 7: v = a.mm(b.transpose(0, 1)).mm(a) * 0.05                 # [4,6].[6,4].[4,6]
        elif k == 8: v = a + b.sum([0], True)                                   # broadcast of a row
        elif k == 9: v = a * b.mean([1], True)                                  # broadcast of a column
        elif k == 10: v = a.logSoftMax(1) + b
        else: v = a.reshape([6, 4]).transpose(0, 1) * 0.5 + b                   # view + transpose: back to [4,6]
        pool.append(v)
    loss = pool[-1]
    for extra in pool[len(leaves):-1:2]:
        loss = loss + extra * 0.25
    return (loss * loss).sum() * 0.01 + loss.sum(), leaves


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_random_expression_graphs_agree_with_the_oracle(gpu, seed):
    rng0 = np.random.default_rng(1000 + seed)
    leaves_np = [rng0.standard_normal((4, 6)) for _ in range(3)]
    out = []
    for B in (OracleBackend(), HipBackend()):
        L, leaves = _build(B, np.random.default_rng(seed), leaves_np)
        L.backprop()
        out.append((B.scalar(L), [B.grad(v) for v in leaves]))
    (lo, go), (lh, gh) = out
    assert abs(lo - lh) <= 1e-10 * max(1.0, abs(lo)), (lo, lh)
    for a, b in zip(go, gh):
        assert np.allclose(a, b, rtol=1e-9, atol=1e-10 * max(1.0, np.abs(a).max())), np.abs(a - b).max()
