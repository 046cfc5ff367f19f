"""op-by-op UMAP loss on the HIP autograd (umap.scala:132-176), mirroring oracle.umap_loss."""
from lamp_amd import autograd as A


def hip_umap_loss(locations, i1, i2, i3, i4, b, min_dist=0.0, repulsion_strength=1.0):
    l1, l2 = locations.indexSelect(0, A.const(i1)), locations.indexSelect(0, A.const(i2))
    l3, l4 = locations.indexSelect(0, A.const(i3)), locations.indexSelect(0, A.const(i4))
    bv = A.const(b)
    n1 = l1.euclideanDistance(l2, 1).view([-1])
    if min_dist == 0.0:
        attractions = (n1 * bv).sum() * (-1.0)
    else:
        attractions = (A.CappedShiftedNegativeExponential(n1, min_dist).log() * bv).sum()
    n2 = l3.euclideanDistance(l4, 1).view([-1])
    if min_dist == 0.0:
        repulsions = ((n2 * (-1.0)).exp() * (-1.0)).log1p().sum()
    else:
        repulsions = (A.CappedShiftedNegativeExponential(n2, min_dist) * (-1.0) + 1e-6).log1p().sum()
    return (attractions / bv.sum() + repulsions * (repulsion_strength / l3.shape[0])) * (-1.0)
