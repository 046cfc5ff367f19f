"""lamp-umap on the MI355X backend: kNN graph, edge weights and the gradient-descent layout.

Mirror of `lamp.umap.Umap` (lamp-umap/src/main/scala/lamp/umap/umap.scala): `umap` (:356-414), `umapCustomKnn` (:415-460),
`edgeWeights` (:50-113), `optimize` (:115-286).  Every stage runs on the GPU through the C ABI:

  kNN indices        lamp_knn_squared_euclidean   (lamp.knn.knnSearch, query rows in minibatches)
  kNN distances      lamp_knn_row_distances       (the JVM loop umap.scala:382-402: exact f64 distances)
  edge weights b     lamp_umap_edge_weights       (JVM loops in the reference)
  layout             lamp_umap_loss_grad_skip_self + AdamW(wd 0, clip 1, beta2 0.95) on f64 locations, `iterations` epochs

Kept from the reference: the optimisation is always f64; negatives are `randint(0, total - 1)` (the last point is never drawn)
and pairs that hit themselves are dropped; the four gathers of `locations` accumulate with weights 1, 2, 4, 8 because of
`IndexSelect`'s backward (`out += out.indexAdd(..)`, ops.scala:186-191).  Not reproducible from the reference: the initial
layout (the JVM's Cmwc5 generator) and the sampled negatives (libtorch's generator) - both come from this library's Philox
generator seeded with `randomSeed`.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import nn
from . import sten as S
from ._capi import lib, f64_array, handle_array


def knn_search(features: S.STen, query: S.STen, k: int, minibatchSize: int = 1000) -> S.STen:
    """lamp.knn.knnSearch with SquaredEuclideanDistance (knn/package.scala:60-121): i64 [q, k] neighbour indices.
    `minibatchSize` is the reference's query batch; the kernel chunks the distance block itself, so batches are merged into
    calls of at most 1048576 queries (same result; every call pays for the data set's norms, its bf16 planes and the sample search of
    kernels/knn_split.hip once)."""
    q = query.shape[0]
    step = max(int(minibatchSize), 1)
    step = max(step, min(q, 1048576) // step * step) if q > step else step
    parts = []
    for lo in range(0, q, step):
        i = C.c_void_p()
        lib.lamp_knn_squared_euclidean(C.byref(i), None, features, query.slice(0, lo, min(lo + step, q)), int(k))
        parts.append(S.STen(i))
    return parts[0] if len(parts) == 1 else S.STen.cat(parts, 0)


def edge_weights(knn_distances: S.STen, knn: S.STen) -> S.STen:
    """Umap.edgeWeights: [m, 3] f64 rows (i, j, b)."""
    o = C.c_void_p()
    lib.lamp_umap_edge_weights(C.byref(o), knn_distances, knn)
    return S.STen(o)


def optimize(edgeWeights: S.STen, total: int, lr: float, iterations: int, minDist: float, negativeSampleSize: int, randomSeed: int,
             balanceAttractionsAndRepulsions: bool, repulsionStrength: float, device: int, numDim: int,
             positiveSamples: Optional[int] = None, log=None, lossHistory: Optional[list] = None) -> Tuple[S.STen, float]:
    """Umap.optimize (umap.scala:115-286).  `lossHistory`: a list that receives the losses of the last 50 iterations (read back after the
    loop: no host synchronisation per iteration, unlike `log`)."""
    lib.lamp_manual_seed(int(randomSeed))
    index1 = edgeWeights.select(1, 0).castToLong()
    index2 = edgeWeights.select(1, 1).castToLong()
    b = edgeWeights.select(1, 2).contiguous()
    locations = S.STen.rand([total, numDim], S.F64, device)
    opt = nn.AdamW_factory(weightDecay=0.0, learningRate=lr, clip=1.0)([locations])
    grad = S.STen.zeros([total, numDim], S.F64, device)
    weights = f64_array([1.0, 2.0, 4.0, 8.0])
    last = 0.0
    kept = []
    for it in range(int(iterations)):
        if positiveSamples is None:
            i1, i2, bb = index1, index2, b
        else:
            pos = S.STen.randint(0, index1.shape[0], [min(int(positiveSamples), index1.shape[0])], S.I64, device)
            i1, i2, bb = index1.indexSelect(0, pos), index2.indexSelect(0, pos), b.indexSelect(0, pos)
        grad.zero_()
        out = C.c_void_p()
        # ii = i1.repeatInterleave(negativeSampleSize), jj = randint(0, total - 1, ...) (umap.scala:211-213), the reference's mask = ii.ne(jj) and its
        # two maskedSelects: all inside the layout kernel - the negatives are drawn there (counter-based, randint's distribution), pairs that hit
        # themselves are skipped and the repulsion is normalised by the number of pairs kept.  Same sums; no 45M-element index tensors
        lib.lamp_umap_loss_grad_sampled(C.byref(out), grad, locations, i1, i2, bb, int(negativeSampleSize), int(total) - 1, float(minDist),
                                        int(bool(balanceAttractionsAndRepulsions)), float(repulsionStrength), weights)
        loss = S.STen(out)
        if lossHistory is not None and it >= int(iterations) - 50:
            kept.append(loss)
        if log is not None or it == int(iterations) - 1:
            last = float(loss.to_numpy().reshape(-1)[0])
            if log is not None:
                log(f"loss in epoch: {(it, last)}")
        opt.step([grad], 1.0)
    if lossHistory is not None:
        lossHistory.extend(float(t.to_numpy().reshape(-1)[0]) for t in kept)
    return locations, last


def optimize_sharded(edgeWeights: S.STen, total: int, lr: float, iterations: int, minDist: float, negativeSampleSize: int, randomSeed: int,
                     balanceAttractionsAndRepulsions: bool, repulsionStrength: float, device: int, numDim: int, comm, world: int, rank: int,
                     log=None) -> Tuple[S.STen, float]:
    """Umap.optimize with the edge list sharded over `world` ranks (lamp-umap itself is single-device): rank r takes
    every world-th edge and draws the negatives of its own edges; the normalisers (sum of b, number of kept negatives) are made
    global with two small all-reduces, the [n, numDim] gradient (16 MB f64 at 1M points) and the loss with one each per iteration;
    every rank applies the same AdamW step, so the layouts stay bit-identical across ranks.  world == 1 reproduces `optimize`."""
    def all_reduce(t):
        if comm is not None:
            lib.lamp_comm_all_reduce(handle_array([t.h]), handle_array([comm.value if isinstance(comm, C.c_void_p) else comm]), 1, 0)

    lib.lamp_manual_seed(int(randomSeed))
    locations = S.STen.rand([total, numDim], S.F64, device)                 # same seed, same initial layout on every rank
    if world > 1:
        lib.lamp_manual_seed(int(randomSeed) + 7919 * (rank + 1))           # but different negatives
    sel = S.STen.from_numpy(np.arange(rank, edgeWeights.shape[0], world, dtype=np.int64), device)
    index1 = edgeWeights.select(1, 0).castToLong().indexSelect(0, sel)
    index2 = edgeWeights.select(1, 1).castToLong().indexSelect(0, sel)
    b = edgeWeights.select(1, 2).contiguous().indexSelect(0, sel)
    bsum = b.sum().view(1)
    all_reduce(bsum)
    opt = nn.AdamW_factory(weightDecay=0.0, learningRate=lr, clip=1.0)([locations])
    grad = S.STen.zeros([total, numDim], S.F64, device)
    weights = f64_array([1.0, 2.0, 4.0, 8.0])
    last = 0.0
    for it in range(int(iterations)):
        # the same draws as optimize's kernel makes (one counter block of the generator per iteration), written out: the count of kept pairs has to
        # be made global between them and the loss
        ii_h, jj_h = C.c_void_p(), C.c_void_p()
        lib.lamp_umap_negatives(C.byref(ii_h), C.byref(jj_h), index1, int(negativeSampleSize), int(total) - 1)
        ii, jj = S.STen(ii_h), S.STen(jj_h)
        kept = C.c_void_p()
        lib.lamp_count_ne(C.byref(kept), ii, jj)
        kept = S.STen(kept)
        all_reduce(kept)
        grad.zero_()
        out = C.c_void_p()
        lib.lamp_umap_loss_grad_sharded(C.byref(out), grad, locations, index1, index2, b, ii, jj, float(minDist), int(bool(balanceAttractionsAndRepulsions)),
                                        float(repulsionStrength), weights, bsum, kept)
        loss = S.STen(out).view(1)
        all_reduce(grad)
        if log is not None or it == int(iterations) - 1:
            all_reduce(loss)
            last = float(loss.to_numpy().reshape(-1)[0])
            if log is not None:
                log(f"loss in epoch: {(it, last)}")
        opt.step([grad], 1.0)
    return locations, last


def umapCustomKnn(knn: S.STen, knnDistances: S.STen, device: int = 0, numDim: int = 2, lr: float = 0.1, iterations: int = 500,
                  minDist: float = 0.0, negativeSampleSize: int = 5, randomSeed: int = 42, balanceAttractionsAndRepulsions: bool = True,
                  repulsionStrength: float = 1.0, positiveSamples: Optional[int] = None, log=None, lossHistory: Optional[list] = None):
    b = edge_weights(knnDistances, knn)
    layout, loss = optimize(b, knn.shape[0], lr, iterations, minDist, negativeSampleSize, randomSeed, balanceAttractionsAndRepulsions,
                            repulsionStrength, device, numDim, positiveSamples, log, lossHistory)
    return layout, b, loss


def umap(data: np.ndarray, device: int = 0, precision: str = "f64", k: int = 10, numDim: int = 2, knnMinibatchSize: int = 1000,
         lr: float = 0.1, iterations: int = 500, minDist: float = 0.0, negativeSampleSize: int = 5, randomSeed: int = 42,
         balanceAttractionsAndRepulsions: bool = True, repulsionStrength: float = 1.0, positiveSamples: Optional[int] = None, log=None,
         lossHistory: Optional[list] = None):
    """Umap.umap: returns (layout [n, numDim] f64, umap graph b [m, 3] f64, final loss).  `precision` is the kNN search
    precision ("f64" = DoublePrecision, "f32" = SinglePrecision); the layout is always optimised in f64."""
    data = np.ascontiguousarray(data, dtype=np.float64)
    X = S.STen.from_numpy(data if precision == "f64" else data.astype(np.float32), device, S.F64 if precision == "f64" else S.F32)
    knn = knn_search(X, X, k, knnMinibatchSize)
    X64 = X if precision == "f64" else S.STen.from_numpy(data, device, S.F64)
    d = C.c_void_p()
    lib.lamp_knn_row_distances(C.byref(d), X64, knn)
    return umapCustomKnn(knn, S.STen(d), device, numDim, lr, iterations, minDist, negativeSampleSize, randomSeed,
                         balanceAttractionsAndRepulsions, repulsionStrength, positiveSamples, log, lossHistory)
