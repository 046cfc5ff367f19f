"""lamp.knn on the MI355X backend (lamp-knn/src/main/scala/lamp/knn/package.scala).

  SquaredEuclideanDistance / JaccardDistance   :12-44    distance functions (markers here: the search kernels have them built in)
  knn, knnMinibatched                          :46-80    index search on device tensors
  knnSearch                                    :98-121   host matrix in, host int matrix out
  regression / classification                  :82-96    host-side post-processing (saddle on the JVM, numpy here)
  knnClassification / knnRegression            :122-160

The search never materialises a query x data distance block: for f32 / f64 data of up to 128 features and k <= 16 the top-k is fused
into the distance GEMM (knn_fused.hip); `minibatchSize` therefore only bounds the rows per launch and does not change the result.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np

from . import sten as S
from ._capi import lib


class _Distance:
    def __init__(self, name, fn):
        self.name, self._fn = name, fn

    def __repr__(self):
        return self.name


SquaredEuclideanDistance = _Distance("SquaredEuclideanDistance", "lamp_knn_squared_euclidean")
JaccardDistance = _Distance("JaccardDistance", "lamp_knn_jaccard")


def knn(d: S.STen, query: S.STen, k: int, distanceMatrix: _Distance = SquaredEuclideanDistance) -> S.STen:
    """indices [q, k] (i64) of the k smallest distances per query row"""
    i = C.c_void_p()
    getattr(lib, distanceMatrix._fn)(C.byref(i), None, d, query, int(k))
    return S.STen(i)


def knnMinibatched(d: S.STen, query: S.STen, k: int, distanceMatrix: _Distance = SquaredEuclideanDistance, minibatchSize: int = 1 << 30) -> S.STen:
    rows = query.shape[0]
    step = max(1, min(int(minibatchSize), rows))
    # the kernel streams the data set once per launch (and the split-f16 path prepares the data set once per call): merge the
    # reference's small minibatches into launches of up to 1048576 rows
    step = max(step, min(rows, 1048576) // step * step)
    parts = [knn(d, query.slice(0, lo, min(lo + step, rows)), k, distanceMatrix) for lo in range(0, rows, step)]
    return parts[0] if len(parts) == 1 else S.STen.cat(parts, 0)


def knnSearch(features: np.ndarray, query: np.ndarray, k: int, distance: _Distance = SquaredEuclideanDistance, device: int = 0,
              precision: str = "f64", minibatchSize: int = 1 << 30) -> np.ndarray:
    dt, npdt = (S.F64, np.float64) if precision == "f64" else (S.F32, np.float32)
    f = S.STen.from_numpy(np.ascontiguousarray(features, dtype=npdt), device, dt)
    q = S.STen.from_numpy(np.ascontiguousarray(query, dtype=npdt), device, dt)
    return knnMinibatched(f, q, k, distance, minibatchSize).to_numpy().astype(np.int32)


def regression(values: np.ndarray, indices: np.ndarray) -> np.ndarray:
    """mean of the neighbours' values per query row (package.scala:82-83)"""
    return np.asarray(values, dtype=np.float64)[indices].mean(1)


def classification(values: Sequence[int], indices: np.ndarray, numClasses: int, log: bool) -> np.ndarray:
    """[queries, numClasses] class frequencies among the neighbours, log(v + 1e-6) if `log` (package.scala:85-96)"""
    sel = np.asarray(values, dtype=np.int64)[indices]
    freq = np.stack([(sel == c).sum(1) / sel.shape[1] for c in range(numClasses)], 1).astype(np.float64)
    return np.log(freq + 1e-6) if log else freq


def knnClassification(features, values, query, k, distance=SquaredEuclideanDistance, device=0, precision="f64", minibatchSize=1 << 30, log=False):
    idx = knnSearch(features, query, k, distance, device, precision, minibatchSize)
    return classification(values, idx, len(set(np.asarray(values).tolist())), log)


def knnRegression(features, values, query, k, distance=SquaredEuclideanDistance, device=0, precision="f64", minibatchSize=1 << 30):
    return regression(values, knnSearch(features, query, k, distance, device, precision, minibatchSize))
