"""Build container only (needs /root/reference): the first 1000 records of the reference's MNIST test resource
(lamp-core/src/test/resources/mnist_test.csv.gz - label + 784 pixels per line, header first) as a compact fixture,
tests/golden/mnist_first1000.npz.  The reference's UMAP test (lamp-umap/src/test/scala/lamp/umap/umap.test.scala:57-79) reads the
first 1000 lines of /mnist_train.csv.gz, a resource that is not in the tree; the test-set file has the same format and
distribution and is the one data file of that kind the reference holds."""
import gzip, os, sys
import numpy as np
src = "/root/reference/lamp-core/src/test/resources/mnist_test.csv.gz"
rows = []
with gzip.open(src, "rt") as f:
    header = f.readline()
    for line in f:
        rows.append([int(v) for v in line.strip().split(",")])
        if len(rows) == 1000:
            break
a = np.array(rows, dtype=np.int64)
assert a.shape == (1000, 785) and a[:, 1:].max() <= 255 and a[:, 1:].min() >= 0
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "mnist_first1000.npz")
np.savez_compressed(out, labels=a[:, 0].astype(np.uint8), pixels=a[:, 1:].astype(np.uint8))
print(out, os.path.getsize(out), "bytes; label histogram", np.bincount(a[:, 0]))
