"""Build-container helper: refresh tests/golden/reference_kats.json["autograd"] with (expected value, source line) of every case of
/root/reference/lamp-core/src/test/scala/lamp/autograd/autograd.test.scala that tests/kats.py implements.  Data only - names, numbers, line
numbers; the operator calls themselves are restated by hand in tests/kats.py.  Not run on the GPU box (the reference is not there)."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/lamp-core/src/test/scala/lamp/autograd/autograd.test.scala"
REL = "lamp-core/src/test/scala/lamp/autograd/autograd.test.scala"
text = open(SRC).read()
lines = text.split("\n")
found = {}
cuda_only = {}
for i, l in enumerate(lines):
    m = re.match(r'\s*testGradientAndValue(ND|CudaOnly)?\("([^"]+)"', l)
    if not m or "def " in l:
        continue
    # the argument list may continue on the following lines: (input, expected[, tolerance])
    blob = " ".join(lines[i:i + 5])
    m2 = re.search(r'\)\(\s*[A-Za-z0-9_.()\[\] ]+?,\s*(-?[0-9.]+(?:[eE]-?[0-9]+)?)d?\s*[,)]', blob)
    if m2 and m.group(1) == "CudaOnly":
        # the fused-attention cases (CUDA only in the reference: q / k / v of shape (1, 8, 1, 8), f32): value, line, finite-difference step
        e = re.search(r'eps\s*=\s*([0-9.eE-]+)', blob)
        cuda_only[m.group(2)] = {"expected": float(m2.group(1)), "source": f"{REL}:{i + 1}", "eps": float(e.group(1)) if e else 1e-6}
    elif m2:
        found[m.group(2)] = (float(m2.group(1)), i + 1)
sys.path.insert(0, ROOT)
path = os.path.join(ROOT, "tests", "golden", "reference_kats.json")
gold = json.load(open(path))
import importlib
os.environ["LAMP_KATS_NO_ASSERT"] = "1"
kats = importlib.import_module("tests.kats")
missing = []
for name in kats.CASES:
    if name in found:
        gold["autograd"][name] = {"expected": found[name][0], "source": f"{REL}:{found[name][1]}"}
    elif name not in gold["autograd"]:
        missing.append(name)
gold["sdpa"] = cuda_only
json.dump(gold, open(path, "w"), indent=1)
print(len(gold["autograd"]), "cases;", "not found in the reference:", missing)
skipped = sorted(set(found) - set(kats.CASES))
print("reference cases not mirrored:", skipped)
