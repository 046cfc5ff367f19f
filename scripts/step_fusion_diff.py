import os, subprocess, sys, numpy as np
sys.path.insert(0, os.getcwd())
from tests.test_resnet_bf16_gpu import _STEP_DIGEST
root = os.getcwd()
got = {}
for flag in ("0", "1"):
    env = dict(os.environ, LAMP_CONV_DGRAD_PAIR=flag, LAMP_NCV_BN_STATS=flag, LAMP_CONV_WGRAD_PAIR=flag, PYTHONPATH=root)
    f = f"/tmp/step{flag}.npz"
    out = subprocess.run([sys.executable, "-c", _STEP_DIGEST, sys.argv[1], f], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    got[flag] = np.load(f)
a, b = got["0"], got["1"]
for k in a.files:
    u, v = a[k].astype(np.float64).ravel(), b[k].astype(np.float64).ravel()
    print(k, a[k].shape, "norm %.3e" % np.linalg.norm(u), "rel %.3e" % (np.linalg.norm(u - v) / max(np.linalg.norm(u), 1e-30)))
