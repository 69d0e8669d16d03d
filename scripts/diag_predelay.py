"""Per-block error after predelay changes (development aid)."""
import sys

import numpy as np

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import oracle  # noqa: E402
from test_gpu_parity import _run_with_predelay_events  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "jack"
period = int(sys.argv[2]) if len(sys.argv) > 2 else 256
nb, n_ref = 120, 8192
pm = period // 256
events = {0: 300, 20 // pm: 1500, 50 // pm: 64, 58 // pm: 0, 90 // pm: 4096}
_, _, _, got, want = _run_with_predelay_events(oracle, n_ref, nb, (5000, 4000), events, mode, period)
d = got - want
for b in range(nb):
    s = slice(b * 256, (b + 1) * 256)
    e = np.sqrt(np.mean(d[:, s] ** 2))
    if e > 1e-6:
        k = np.argmax(np.abs(d[0, s]) > 1e-5)
        print(b, f"{e:.3e}", "first bad sample", k, "sig", f"{np.sqrt(np.mean(want[:, s] ** 2)):.3e}")
