"""Host cold preparation (A6-A10) timings: values only and with a 16-parameter Jacobian, headline and C5 shapes."""
import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from smcpp_amd import _engine as E, synth
E.lib().smcpp_set_num_threads(int(os.environ.get("PREP_THREADS", 12)))
for M, n in ((64, 20), (256, 50)):
    hs = synth.hidden_states(M); a, s = synth.model_pieces()
    keys = np.array([[x, b, nb] for x in (0, 1, 2) for nb in (n,) for b in range(n + 1)][:3 * (n + 1)], dtype=np.int32)
    da = np.eye(len(a))
    for name, f in (("values", lambda: E.host_prep_onepop(n, hs, 0.5, a, s, synth.THETA, synth.RHO, 1.0, keys)),
                    ("nder=16", lambda: E.host_prep_onepop_jac(n, hs, 0.5, a, da, s, synth.THETA, synth.RHO, 1.0, keys))):
        f(); f()
        t = time.perf_counter(); N = 10
        for _ in range(N): f()
        print(f"M={M} n={n} {name}: {(time.perf_counter() - t) / N * 1e3:.3f} ms per call (incl. ctypes marshalling)")
