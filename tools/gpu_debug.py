"""Diagnostic run for the GPU box: prints per-case deviations from the golden vectors (not a pytest file)."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import load_golden, GOLDEN_CASES, rel_err
from test_gpu_parity import make_im

cases = sys.argv[1:] or GOLDEN_CASES
for name in cases:
    g = load_golden(name)
    try:
        im = make_im(g)
        im.save_gamma = True
        t = time.time(); im.E_step(); dt = time.time() - t
        ll = im.loglik()
        xs = im.xisums[0]
        gs = im.gamma_sums[0]
        gam = im.gammas[0]
        q = np.array(im.Q(separate=True))
        arg = gam.argmax(0)
        mism = np.nonzero(arg != g["gamma_argmax"])[0]
        strong = g["gamma_margin"] > 1e-5
        gserr = max(np.max(np.abs(gs[tuple(int(x) for x in k)] - v)) / max(np.abs(v).max(), 1e-300)
                    for k, v in zip(g["gs_keys"], g["gs_vals"]))
        st = int(g["gamma_stride"])
        print(json.dumps(dict(case=name, secs=round(dt, 4), ll=ll, ll_rel=abs(ll - float(g["loglik"])) / abs(float(g["loglik"])),
                              xisum_rel=rel_err(xs, g["xisum"]), gs_rel=float(gserr),
                              q_rel=float(np.max(np.abs(q - g["q"]) / np.maximum(np.abs(g["q"]), 1e-12))),
                              gamma_sub_abs=float(np.max(np.abs(gam[:, ::st] - g["gamma_sub"]))),
                              argmax_mismatch=int(len(mism)), argmax_strong_mismatch=int(strong[mism].sum()),
                              timing=im.last_timing())), flush=True)
    except Exception as e:  # noqa
        import traceback; traceback.print_exc()
        print(json.dumps(dict(case=name, error=str(e))), flush=True)
