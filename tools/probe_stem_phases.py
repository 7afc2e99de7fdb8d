"""Phase shares (az_nn_debug bit 4: s_memtime stamps per wavefront) of the stem and of a residual block of the
evaluator: P1 (tile -> normalised image in LDS), MFMA + epilogue, P3 (store) and the barriers between them.

    python tools/probe_stem_phases.py
"""
import os, sys, runpy
sys.argv = ["probe_nn.py", "heads", "0", "1"]
ns = runpy.run_path(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools", "probe_nn.py"))
import numpy as np, ctypes as C
L = ns["L"]; timed = ns["timed"]
for name in ("stem_embed", "conv"):
    L.az_nn_debug(0); t0 = timed(ns[name])
    L.az_nn_debug(16); t1 = timed(ns[name])
    buf = np.zeros(2048 * 8, dtype=np.uint64)
    L.az_nn_conv_profile.argtypes = [C.c_void_p, C.c_int]
    L.az_nn_conv_profile(buf.ctypes.data, buf.size)
    ph = buf.reshape(2048, 8)[:, :6].astype(np.float64)
    tot = ph.sum(1).mean()
    print(name, "%.1f us (%.1f with stamps):" % (t0, t1), " ".join("%s %.1f%%" % (n, 100 * ph[:, k].mean() / tot) for k, n in enumerate(("P1", "bar1", "MFMA+epi", "wait", "bar2", "P3"))))
L.az_nn_debug(0)
