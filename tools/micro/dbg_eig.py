import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from evcont_amd import ops
from test_gpu_eigensolvers import spectrum, with_spectrum
for kind, n in (("triple", 30), ("graded", 32), ("graded", 17), ("graded", 13), ("graded", 8), ("random", 30)):
    rng = np.random.default_rng(1000 + n)
    vals = spectrum(kind, n, rng)
    S = with_spectrum(vals, rng)
    X, U, s = ops.loewdin(torch.from_numpy(S).to("cuda:0"))
    X, U, s = X.cpu().numpy(), U.cpu().numpy(), s.cpu().numpy()
    w, v = np.linalg.eigh(S)
    Xref = (v / np.sqrt(w)) @ v.T
    # high-precision reference through the defining equation: residuals
    r_ours = np.abs(X @ S @ X - np.eye(n)).max(); r_np = np.abs(Xref @ S @ Xref - np.eye(n)).max()
    print(os.environ.get("EVC_EIGH_F32"), kind, n, "X-Xnp rel %.1e" % (np.abs(X - Xref).max() / np.abs(Xref).max()),
          "XSX-I ours %.1e numpy %.1e" % (r_ours, r_np), "U orth %.1e" % np.abs(U.T @ U - np.eye(n)).max(),
          "eig err %.1e" % np.abs(np.sort(s) - w).max())
