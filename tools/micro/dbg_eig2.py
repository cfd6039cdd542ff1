import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from evcont_amd import ops
from test_gpu_eigensolvers import spectrum, with_spectrum
np.set_printoptions(linewidth=200, precision=2)
kind, n = "graded", 13
rng = np.random.default_rng(1000 + n)
S = with_spectrum(spectrum(kind, n, rng), rng)
X, U, s = [t.cpu().numpy() for t in ops.loewdin(torch.from_numpy(S).to("cuda:0"))]
print("s", s)
E = U.T @ U - np.eye(n)
print("UtU-I\n", E)
print("U U^T - I max", np.abs(U @ U.T - np.eye(n)).max())
print("residual cols", np.abs(S @ U - U * s).max(axis=0))
