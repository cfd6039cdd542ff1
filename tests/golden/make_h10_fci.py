"""Generates tests/golden/h10_fci_t5.npz: BASELINE configs[1] with PHYSICAL inputs — H10 chain,
STO-3G, 5 FCI training states.

Unlike the case_*.npz fixtures (produced by running the reference's own functions) this one is
produced by this repository's host-side generators, evcont_amd/hchain.py (closed-form s-Gaussian
integrals) and evcont_amd/fci_small.py (determinant FCI, PySCF's t-RDM conventions); what it pins is
physics: the continuation must reproduce the independently computed FCI energies at the training
geometries and stay above the FCI energy elsewhere.  ~2 minutes of CPU.

    python tests/golden/make_h10_fci.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from evcont_amd.containers import grow_trdms          # noqa: E402
from evcont_amd.fci_small import SmallFCI             # noqa: E402
from evcont_amd.hchain import s_gaussian_mol          # noqa: E402
from evcont_amd.synthetic import pack_rows            # noqa: E402
from oracle import evcont_oracle as orc               # noqa: E402

N = 10
SPACINGS = (1.4, 1.8, 2.2, 2.8, 3.6)


def chain(d):
    R = np.zeros((N, 3))
    R[:, 0] = d * np.arange(N)
    return R


def oao_integrals(m):
    b = orc.AOBundle(m.S, m.hcore, m.eri, m.ipovlp, m.dhcore, m.eri_ip1, m.aoslices, m.enuc, m.gnuc)
    return orc.integrals_oao(b, orc.loewdin_trafo(m.S))


def main():
    f = SmallFCI()
    vecs, ens, S, one, two = [], [], None, None, None
    for d in SPACINGS:
        m = s_gaussian_mol(chain(d), need_grad=False)
        h1, h2 = oao_integrals(m)
        e, c = f.kernel(h1, h2, N, m.nelec)
        vecs.append(c)
        ens.append(e + m.enuc)
        ov = np.array([np.vdot(vecs[-1], v) for v in vecs])
        r1 = np.empty((len(vecs), N, N))
        r2 = np.empty((len(vecs), N, N, N, N))
        for i, v in enumerate(vecs):
            r1[i], r2[i] = f.trans_rdm12(vecs[-1], v, N, m.nelec)
        S, one, two = grow_trdms(S, one, two, ov, r1, r2)
        print(f"d = {d}: E_FCI = {ens[-1]:.10f}", flush=True)
    # one off-training geometry (bent chain) with its FCI energy, for the variational check
    rng = np.random.default_rng(2024)
    R_test = chain(2.0) + 0.2 * rng.standard_normal((N, 3))
    mt = s_gaussian_mol(R_test, need_grad=False)
    h1, h2 = oao_integrals(mt)
    e_test = f.kernel(h1, h2, N, mt.nelec)[0] + mt.enuc
    print(f"test geometry: E_FCI = {e_test:.10f}")
    assert np.allclose(two, two.transpose(0, 1, 4, 5, 2, 3), atol=1e-12)     # electron-pair exchange
    assert np.allclose(two, two.transpose(1, 0, 2, 3, 4, 5), atol=0)         # container stores both blocks equal
    np.savez_compressed(os.path.join(HERE, "h10_fci_t5.npz"), spacings=np.array(SPACINGS), overlap=S, one_rdm=one,
                        two_rdm_pack2=pack_rows(two, True, True), ens=np.array(ens), R_test=R_test,
                        e_fci_test=e_test)


if __name__ == "__main__":
    main()
