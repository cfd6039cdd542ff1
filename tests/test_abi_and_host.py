"""CPU tests: the C-ABI library loads and exports every symbol include/evcont_hip.h declares
(no compute calls: there is no GPU here), argument validation that happens on the host side of
the ABI, and the pure-host logic (layouts, sharding, cache, array-level mol adapter)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from evcont_amd import build, _lib
    build.build()                       # hipcc cross-compiles gfx950 without a GPU
    return _lib.load()


def header_symbols():
    text = open(os.path.join(REPO, "include", "evcont_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(evc_[A-Za-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound(lib):
    from evcont_amd import _lib
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/evcont_hip.h but not exported"
    assert set(syms) == set(_lib.SIGNATURES), (set(syms) ^ set(_lib.SIGNATURES))
    assert lib.evc_abi_version() == _lib.ABI_VERSION


def test_struct_layouts_match_header():
    from evcont_amd._lib import TrdmSet, Geometry, Outputs
    # sizes implied by the C declarations (x86-64): 4 int32 + 6 int64 + 3 ptr ; 2 int32 + double + 8 ptr ; 6 ptr
    assert C.sizeof(TrdmSet) == 16 + 6 * 8 + 3 * 8
    assert C.sizeof(Geometry) == 8 + 8 + 8 * 8
    assert C.sizeof(Outputs) == 6 * 8


def test_argument_validation_without_gpu(lib):
    """Argument errors are detected before anything is enqueued (rc < 0, message set)."""
    from evcont_amd._lib import TrdmSet
    assert lib.evc_gemv_rows(None, 4, 4, 4, None, 1.0, None, None, 0, None) < 0
    assert b"null pointer" in lib.evc_last_error()
    assert lib.evc_gemv_rows(16, 4, 5, 5, 16, 1.0, 16, 16, 1 << 20, None) < 0       # odd ld
    assert b"even" in lib.evc_last_error()
    assert lib.evc_quarter_transform(16, 16, 0, 500, 32, None) < 0
    assert lib.evc_loewdin(16, None, 200, 16, 16, 16, None, None) < 0
    t = TrdmSet(n=30, ntrain=20, layout=4, rows2=1, rows2_total=1, cols2=1, ld2=2, ld1=900)
    assert lib.evc_workspace_bytes(C.byref(t), 30) == 0
    assert b"layout" in lib.evc_last_error()
    t = TrdmSet(n=30, ntrain=20, layout=2, rows2=210, row_offset=0, rows2_total=210, cols2=405450, ld2=405456,
                ld1=900, two_rdm=256, one_rdm=256, s_train=256)
    nbytes = lib.evc_workspace_bytes(C.byref(t), 30)
    assert 4 * 810000 * 8 < nbytes < 200 << 20
    assert lib.evc_gemv_rows_ws_bytes(210, 405450) >= 210 * 8


def test_missing_library_fails_loudly(tmp_path):
    from evcont_amd import _lib
    with pytest.raises(_lib.EvcontHipError):
        _lib.load(str(tmp_path / "nope.so"))


def test_no_device_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from evcont_amd import _lib
    from evcont_amd.evaluator import DeviceTRDMs
    with pytest.raises(_lib.EvcontHipError):
        DeviceTRDMs(np.zeros((2, 2, 3, 3)), np.zeros((2, 2, 3, 3, 3, 3)), np.eye(2))
    import evcont_amd.electron_integral_utils as eiu
    with pytest.raises(_lib.EvcontHipError):
        eiu.get_loewdin_trafo(np.eye(3))


def test_layout_shapes_and_inference():
    from evcont_amd.evaluator import layout_shape, infer_layout
    assert layout_shape(6, 20, 30) == (400, 810000)
    assert layout_shape(5, 20, 30) == (210, 810000)
    assert layout_shape(3, 20, 30) == (400, 405450)
    assert layout_shape(2, 20, 30) == (210, 405450)
    assert infer_layout(np.zeros((3, 3, 2, 2, 2, 2)), 3, 2) == 6
    assert infer_layout(np.zeros((6, 10)), 3, 2) == 2
    with pytest.raises(ValueError):
        infer_layout(np.zeros((6, 11)), 3, 2)
    with pytest.raises(AssertionError):
        infer_layout(np.zeros((6, 2, 2, 2)), 3, 2)


def test_shard_rows_cover_and_balance():
    from evcont_amd.distributed import shard_rows
    for rows, world in [(210, 8), (210, 1), (400, 8), (15, 4), (3, 8), (465, 8)]:
        got = [shard_rows(rows, world, r) for r in range(world)]
        assert got[0][0] == 0 and got[-1][1] == rows
        assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
        sizes = [b - a for a, b in got]
        assert max(sizes) == -(-rows // world)


def test_cache_fingerprint():
    from evcont_amd import cache
    cache.clear()
    a = np.random.default_rng(0).standard_normal((3, 3, 4, 4))
    b = np.random.default_rng(1).standard_normal((3, 3, 4, 4, 4, 4))
    S = np.eye(3)
    k1 = cache.key_of(a, b, S)
    assert cache.key_of(a, b, S) == k1
    assert cache.key_of(a, b[:2, :2], S) != k1        # a view is a different object
    b[0, 0, 0, 0, 0, 0] += 1.0
    assert cache.key_of(a, b, S) != k1                # in-place change seen by the content sample
    cache.put(k1, "x")
    assert cache.get(k1) == "x" and cache.get(("other",)) is None
    # block checksums: an edit the strided sample of the key does not see is caught when the entry is reused
    big = np.random.default_rng(2).standard_normal((4, 4, 6, 6, 6, 6))           # 82944 elements, sample stride 20
    kb = cache.key_of(a, big, S)
    cache.put(kb, "y", (a, big, S))
    assert cache.get(kb, (a, big, S)) == "y"
    big.reshape(-1)[7] += 1.0e-3                                                    # not a sampled element
    assert cache.key_of(a, big, S) == kb
    assert cache.get(kb, (a, big, S)) == "y" or cache.get(kb, (a, big, S)) is None   # a spot check may or may not see it
    cache.put(kb, "z", (a, big, S))
    big.reshape(-1)[7] -= 1.0e-3
    assert cache.get(kb, (a, big, S)) is None                                       # first reuse verifies every block
    for i in range(10):
        cache.put(("k", i), i)
    assert cache.get(k1) is None                      # LRU eviction
    cache.clear()


def test_cache_repeated_gets_on_unchanged_arrays_all_hit():
    """Stored and spot-checked block sums come from one routine: ten reuses of an unchanged large entry all hit
    (with np.sum against np.add.reduceat about 80 % of the blocks differed in the last bits and every third
    get_energy_with_grad call dropped the resident training set)."""
    from evcont_amd import cache
    cache.clear()
    rng = np.random.default_rng(11)
    one = rng.standard_normal((20, 20, 30, 30))
    two = rng.standard_normal((210, 108345))
    S = np.eye(20)
    k = cache.key_of(one, two, S)
    cache.put(k, "dev", (one, two, S))
    assert [cache.get(k, (one, two, S)) for _ in range(10)] == ["dev"] * 10
    view = rng.standard_normal((6, 6, 8, 8, 8, 8))[:5, :5]          # the sliced views callers pass
    kv = cache.key_of(one, view, S)
    cache.put(kv, "view", (one, view, S))
    assert [cache.get(kv, (one, view, S)) for _ in range(10)] == ["view"] * 10
    for a in (one, two, view):
        ref = cache._all_block_sums(a)
        assert all(cache._block_sum(a, b) == ref[b] for b in range(0, len(ref), 37))
    cache.clear()


def test_array_level_mol_adapter():
    from evcont_amd.integrals import ao_arrays, is_array_mol, energy_nuc, grad_nuc, nao_of
    from evcont_amd.synthetic import make_ao_arrays
    ao = make_ao_arrays(5, 2, 3, ao_sizes=(2, 3))
    assert is_array_mol(ao) and ao_arrays(ao) is ao and nao_of(ao) == 5
    assert energy_nuc(ao) == ao.enuc and np.array_equal(grad_nuc(ao), ao.gnuc)

    class FakePyscfMol:      # has .intor -> treated as a PySCF molecule
        def intor(self, *a, **k):
            raise RuntimeError("would call libcint")
        S = hcore = eri = None
    assert not is_array_mol(FakePyscfMol())


def test_synthetic_layouts_consistent():
    """The four layouts built by pack_rows describe the same object (oracle energies agree)."""
    from evcont_amd.synthetic import make_ao_arrays, make_trdms, pack_rows
    from oracle import evcont_oracle as orc
    ao = make_ao_arrays(4, 2, 5)
    S, one, two = make_trdms(4, 3, 6)
    b = orc.AOBundle(ao.S, ao.hcore, ao.eri, ao.ipovlp, ao.dhcore, ao.eri_ip1, ao.aoslices, ao.enuc, ao.gnuc)
    ref = orc.energy_with_grad(b, one, two, S)
    for p, e in [(True, False), (False, True), (True, True)]:
        E, g = orc.energy_with_grad(b, one, pack_rows(two, p, e), S)
        assert abs(E - ref[0]) < 1e-12
        np.testing.assert_allclose(g, ref[1], rtol=0, atol=1e-11)


def test_sym8_layout_shape_and_compression_switch():
    """Host side of the 8-fold compressed layout: shape arithmetic (include/evcont_hip.h EVC_LAYOUT_SYM8) and the
    package-level switch (no device needed)."""
    from evcont_amd import _lib
    from evcont_amd.evaluator import layout_shape
    from evcont_amd import ab_initio_eigenvector_continuation as aec
    assert _lib.LAYOUT_SYM8 == 8
    text = open(os.path.join(REPO, "include", "evcont_hip.h")).read()
    assert re.search(r"#define\s+EVC_LAYOUT_SYM8\s+8\b", text)
    for T, n in ((20, 30), (5, 10), (3, 1), (30, 28)):
        ms = n * (n + 1) // 2
        assert layout_shape(8, T, n) == (T * (T + 1) // 2, ms * (ms + 1) // 2)
    assert layout_shape(8, 20, 30) == (210, 108345)
    assert layout_shape(2, 20, 30)[1] / layout_shape(8, 20, 30)[1] > 3.7
    assert aec.get_trdm_compression() in (None, "sym8", "auto")
    old = aec.get_trdm_compression()
    try:
        aec.set_trdm_compression("sym8")
        assert aec.get_trdm_compression() == "sym8"
        with pytest.raises(ValueError):
            aec.set_trdm_compression("other")
    finally:
        aec.set_trdm_compression(old)


def test_default_compression_is_decided_per_call():
    """The default mode "auto" of the mol-level API (evcont/ab_initio_gradients_loewdin.py:308-379 called with
    whatever the container holds, MD_utils.py:40-57): the compressed copy only where the caller cannot tell the
    difference -- Hermitian, no predicted RDMs, integrals with the symmetries of real ones."""
    from evcont_amd import ab_initio_eigenvector_continuation as aec
    from evcont_amd.synthetic import make_ao_arrays, make_trdms
    if "EVCONT_AMD_COMPRESS" not in os.environ:
        assert aec._mode_from_env() == "auto"
    sym = make_ao_arrays(5, 2, 1, ip1_rs_symmetric=True)
    gen = make_ao_arrays(5, 2, 2)                       # eri_ip1 a general tensor (the golden fixtures' kind)
    assert sym.integral_symmetry is True and gen.integral_symmetry is False
    assert aec.integrals_have_symmetry(sym) and not aec.integrals_have_symmetry(gen)
    for ao in (sym, gen):                               # undeclared: checked numerically
        ao.integral_symmetry = None
    assert aec.integrals_have_symmetry(sym) and not aec.integrals_have_symmetry(gen)
    bad = make_ao_arrays(5, 2, 3, ip1_rs_symmetric=True)
    bad.integral_symmetry = None
    bad.eri[0, 1, 2, 3] += 1.0e-3                       # eri no longer 8-fold symmetric
    assert not aec.integrals_have_symmetry(bad)
    iu, ju = np.tril_indices(5)
    packed = make_ao_arrays(5, 2, 4, ip1_rs_symmetric=True)
    packed.integral_symmetry = None
    packed.eri = packed.eri[iu, ju][:, iu, ju]           # s4
    packed.eri_ip1 = packed.eri_ip1[:, :, :, iu, ju]     # s2kl
    assert aec.integrals_have_symmetry(packed)

    class Mole:                                          # anything with .intor is a PySCF molecule: libcint integrals
        def intor(self, *a, **k):
            raise AssertionError("not called")
    assert aec.integrals_have_symmetry(Mole())
    S1, one1, two1 = make_trdms(5, 2, 5)
    S2, one2, two2 = make_trdms(5, 2, 6)
    rc = aec.resolve_compression
    assert rc("auto", one1, two1, S1, sym) == "sym8"
    assert rc("auto", one1, two1, S1, sym, hermitian=False) is None
    assert rc("auto", one1, two1, S1, sym, want_rdms=True) is None
    assert rc("auto", one2, two2, S2, gen) is None
    assert rc(None, one1, two1, S1, sym) is None
    assert rc("sym8", one1, two1, S1, gen) == "sym8" and rc("sym8", one1, two1, S1, gen, want_rdms=True) == "sym8"
    assert rc("sym8", one1, two1, S1, sym, hermitian=False) is None
    old = aec.get_trdm_compression()
    try:
        aec.set_trdm_compression("auto")
        assert rc("default", one1, two1, S1, sym) == "sym8"
        aec.set_trdm_compression(None)
        assert rc("default", one1, two1, S1, sym) is None
    finally:
        aec.set_trdm_compression(old)


@pytest.mark.parametrize("n", [1, 2, 3, 5])
def test_sym8_column_images_reproduce_the_symmetrised_tensor(n):
    """Host half of DeviceTRDMs.compress_sym8_: the eight gather columns per compressed entry, for the unpacked and
    the electron-pair-packed source layouts, against a brute-force 8-fold symmetrisation."""
    from evcont_amd.evaluator import sym8_column_images, layout_shape
    from evcont_amd.synthetic import make_trdms, pack_rows
    T = 2
    _, _, two = make_trdms(n, T, 7 + n)            # (T,T,n,n,n,n), pair symmetric but not 8-fold
    g = two[1, 0]
    a = g + g.transpose(1, 0, 2, 3)
    a = a + a.transpose(0, 1, 3, 2)
    gs = (a + a.transpose(2, 3, 0, 1)) / 8.0
    iu, ju = np.tril_indices(n)
    U, V = np.tril_indices(len(iu))
    want = gs[iu[U], ju[U], iu[V], ju[V]]
    assert want.shape[0] == layout_shape(8, T, n)[1]
    full_row = g.reshape(-1)
    got6 = sum(full_row[ix] for ix in sym8_column_images(6, n)) / 8.0
    np.testing.assert_allclose(got6, want, rtol=0, atol=1e-15)
    packed_row = pack_rows(two, True, True)[1]      # pairs in np.tril_indices order: (0,0), (1,0), (1,1)
    got2 = sum(packed_row[ix] for ix in sym8_column_images(2, n)) / 8.0
    np.testing.assert_allclose(got2, want, rtol=0, atol=1e-15)


def test_size_limits_of_round3(lib):
    """Training sets up to 512 states and orbital spaces up to 96 are accepted (the reference bounds neither; its
    scripts go to T = 100 and N = 58); the large-T subspace kernel sizes its own workspace."""
    from evcont_amd._lib import TrdmSet

    def ws(n, T):
        P, n2 = T * (T + 1) // 2, n * n
        M = n2 * (n2 + 1) // 2
        t = TrdmSet(n=n, ntrain=T, layout=2, rows2=P, row_offset=0, rows2_total=P, cols2=M, ld2=(M + 15) // 16 * 16,
                    ld1=n2 + (n2 & 1), two_rdm=256, one_rdm=256, s_train=256)
        return lib.evc_workspace_bytes(C.byref(t), 3)

    assert ws(8, 100) > 0 and ws(8, 512) > 0
    assert ws(8, 513) == 0 and b"ntrain" in lib.evc_last_error()
    assert ws(96, 2) > 4 * 96 ** 4 * 8
    assert ws(97, 2) == 0 and b"n=97" in lib.evc_last_error()
    assert lib.evc_subspace_solve_ws_bytes(32, 1) == 0          # register / LDS kernel: no workspace
    tp = lambda T: (T + 15) // 16 * 16
    assert lib.evc_subspace_solve_ws_bytes(100, 1) == 3 * tp(100) ** 2 * 8
    assert lib.evc_subspace_solve_ws_bytes(100, 7) == 7 * 3 * tp(100) ** 2 * 8
    assert lib.evc_subspace_solve_ws_bytes(200, 1) > 3 * tp(200) ** 2 * 8    # + the matrix itself (does not fit LDS)
    assert lib.evc_subspace_solve_ws_bytes(513, 1) == 0
    # T > 32 without a workspace is an argument error, not a launch
    assert lib.evc_subspace_solve(16, 16, 16, 40, 2, 1, 0.0, 16, 16, None, None, None, None, 0, None) < 0
    assert b"workspace" in lib.evc_last_error()


def test_integral_symmetry_check_on_host():
    """evaluator.check_integral_symmetry (the guard of the compressed layout / packed inputs) on numpy arrays."""
    from evcont_amd.evaluator import check_integral_symmetry
    from evcont_amd._lib import EvcontHipError
    from evcont_amd.synthetic import make_ao_arrays
    n = 5
    good = make_ao_arrays(n, 2, 1, ip1_rs_symmetric=True)
    check_integral_symmetry(good.eri, good.eri_ip1, n)
    iu, ju = np.tril_indices(n)
    check_integral_symmetry(good.eri[iu, ju][:, iu, ju], good.eri_ip1[:, :, :, iu, ju], n)     # packed s4 / s2kl
    check_integral_symmetry(np.stack([good.eri, good.eri]), np.stack([good.eri_ip1] * 2), n)  # batch axis
    bad = make_ao_arrays(n, 2, 2)                                                              # general eri_ip1
    with pytest.raises(EvcontHipError, match="eri_ip1"):
        check_integral_symmetry(bad.eri, bad.eri_ip1, n)
    e = good.eri.copy()
    e[0, 1, 2, 3] += 1e-4
    with pytest.raises(EvcontHipError, match="eri"):
        check_integral_symmetry(e, None, n)
    m = good.eri[iu, ju][:, iu, ju].copy()
    m[0, 3] += 1e-4
    with pytest.raises(EvcontHipError, match="packed s4"):
        check_integral_symmetry(m, None, n)


def test_spot_check_of_the_integral_symmetries():
    """The sampled test the host-side packing helpers run on every call after their first complete one
    (evaluator.spot_check_integral_symmetry): silent on symmetric tensors (full or already packed), raises on a tensor
    that lost a symmetry."""
    import numpy as np
    import pytest
    from evcont_amd import _lib
    from evcont_amd.evaluator import spot_check_integral_symmetry
    from evcont_amd.synthetic import make_ao_arrays
    n = 7
    ao = make_ao_arrays(n, 2, 5, ip1_rs_symmetric=True)
    spot_check_integral_symmetry(ao.eri, ao.eri_ip1, n)
    iu, ju = np.tril_indices(n)
    spot_check_integral_symmetry(ao.eri.reshape(n, n, n, n)[iu, ju][:, iu, ju], None, n)      # packed: nothing to sample
    bad = make_ao_arrays(n, 2, 6, ip1_rs_symmetric=False)
    with pytest.raises(_lib.EvcontHipError):
        spot_check_integral_symmetry(bad.eri, bad.eri_ip1, n)
    e = np.array(ao.eri, copy=True).reshape(n, n, n, n)
    e += 1e-3 * np.random.default_rng(1).standard_normal(e.shape)
    with pytest.raises(_lib.EvcontHipError):
        spot_check_integral_symmetry(e, None, n)


def test_packed_input_flags_beyond_32_orbitals():
    """Host-side rule of the 64-wide pipeline (evaluator._ip1_flag): packed s4 / s2kl inputs are accepted up to 64
    orbitals on the compressed layout, and beyond 32 orbitals the two large arrays are packed together or not at all."""
    import types
    import pytest
    from evcont_amd import _lib
    from evcont_amd.evaluator import _ip1_flag
    both = _lib.FLAG_IP1_S2KL | _lib.FLAG_ERI_S4
    t58 = types.SimpleNamespace(layout=_lib.LAYOUT_SYM8, n=58)
    ao = lambda s2kl, s4: types.SimpleNamespace(ip1_s2kl=s2kl, eri_s4=s4, eri_ip1=object())
    assert _ip1_flag(t58, ao(True, True)) == both
    assert _ip1_flag(t58, ao(False, False)) == 0
    with pytest.raises(_lib.EvcontHipError):
        _ip1_flag(t58, ao(True, False))
    with pytest.raises(_lib.EvcontHipError):
        _ip1_flag(types.SimpleNamespace(layout=_lib.LAYOUT_SYM8, n=70), ao(True, True))
    with pytest.raises(_lib.EvcontHipError):
        _ip1_flag(types.SimpleNamespace(layout=2, n=20), ao(True, True))
    assert _ip1_flag(types.SimpleNamespace(layout=_lib.LAYOUT_SYM8, n=20), ao(True, False)) == _lib.FLAG_IP1_S2KL
