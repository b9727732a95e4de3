"""Host-side logic and the C-ABI surface (CPU only; no compute kernels are launched)."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest
import torch

import problems as P
from conftest import ROOT


def test_library_loads_and_exports_every_declared_symbol():
    from sepfwi import _native
    L = _native.lib()
    hdr = open(os.path.join(ROOT, "include", "sepfwi.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(sepfwi_[a-z_]+)\s*\(", hdr))
    assert declared == set(_native.EXPORTS), declared ^ set(_native.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.sepfwi_version() >= 100
    assert isinstance(L.sepfwi_last_error(), bytes)


def test_shipped_library_knows_only_the_public_options():
    """include/sepfwi.h: the shipped library accepts the six user options and nothing else -- the tuning knobs and, above all, the
    timing-only switches that give WRONG results (pk_nosync, pk_lock) are unknown names there; they exist only in the
    -DSEPFWI_PROBES build of the same sources, which the A/B scripts and the structure tests load explicitly."""
    from sepfwi import _native
    L = _native.lib()
    assert set(_native.PUBLIC_OPTIONS) == {"bwd_fuse", "batch", "img_every", "quiet_skip", "obs_cache_mb", "probe"}
    for name in _native.PUBLIC_OPTIONS:
        v = L.sepfwi_get_option(name.encode())
        assert v >= 0 and L.sepfwi_set_option(name.encode(), v) == 0, name
    hidden = [k for k in P.OPTION_DEFAULTS if k not in _native.PUBLIC_OPTIONS]
    assert "pk_nosync" in hidden and "pk_lock" in hidden and len(hidden) >= 20
    for name in hidden:
        assert L.sepfwi_get_option(name.encode()) == -1, name
        assert L.sepfwi_set_option(name.encode(), P.OPTION_DEFAULTS[name]) == -1, name      # SEPFWI_EINVAL
        assert b"unknown option" in L.sepfwi_last_error()
    assert L.sepfwi_set_option(b"bwd_fuse", 3) == -1 and L.sepfwi_set_option(b"img_every", 0) == -1      # values are checked too
    with _native.use_variant("probes") as LP:
        for name, v in P.OPTION_DEFAULTS.items():
            assert LP.sepfwi_get_option(name.encode()) == v, name
            assert LP.sepfwi_set_option(name.encode(), v) == 0, name


def test_status_queries_without_a_session_are_errors_not_crashes(tmp_path):
    """sepfwi_get_stats / sepfwi_loop_status / sepfwi_debug_field on a parameter file no call has used yet, and with NULL or undersized
    buffers: SEPFWI_EINVAL with a message (the reference has no such queries; its failures are exit(1), Src/utilities.h:28-36)."""
    import ctypes as C
    from sepfwi import _native
    L = _native.lib()
    pb = P.make_problem(str(tmp_path), nSteps=20)
    fn = pb["para_fname"].encode()
    buf = C.create_string_buffer(64)
    assert L.sepfwi_loop_status(fn, 0, buf, 64) == -1 and b"no session" in L.sepfwi_last_error()
    assert L.sepfwi_loop_status(fn, 0, None, 64) == -1 and L.sepfwi_loop_status(fn, 0, buf, 0) == -1 and L.sepfwi_loop_status(None, 0, buf, 64) == -1
    st = _native.Stats()
    assert L.sepfwi_get_stats(fn, 0, C.byref(st)) == -1 and b"no session" in L.sepfwi_last_error()
    out = (C.c_float * 4)()
    assert L.sepfwi_debug_field(fn, 0, 0, 0, out) == -1


def test_library_does_not_link_hipfft():
    """hipFFT is opened with dlopen when the first FFT plan is made (csrc/conditioning.hip FftApi): it is not a DT_NEEDED entry, so
    a ROCm image without it still loads the propagator; the HIP runtime is the only ROCm library the loader must find."""
    import subprocess
    from sepfwi import _native
    dyn = subprocess.run(["readelf", "-d", _native.LIB_PATH], capture_output=True, text=True, check=True, stdin=subprocess.DEVNULL).stdout
    needed = re.findall(r"\(NEEDED\)\s+Shared library: \[([^\]]+)\]", dyn)
    assert any(n.startswith("libamdhip64") for n in needed), needed
    assert not any("fft" in n.lower() or "rocblas" in n.lower() or "torch" in n.lower() for n in needed), needed


def test_no_gpu_is_a_loud_error_not_a_fallback(tmp_path):
    """On a box without a HIP device the operator must raise (never compute on the CPU)."""
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from sepfwi import fwi_ops
    from sepfwi._native import SepFwiError
    pb = P.make_problem(str(tmp_path), nSteps=20)
    lam, mu, den = pb["lame_init"]
    with pytest.raises(SepFwiError) as e:
        fwi_ops.obscalc(lam, mu, den, pb["Stf"], 1, pb["Shot_ids"], pb["para_fname"])
    assert e.value.code == -4   # SEPFWI_EHIP


def test_cpml_profiles_equal_oracle(oracle):
    from sepfwi import _native
    L = _native.lib()
    for (n, npml, dh, f0, dt) in [(96, 10, 10.0, 25.0, 1e-3), (1064, 32, 10.0, 10.0, 1e-3), (265, 32, 20.0, 10.0, 2e-3), (70, 32, 5.0, 40.0, 2.5e-4)]:
        arrs = [np.zeros(n, np.float32) for _ in range(6)]
        _native.check(L.sepfwi_cpml_profiles(*[a.ctypes.data for a in arrs], n, npml, dh, f0, dt))
        ref = oracle.cpml_init(n, npml, dh, f0, dt)
        for k, a in zip(("K", "a", "b", "K_half", "a_half", "b_half"), arrs):
            assert np.array_equal(a, ref[k]), (n, k)
        # structure the kernels rely on: a == 0 outside the strips, K == 1 there, b == 1 there
        inner = slice(npml + 1, n - npml - 1)
        assert np.all(arrs[1][inner] == 0) and np.all(arrs[4][inner] == 0)
        assert np.all(arrs[0][inner] == 1) and np.all(arrs[2][inner] == 1)


def test_stf_taper_equals_oracle(oracle):
    from sepfwi import _native
    L = _native.lib()
    rng = np.random.default_rng(0)
    for nt, dt in [(1501, 2e-3), (4000, 1e-3), (300, 1e-3), (37, 4e-3)]:
        s = rng.standard_normal(nt).astype(np.float32)
        t = s.copy()
        _native.check(L.sepfwi_stf_taper(t.ctypes.data, nt, dt, 0.001))
        assert np.array_equal(t, oracle.window_stf(s, dt))
        assert t[0] == 0.0


def test_shot_split_equals_torch_linspace_truncation():
    from sepfwi import _native
    from sepfwi.ops import split_shots
    L = _native.lib()
    for g in list(range(1, 70)) + [128, 255, 256, 1000]:
        for n in range(1, min(g, 16) + 1):
            st = np.zeros(n + 1, np.int32)
            _native.check(L.sepfwi_shot_split(g, n, st.ctypes.data))
            assert st.tolist() == split_shots(g, n), (g, n)
            assert st[0] == 0 and st[-1] == g and np.all(np.diff(st) >= 0)
    st = np.zeros(4, np.int32)
    assert L.sepfwi_shot_split(2, 3, st.ctypes.data) == -1     # ngpu > nshots (Torch_Fwi.cpp:49-52)
    with pytest.raises(RuntimeError):
        split_shots(2, 3)


def test_json_errors_are_reported_not_fatal(tmp_path):
    from sepfwi import _native
    L = _native.lib()
    z = np.zeros(4, np.float32)
    ids = np.zeros(1, np.int32)
    bad = tmp_path / "bad.json"
    bad.write_text('{"nz": 10, "nx": ')
    rc = L.sepfwi_cufd(None, None, None, None, None, z.ctypes.data, z.ctypes.data, z.ctypes.data, z.ctypes.data, 2, 0, 1,
                       ids.ctypes.data, str(bad).encode())
    assert rc == -5 and b"JSON" in L.sepfwi_last_error()
    rc = L.sepfwi_cufd(None, None, None, None, None, z.ctypes.data, z.ctypes.data, z.ctypes.data, z.ctypes.data, 2, 0, 1,
                       ids.ctypes.data, str(tmp_path / "nope.json").encode())
    assert rc == -2
    rc = L.sepfwi_cufd(None, None, None, None, None, z.ctypes.data, z.ctypes.data, z.ctypes.data, z.ctypes.data, 7, 0, 1,
                       ids.ctypes.data, str(bad).encode())
    assert rc == -1 and b"calc_id" in L.sepfwi_last_error()
    # calc_id 3 (SEPFWI_CALC_OBSERVE_TO_STORE, the one value beyond the reference's 0 / 1 / 2) is accepted: the call gets as far as the file
    rc = L.sepfwi_cufd(None, None, None, None, None, z.ctypes.data, z.ctypes.data, z.ctypes.data, z.ctypes.data, 3, 0, 1,
                       ids.ctypes.data, str(bad).encode())
    assert rc == -5
    hdr = open(os.path.join(ROOT, "include", "sepfwi.h")).read()
    assert "#define SEPFWI_CALC_OBSERVE_TO_STORE 3" in hdr and "#define SEPFWI_CALC_OBSERVE 2" in hdr


def test_para_and_survey_writers_schema(tmp_path):
    pb = P.make_problem(str(tmp_path), nSteps=20, nshots=3)
    para = json.loads(open(pb["para_fname"]).readline())
    assert set(para) == {"nz", "nx", "dz", "dx", "nSteps", "dt", "f0", "nPoints_pml", "nPad", "survey_fname", "data_dir_name"}
    assert isinstance(para["dt"], float)                         # Parameter.cpp:88 asserts IsDouble()
    sv = json.loads(open(pb["survey_fname"]).readline())
    assert sv["nShots"] == 3 and set(sv["shot0"]) == {"z_src", "x_src", "nrec", "z_rec", "x_rec"}
    assert (pb["nz_pad"]) % 32 == 0


def test_source_and_padding_helpers():
    from sepfwi import utils as ft
    s = ft.sourceGene(10.0, 1501, 0.002)
    it = int(round(1.2 / 10.0 / 0.002))
    assert s.dtype == np.float64 and abs(s[it] - 1e7) < 1e-3 and np.argmax(s) == it
    a = np.arange(12, dtype=np.float32).reshape(3, 4)
    p = ft.padding_numpy_array(a, 2, 3)
    assert p.shape == (3 + 4 + 3, 4 + 4)
    assert np.all(p[:2, 2:-2] == a[0]) and np.all(p[-5:, 2:-2] == a[-1]) and np.all(p[:, :2] == p[:, 2:3]) and np.all(p[:, -2:] == p[:, -3:-2])
    t = torch.tensor(a)
    tp, _, _ = ft.padding(t, t, t, 3, 4, 3, 4, 2, 3)
    assert np.array_equal(tp.numpy(), p)
    assert ft.nPad_for(101, 32) == 27 and ft.nPad_for(64, 32) == 32


class _FakeOps:
    """records the operator inputs; returns analytic 'gradients'"""
    def __init__(self):
        self.calls = []

    def backward(self, Lambda, Mu, Den, Stf, ngpu, Shot_ids, para_fname):
        self.calls.append((Lambda.detach().clone(), Mu.detach().clone(), Den.detach().clone(), ngpu, Shot_ids, para_fname))
        return [torch.tensor([3.5]), torch.full_like(Lambda, 2.0), torch.full_like(Mu, -1.0), torch.full_like(Den, 0.5), torch.zeros_like(Stf)]


def test_fwifunction_contract_and_module_chain_rule(monkeypatch, tmp_path):
    import sepfwi.ops as ops
    from sepfwi import modules as M
    fake = _FakeOps()
    monkeypatch.setattr(ops, "fwi_ops", fake)
    pb = P.make_problem(str(tmp_path), nz=12, nx=14, nPml=4, nSteps=10, nshots=1, nPad=3)
    vp = torch.tensor(pb["init"]["vp"], requires_grad=True)
    vs = torch.tensor(pb["init"]["vs"], requires_grad=True)
    rho = torch.tensor(pb["init"]["rho"], requires_grad=True)
    mask = torch.zeros(pb["nz_pad"], pb["nx_pad"]); mask[4:16, 4:18] = 1.0
    fwi = M.FWI(vp, vs, rho, pb["Stf"], pb["opt"], Mask=mask)
    assert [n for n, _ in fwi.named_parameters()] == ["Vp", "Vs", "Den"]
    assert set(dict(fwi.named_buffers())) == {"Vp_ref", "Vs_ref", "Den_ref", "Mask"}   # Mask: a buffer so .to(device) moves it
    loss = fwi(pb["Shot_ids"], ngpu=1)
    assert float(loss) == 3.5
    (10.0 * loss).backward()      # grad_misfit is ignored by the reference's Function (FWI_ops.py:54-63)
    lam, mu, den, ngpu, ids, pf = fake.calls[0]
    assert lam.shape == (pb["nz_pad"], pb["nx_pad"]) and pf == pb["para_fname"] and ngpu == 1
    assert torch.allclose(mu[4:16, 4:18], torch.tensor(pb["init"]["vs"]) ** 2 * torch.tensor(pb["init"]["rho"]) / 1e6)
    # chain rule with gL=2, gM=-1, gD=0.5 on the unmasked physical cells
    vp0, vs0, r0 = [torch.tensor(pb["init"][k]) for k in ("vp", "vs", "rho")]
    assert torch.allclose(fwi.Vp.grad, 2.0 * 2 * vp0 * r0 / 1e6, rtol=1e-5)
    assert torch.allclose(fwi.Vs.grad, (2.0 * (-4 * vs0) + (-1.0) * 2 * vs0) * r0 / 1e6, rtol=1e-5)
    assert torch.allclose(fwi.Den.grad, 2.0 * (vp0 ** 2 - 2 * vs0 ** 2) / 1e6 - 1.0 * vs0 ** 2 / 1e6 + 0.5, rtol=1e-5)


def test_obj_wrapper_roundtrip(monkeypatch, tmp_path):
    import sepfwi.ops as ops
    from sepfwi import modules as M
    from sepfwi.obj_wrapper import PyTorchObjective
    monkeypatch.setattr(ops, "fwi_ops", _FakeOps())
    pb = P.make_problem(str(tmp_path), nz=12, nx=14, nPml=4, nSteps=10, nshots=1, nPad=3)
    T = lambda k: torch.tensor(pb["init"][k], requires_grad=True)
    fwi = M.FWI(T("vp"), T("vs"), T("rho"), pb["Stf"], pb["opt"])
    obj = PyTorchObjective(fwi, lambda: fwi(pb["Shot_ids"], ngpu=1))
    assert obj.x0.dtype == np.float64 and obj.x0.size == 3 * 12 * 14 and obj.bounds is None
    jac = obj.jac
    assert obj.fun(obj.x0) == 3.5
    g = jac(obj.x0)
    assert g.dtype == np.float64 and g.shape == obj.x0.shape and np.all(np.isfinite(g))


def test_das_fiber_key_is_optional_and_validated(tmp_path):
    """The fibre-direction extension does not change default files (byte-identical to the reference's paraGen output)
    and rejects anything but the two directions, on the Python and on the C side."""
    import json
    from sepfwi import utils as ft
    a, b = str(tmp_path / "a.json"), str(tmp_path / "b.json")
    ft.paraGen(96, 80, 10.0, 10.0, 100, 1e-3, 10.0, 10, 12, a, "s.json", str(tmp_path / "D"))
    ft.paraGen(96, 80, 10.0, 10.0, 100, 1e-3, 10.0, 10, 12, b, "s.json", str(tmp_path / "D"), das_fiber="vertical")
    assert "das_fiber" not in json.load(open(a))
    assert json.load(open(b))["das_fiber"] == "vertical"
    with pytest.raises(ValueError):
        ft.paraGen(96, 80, 10.0, 10.0, 100, 1e-3, 10.0, 10, 12, a, "s.json", str(tmp_path / "D"), das_fiber="diagonal")


@pytest.mark.parametrize("N,nPml,dh,f0,dt", [(2064, 32, 10.0, 10.0, 1e-3), (1064, 32, 10.0, 10.0, 1e-3), (80, 10, 10.0, 25.0, 1e-3), (265, 32, 20.0, 10.0, 2e-3)])
def test_cpml_profiles_are_trivial_outside_the_layers(N, nPml, dh, f0, dt):
    """The kernels skip the a-terms and the 1/K loads outside `x < nPml or x > N-nPml-1`: K must be exactly 1 and a exactly 0
    there, for the integer and the half-grid profiles (cpmlInit, utilities.cu:243-359)."""
    from sepfwi import _native
    L = _native.lib()
    arrs = [np.zeros(N, np.float32) for _ in range(6)]
    _native.check(L.sepfwi_cpml_profiles(*[a.ctypes.data for a in arrs], N, nPml, dh, f0, dt))
    K, a, b, Kh, ah, bh = arrs
    inside = np.arange(N)
    layer = (inside < nPml) | (inside > N - nPml - 1)
    for prof, one in ((K, 1.0), (Kh, 1.0), (a, 0.0), (ah, 0.0)):
        assert np.all(prof[~layer] == one)
    assert np.any(K[layer] != 1.0) and np.any(a[layer] != 0.0)


def test_bench_launcher_command_line():
    """`python bench.py --gpus N` with no WORLD_SIZE turns itself into the N-rank job: the driver's own command line."""
    import bench
    cmd = bench.launcher_cmd(4, ["--gpus", "4", "--steps", "3", "--warmup", "1"], port=29517)
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    assert cmd[-7].endswith("bench.py") and cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    free = bench.launcher_cmd(2, [])
    assert 1024 < int(free[free.index("--master-port") + 1]) < 65536


def _module_case(tmp_path, cls_name, fields, monkeypatch):
    import sepfwi.ops as ops
    from sepfwi import modules as M
    fake = _FakeOps()
    monkeypatch.setattr(ops, "fwi_ops", fake)
    pb = P.make_problem(str(tmp_path), nz=12, nx=14, nPml=4, nSteps=10, nshots=1, nPad=3)
    params = [torch.tensor(np.ascontiguousarray(f, dtype=np.float32), requires_grad=True) for f in fields]
    mask = torch.zeros(pb["nz_pad"], pb["nx_pad"]); mask[4:16, 4:18] = 1.0
    fwi = getattr(M, cls_name)(params[0], params[1], params[2], pb["Stf"], pb["opt"], Mask=mask)
    loss = fwi(pb["Shot_ids"], ngpu=1)
    loss.backward()
    lam, mu, den = [t[4:16, 4:18].numpy().astype(np.float64) for t in fake.calls[0][:3]]
    return pb, fwi, lam, mu, den


def test_impedance_velocity_modules(monkeypatch, tmp_path):
    """FWI_Vp_Vs_IP / FWI_Vp_Vs_IS (FWI_ops.py:270-393): Lame map and chain rule with gLambda = 2, gMu = -1, gDen = 0.5
    (_FakeOps) against the formulas written out by hand."""
    pb = P.make_problem(str(tmp_path / "m"), nz=12, nx=14, nPml=4, nSteps=10, nshots=1, nPad=3)
    vp, vs, rho = [pb["init"][k].astype(np.float64) for k in ("vp", "vs", "rho")]
    gl, gm, gd = 2.0, -1.0, 0.5
    ip = rho * vp
    _, fwi, lam, mu, den = _module_case(tmp_path / "ip", "FWI_Vp_Vs_IP", (vp, vs, ip), monkeypatch)
    assert [n for n, _ in fwi.named_parameters()] == ["Vp", "Vs", "IP"]
    np.testing.assert_allclose(den, rho, rtol=1e-6)
    np.testing.assert_allclose(mu, rho * vs ** 2, rtol=1e-6)
    np.testing.assert_allclose(lam, rho * (vp ** 2 - 2 * vs ** 2), rtol=1e-5)
    np.testing.assert_allclose(fwi.IP.grad.numpy(), gl * (vp - 2 * vs ** 2 / vp) + gm * vs ** 2 / vp + gd / vp, rtol=1e-5)
    np.testing.assert_allclose(fwi.Vs.grad.numpy(), gl * (-4 * ip * vs / vp) + gm * 2 * ip * vs / vp, rtol=1e-5)
    np.testing.assert_allclose(fwi.Vp.grad.numpy(), gl * (ip + 2 * ip * vs ** 2 / vp ** 2) - gm * ip * vs ** 2 / vp ** 2 - gd * ip / vp ** 2, rtol=1e-5)
    is_ = rho * vs
    _, fwi, lam, mu, den = _module_case(tmp_path / "is", "FWI_Vp_Vs_IS", (vp, vs, is_), monkeypatch)
    assert [n for n, _ in fwi.named_parameters()] == ["Vp", "Vs", "IS"]
    np.testing.assert_allclose(den, rho, rtol=1e-6)
    np.testing.assert_allclose(mu, rho * vs ** 2, rtol=1e-6)
    np.testing.assert_allclose(lam, rho * (vp ** 2 - 2 * vs ** 2), rtol=1e-5)
    np.testing.assert_allclose(fwi.IS.grad.numpy(), gl * (vp ** 2 / vs - 2 * vs) + gm * vs + gd / vs, rtol=1e-5)
    np.testing.assert_allclose(fwi.Vp.grad.numpy(), gl * 2 * is_ * vp / vs, rtol=1e-5)
    np.testing.assert_allclose(fwi.Vs.grad.numpy(), gl * (-is_ * vp ** 2 / vs ** 2 - 2 * is_) + gm * is_ - gd * is_ / vs ** 2, rtol=1e-5)


def test_rock_physics_maps_match_reference_vectors(monkeypatch, tmp_path):
    """FWI_Rock_Physics_VRH / _gassmann (FWI_ops.py:401-619) and utils.pcs2dv_* against golden vectors produced by importing
    the reference's own fwi_utils.py (scripts/make_golden_rockphysics.py): (phi, cc, sw) -> vp, vs, rho."""
    from conftest import GOLDEN
    from sepfwi import utils as ft
    G = np.load(os.path.join(GOLDEN, "rock_physics.npz"))
    phi, cc, sw = G["phi"], G["cc"], G["sw"]
    for name, fn in (("vrh", ft.pcs2dv_vrh), ("gas", ft.pcs2dv_gassmann)):
        vp, vs, rho = fn(phi, cc, sw)
        for got, key in ((vp, "vp"), (vs, "vs"), (rho, "rho")):
            np.testing.assert_allclose(got, G["%s_%s" % (name, key)], rtol=1e-13)
    # the modules' Lame maps give the same media: lambda = rho (vp^2 - 2 vs^2), mu = rho vs^2 [MPa]
    from sepfwi import modules as M
    pb = P.make_problem(str(tmp_path), nz=12, nx=17, nPml=4, nSteps=10, nshots=1, nPad=3)
    T = lambda a: torch.tensor(a, dtype=torch.float64)
    for cls, name in ((M.FWI_Rock_Physics_VRH, "vrh"), (M.FWI_Rock_Physics_gassmann, "gas")):
        lam, mu, den = cls.lame(None, T(phi), T(cc), T(sw))
        vp, vs, rho = G[name + "_vp"], G[name + "_vs"], G[name + "_rho"]
        np.testing.assert_allclose(den.numpy(), rho, rtol=1e-12)
        np.testing.assert_allclose(mu.numpy(), rho * vs ** 2 / 1e6, rtol=1e-10)
        np.testing.assert_allclose(lam.numpy(), rho * (vp ** 2 - 2 * vs ** 2) / 1e6, rtol=1e-9)
    # and they are differentiable modules with the reference's parameter names
    import sepfwi.ops as ops
    monkeypatch.setattr(ops, "fwi_ops", _FakeOps())
    P32 = lambda a: torch.tensor(a.astype(np.float32), requires_grad=True)
    fwi = M.FWI_Rock_Physics_gassmann(P32(phi), P32(cc), P32(sw), pb["Stf"], pb["opt"])
    fwi(pb["Shot_ids"], ngpu=1).backward()
    assert [n for n, _ in fwi.named_parameters()] == ["PHI", "CC", "SW"]
    assert all(torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0 for p in fwi.parameters())


def test_wrong_shapes_are_refused_before_the_library_sees_them(tmp_path):
    """The C ABI carries no sizes (like the reference's raw data_ptr<float>() hand-over, Src/Torch_Fwi.cpp:55-58): the wrapper
    checks tensors against the parameter file, so a wrong shape is a ValueError and not an out-of-bounds read or write."""
    from sepfwi import fwi_ops
    pb = P.make_problem(str(tmp_path), nSteps=40, nshots=2)
    lam, mu, den = pb["lame_init"]
    ok = dict(Lambda=lam, Mu=mu, Den=den, Stf=pb["Stf"], ids=pb["Shot_ids"])
    bad = [dict(ok, Lambda=lam[:-1], Mu=mu[:-1], Den=den[:-1]),              # model of another grid
           dict(ok, Mu=mu[:, :-1].contiguous()),                             # three different shapes
           dict(ok, Stf=pb["Stf"][:, :-3].contiguous()),                     # gStf would be written past its end
           dict(ok, ids=torch.tensor([0, 2], dtype=torch.int32)),            # row 2 of a two-row Stf
           dict(ok, ids=torch.tensor([-1], dtype=torch.int32))]
    for a in bad:
        with pytest.raises(ValueError):
            fwi_ops._cufd(1, 0, a["Lambda"], a["Mu"], a["Den"], a["Stf"], a["ids"], pb["para_fname"])
    with pytest.raises(TypeError):
        fwi_ops._cufd(1, 0, lam.double(), mu, den, pb["Stf"], pb["Shot_ids"], pb["para_fname"])


def test_packed_observed_file_round_trip(tmp_path):
    """utils.pack_observed / read_packed_gather: the pack holds exactly the bytes of the Shot_ett files."""
    from sepfwi import utils as ft
    rng = np.random.default_rng(1)
    d = str(tmp_path)
    nS = 37
    g = {sid: rng.standard_normal((5 + sid, nS)).astype(np.float32) for sid in (0, 3, 4)}
    for sid, a in g.items():
        a.tofile(os.path.join(d, "Shot_ett%d.bin" % sid))
    pack = ft.pack_observed(d, [4, 0, 3], nS, os.path.join(d, "all.pack"))
    for sid, a in g.items():
        assert np.array_equal(ft.read_packed_gather(pack, sid), a)
    assert os.path.getsize(pack) == 16 + 3 * 16 + sum(a.size * 4 for a in g.values())
    with pytest.raises(KeyError):
        ft.read_packed_gather(pack, 1)
    para = os.path.join(d, "p.json")
    ft.paraGen(64, 64, 10.0, 10.0, nS, 1e-3, 10.0, 8, 0, para, os.path.join(d, "s.json"), os.path.join(d, "Data"), obs_pack_fname=pack)
    assert json.load(open(para))["obs_pack_fname"] == pack


def test_elastic_propagator_operands_and_chain_rule(monkeypatch, tmp_path):
    """The OO caller (propagator.py:57-226): moduli WITHOUT the 1e-6 (km/s velocities), replicate padding, one Ricker row per
    source, the survey written as given, and the hand-written chain rule to (vp, vs, rho) on the cropped gradients."""
    import sepfwi.ops as ops
    from sepfwi.propagator import ElasticPropagator, Model, Propagator, Survey
    fake = _FakeOps()
    fake.obscalc = lambda *a: fake.calls.append(("obscalc",) + a)
    monkeypatch.setattr(ops, "fwi_ops", fake)
    rng = np.random.default_rng(3)
    nz, nx, nPml, nt = 12, 14, 4, 10
    vp = (3.0 + 0.2 * rng.random((nz, nx))).astype(np.float32)
    vs = (vp / 1.8).astype(np.float32)
    rho = (2400.0 + 50.0 * rng.random((nz, nx))).astype(np.float32)
    model = Model(nx, nz, 10.0, 10.0, nt, 1e-3, nPml, vp, vs, rho, str(tmp_path))
    survey = Survey(20.0, np.array([6, 12]), np.array([5, 5]), np.arange(5, 16), np.full(11, 13))
    with pytest.raises(NotImplementedError):
        Propagator(model, survey).apply_forward(None)
    prop = ElasticPropagator(model, survey)
    prop.apply_forward(ngpu=1)
    tag, lam, mu, den, stf, ngpu, ids, pf = fake.calls.pop()
    nPad = 32 - (nz + 2 * nPml) % 32
    assert tag == "obscalc" and lam.shape == (nz + 2 * nPml + nPad, nx + 2 * nPml) and stf.shape == (2, nt) and ids.tolist() == [0, 1]
    assert torch.allclose(lam[nPml:nPml + nz, nPml:nPml + nx], torch.tensor(rho * (vp ** 2 - 2 * vs ** 2)))
    assert torch.equal(lam[0, :], lam[nPml, :]) and torch.equal(mu[:, 0], mu[:, nPml]) and torch.equal(den[-1, :], den[nPml + nz - 1, :])
    para, srv = json.load(open(pf)), json.load(open(os.path.join(str(tmp_path), "survey_file.json")))
    assert para["nz"] == lam.shape[0] and para["nPad"] == nPad and para["f0"] == 20.0 and srv["nShots"] == 2
    assert srv["shot1"]["x_src"] == 12 and srv["shot0"]["z_rec"] == [13] * 11
    misfit, gvp, gvs, grho, gstf = prop.apply_gradient(Model(nx, nz, 10.0, 10.0, nt, 1e-3, nPml, vp * 1.01, vs, rho, str(tmp_path)))
    assert misfit.shape == (1,) and float(misfit[0]) == 3.5 and gvp.shape == (nz, nx) and gstf.shape == (2, nt)
    vp1 = vp * np.float32(1.01)       # gL = 2, gM = -1, gD = 0.5 from the fake operator
    np.testing.assert_allclose(gvp, 2 * rho * vp1 * 2.0, rtol=1e-6)
    np.testing.assert_allclose(gvs, -4 * rho * vs * 2.0 + 2 * rho * vs * -1.0, rtol=1e-6)
    np.testing.assert_allclose(grho, (vp1 ** 2 - 2 * vs ** 2) * 2.0 + vs ** 2 * -1.0 + 0.5, rtol=1e-5)


def test_lean_lbfgsb_driver_gives_scipys_iterates_bit_for_bit():
    """sepfwi.obj_wrapper.minimize_lbfgsb drives SciPy's compiled L-BFGS-B routine itself (vectorised bound codes instead of
    optimize.minimize's per-element Python loops): same iterates, evaluations, message -- with every kind of bound in play."""
    from scipy import optimize
    from sepfwi.obj_wrapper import minimize_lbfgsb
    rng = np.random.default_rng(0)
    n = 60

    def fun(x):
        return float(np.sum(100 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2))

    def jac(x):
        g = np.zeros_like(x)
        g[:-1] = -400 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1])
        g[1:] += 200 * (x[1:] - x[:-1] ** 2)
        return g

    x0 = rng.uniform(-2, 2, n)
    lb, ub = np.full(n, -np.inf), np.full(n, np.inf)
    lb[::3] = -0.5; ub[1::3] = 0.8; lb[2::6] = 0.1; ub[2::6] = 0.7      # lower-only, upper-only, boxed, free
    B = optimize.Bounds(lb, ub)
    for opts in (dict(maxiter=40, maxcor=5, ftol=1e-12, gtol=1e-16, maxfun=1500, maxls=20), dict(maxiter=500, maxcor=7, ftol=1e-9, gtol=1e-6, maxfun=60, maxls=20)):
        h1, h2 = [], []
        r1 = optimize.minimize(fun, x0, method="L-BFGS-B", jac=jac, bounds=B, callback=lambda x: h1.append(x.copy()), options=opts)
        r2 = minimize_lbfgsb(fun, x0, jac, bounds=B, callback=lambda x: h2.append(x.copy()), **opts)
        assert (r1.nit, r1.nfev, r1.status, r1.message, r1.fun) == (r2.nit, r2.nfev, r2.status, r2.message, r2.fun)
        assert len(h1) == len(h2) and all(np.array_equal(a, b) for a, b in zip(h1, h2))
        assert np.array_equal(r1.x, r2.x) and np.array_equal(r1.jac, r2.jac) and np.all(r2.x >= lb) and np.all(r2.x <= ub)
        v = rng.standard_normal(n)
        assert np.array_equal(r1.hess_inv.matvec(v), r2.hess_inv.matvec(v))
    def stop(x):
        raise StopIteration
    assert minimize_lbfgsb(fun, x0, jac, bounds=B, callback=stop).nit == 1
    assert minimize_lbfgsb(fun, x0, jac, bounds=None, maxiter=5).nit == 5          # no bounds: the public route
    with pytest.raises(ValueError):
        minimize_lbfgsb(fun, x0, jac, bounds=optimize.Bounds(np.ones(n), np.zeros(n)))


def test_only_some_fields_inverted(monkeypatch, tmp_path):
    """The rock-physics experiments invert ONE of the three fields (Main-004-FWI-Rock-Physics.py:117-119: PHI and CC without
    requires_grad, SW with): the fixed ones are plain attributes, not parameters, SciPy's vector holds the inverted field only,
    and its gradient is the corresponding block of the all-fields gradient."""
    import sepfwi.ops as ops
    from sepfwi import modules as M
    from sepfwi.obj_wrapper import PyTorchObjective
    monkeypatch.setattr(ops, "fwi_ops", _FakeOps())
    pb = P.make_problem(str(tmp_path), nz=12, nx=14, nPml=4, nSteps=10, nshots=1, nPad=3)
    T = lambda k, g: torch.tensor(pb["init"][k], requires_grad=g)
    full = M.FWI(T("vp", True), T("vs", True), T("rho", True), pb["Stf"], pb["opt"])
    full(pb["Shot_ids"], ngpu=1).backward()
    for which in ("vp", "vs", "rho"):
        part = M.FWI(T("vp", which == "vp"), T("vs", which == "vs"), T("rho", which == "rho"), pb["Stf"], pb["opt"])
        name = {"vp": "Vp", "vs": "Vs", "rho": "Den"}[which]
        assert [n for n, _ in part.named_parameters()] == [name]
        obj = PyTorchObjective(part, lambda: part(pb["Shot_ids"], ngpu=1))
        assert obj.x0.size == 12 * 14
        fun, jac = obj.fun, obj.jac
        assert fun(obj.x0) == 3.5
        g = jac(obj.x0).reshape(12, 14)
        np.testing.assert_allclose(g, getattr(full, name).grad.numpy().astype(np.float64), rtol=1e-6)


def test_coarse_parameter_grid_is_interpolated(monkeypatch, tmp_path):
    """opt["nz_orig"], opt["nx_orig"] smaller than the modelling grid: the parameters live on the coarse grid and are resized
    bilinearly before the replicate padding (fwi_utils.py:31-44; the reference's scripts always set them equal).  The operator
    sees the padded fine grid, the gradient comes back on the coarse one, and the one-launch maps stand aside."""
    import sepfwi.ops as ops
    from sepfwi import modules as M
    fake = _FakeOps()
    monkeypatch.setattr(ops, "fwi_ops", fake)
    pb = P.make_problem(str(tmp_path), nz=12, nx=14, nPml=4, nSteps=10, nshots=1, nPad=3)
    opt = dict(pb["opt"], nz_orig=6, nx_orig=7)
    coarse = lambda k: torch.tensor(np.ascontiguousarray(pb["init"][k][::2, ::2]), requires_grad=True)
    fwi = M.FWI(coarse("vp"), coarse("vs"), coarse("rho"), pb["Stf"], opt)
    assert not fwi._fusable()
    loss = fwi(pb["Shot_ids"], ngpu=1)
    loss.backward()
    lam = fake.calls[0][0]
    assert lam.shape == (pb["nz_pad"], pb["nx_pad"]) and fwi.Vp.grad.shape == (6, 7) and torch.isfinite(fwi.Vp.grad).all()
    # a constant field stays that constant on the fine grid
    c = lambda v: torch.full((6, 7), v, requires_grad=True)
    fwi2 = M.FWI(c(3000.0), c(1700.0), c(2400.0), pb["Stf"], opt)
    fwi2(pb["Shot_ids"], ngpu=1)
    assert torch.allclose(fake.calls[1][2], torch.full((pb["nz_pad"], pb["nx_pad"]), 2400.0))


def test_header_is_valid_c_and_cpp(tmp_path):
    """include/sepfwi.h compiles as C99 and as C++17 (the reference's shim, Src/Torch_Fwi.cpp, is C++: INTEGRATION.md option A), warnings
    as errors, and a C++ caller written as that document shows type-checks against it."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None or shutil.which("g++") is None:
        pytest.skip("no host compiler")
    inc = os.path.join(ROOT, "include")
    c = tmp_path / "t.c"
    c.write_text('#include "sepfwi.h"\nint main(void) { return sepfwi_version() > 0 ? 0 : 1; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-I", inc, str(c)])
    cpp = tmp_path / "t.cpp"
    cpp.write_text('#include <stdexcept>\n#include <string>\n#include "sepfwi.h"\n'
                   'float call(float *gL, float *gM, float *gD, float *gS, const float *L, const float *M, const float *D, const float *stf,\n'
                   '           int gpu, int n, const int *ids, const std::string &para) {\n'
                   '    float misfit = 0.0f;\n'
                   '    if (sepfwi_cufd(&misfit, gL, gM, gD, gS, L, M, D, stf, 1, gpu, n, ids, para.c_str()) != 0)\n'
                   '        throw std::runtime_error(sepfwi_last_error());\n'
                   '    return misfit;\n}\n')
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", inc, str(cpp)])


def test_collective_record_without_a_process_group():
    """dist.collective_stats() outside torch.distributed: one rank, no backend, nothing counted (bench.py prints "rccl" only for N > 1)."""
    from sepfwi import dist
    cs = dist.collective_stats(reset=True)
    assert cs["ranks"] == 1 and cs["backend"] is None and cs["calls"] == 0 and cs["allreduce_ms"] is None and cs["staged"] == 0


def test_paragen_writes_the_conditioning_switch_only_when_asked(tmp_path):
    """Default parameter files stay byte-identical to the reference's schema; "conditioning" appears only when given, and only the two
    legal values are accepted."""
    from sepfwi import utils as ft
    a, b = str(tmp_path / "a.json"), str(tmp_path / "b.json")
    ft.paraGen(96, 80, 10.0, 10.0, 100, 1e-3, 10.0, 10, 4, a, "s.json", str(tmp_path / "D"))
    ft.paraGen(96, 80, 10.0, 10.0, 100, 1e-3, 10.0, 10, 4, b, "s.json", str(tmp_path / "D"), filter_para=[1, 2, 3, 4], conditioning="reference")
    ja, jb = json.load(open(a)), json.load(open(b))
    assert "conditioning" not in ja and jb["conditioning"] == "reference" and jb["filter"] == [1, 2, 3, 4]
    with pytest.raises(ValueError):
        ft.paraGen(96, 80, 10.0, 10.0, 100, 1e-3, 10.0, 10, 4, a, "s.json", str(tmp_path / "D"), conditioning="sometimes")
