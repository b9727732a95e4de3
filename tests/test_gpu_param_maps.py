"""Fused parameterisation maps (csrc/param_maps.hip) on the GPU against the torch expressions of the
reference's modules (FWI_ops.py:116-127,194-204,256-266,319-330,381-393) evaluated on the CPU: forward maps bit for bit
(same float32 operation order, no contraction), chain rule to rounding (the padding transpose sums in another order)."""
import numpy as np
import pytest
import torch

import problems as P

pytestmark = pytest.mark.gpu

CLASSES = ["FWI", "FWI_Lame_Den", "FWI_IP_IS_Den", "FWI_Vp_Vs_IP", "FWI_Vp_Vs_IS"]


def _fields(cls_name, pb):
    vp, vs, rho = [pb["true"][k].astype(np.float64) for k in ("vp", "vs", "rho")]
    if cls_name == "FWI":
        return vp, vs, rho
    if cls_name == "FWI_Lame_Den":
        return rho * (vp ** 2 - 2 * vs ** 2) / 1e6, rho * vs ** 2 / 1e6, rho
    if cls_name == "FWI_IP_IS_Den":
        return vp / 1e3 * rho, vs / 1e3 * rho, rho
    if cls_name == "FWI_Vp_Vs_IP":
        return vp, vs, rho * vp
    return vp, vs, rho * vs


@pytest.mark.parametrize("cls_name", CLASSES)
def test_fused_maps_equal_torch_expressions(tmp_path, hip_ops, cls_name):
    from sepfwi import modules as M
    pb = P.make_problem(str(tmp_path), nz=37, nx=70, nPml=9, nSteps=10, nshots=1, nPad=5)
    rng = np.random.default_rng(3)
    mask = (rng.uniform(size=(pb["nz_pad"], pb["nx_pad"])) > 0.3).astype(np.float32)
    mask[20:30, 15:40] = rng.uniform(0.1, 0.9, (10, 25))          # fractional blending too
    mask[:, :5] = 1.0; mask[-7:, :] = 1.0                          # padding strips that DO depend on the edge cells
    gl, gm, gd = [torch.tensor(rng.standard_normal((pb["nz_pad"], pb["nx_pad"])).astype(np.float32)) for _ in range(3)]
    outs = {}
    for dev in ("cpu", "cuda"):
        f = [torch.tensor(a.astype(np.float32), device=dev, requires_grad=True) for a in _fields(cls_name, pb)]
        # the reference model differs from the current one (as after some iterations)
        ref = [torch.tensor((a * 1.03).astype(np.float32), device=dev) for a in _fields(cls_name, pb)]
        mod = getattr(M, cls_name)(ref[0], ref[1], ref[2], pb["Stf"], pb["opt"], Mask=torch.tensor(mask, device=dev))
        for n, t in zip(mod.NAMES, f):
            setattr(mod, n, torch.nn.Parameter(t))
        assert mod._fusable() == (dev == "cuda")
        lam, mu, den = mod.lame_padded()
        (lam * gl.to(dev)).sum().add((mu * gm.to(dev)).sum()).add((den * gd.to(dev)).sum()).backward()
        outs[dev] = [t.detach().cpu().numpy() for t in (lam, mu, den)] + [getattr(mod, n).grad.cpu().numpy() for n in mod.NAMES]
    for k in range(3):
        assert np.array_equal(outs["cuda"][k], outs["cpu"][k]), (cls_name, "forward", k)
    for k in range(3, 6):
        scale = np.abs(outs["cpu"][k]).max()
        assert np.abs(outs["cuda"][k] - outs["cpu"][k]).max() <= 2e-6 * scale, (cls_name, "backward", k)


@pytest.mark.parametrize("cls_name", ["FWI_Rock_Physics_VRH", "FWI_Rock_Physics_gassmann"])
def test_fused_rock_physics_maps_equal_torch_expressions(tmp_path, hip_ops, cls_name):
    """Kinds 5 and 6 of csrc/param_maps.hip (FWI_ops.py:451-497 Voigt-Reuss-Hill, :567-611 Biot-Gassmann): pad + mask blend +
    rock-physics map in one launch and its whole chain rule in one, against the torch expressions on CPU tensors (the
    reference's arrangement).  VRH forward: bit for bit (same float32 operations in the same order).  Gassmann forward: to two
    units in the last place of the velocities -- it takes two square roots, and torch's vectorised CPU sqrt is itself not
    correctly rounded (0.7 % of its results differ from IEEE sqrt by one ulp), so there is no bit pattern to match.  Chain rule:
    reverse-mode over the same operation list as autograd, <= 1e-6 of the largest entry (measured 1.5e-7 / 4.2e-7)."""
    from sepfwi import modules as M
    pb = P.make_problem(str(tmp_path), nz=37, nx=70, nPml=9, nSteps=10, nshots=1, nPad=5)
    rng = np.random.default_rng(11)
    mask = (rng.uniform(size=(pb["nz_pad"], pb["nx_pad"])) > 0.3).astype(np.float32)
    mask[20:30, 15:40] = rng.uniform(0.1, 0.9, (10, 25))
    mask[:, :5] = 1.0; mask[-7:, :] = 1.0
    fields = [P.smooth_random(rng, (37, 70), lo, hi).astype(np.float32) for lo, hi in ((0.10, 0.30), (0.05, 0.45), (0.2, 0.9))]
    gl, gm, gd = [torch.tensor(rng.standard_normal((pb["nz_pad"], pb["nx_pad"])).astype(np.float32)) for _ in range(3)]
    outs = {}
    for dev in ("cpu", "cuda"):
        f = [torch.tensor(a, device=dev, requires_grad=True) for a in fields]
        ref = [torch.tensor((a * 1.03).astype(np.float32), device=dev) for a in fields]
        mod = getattr(M, cls_name)(ref[0], ref[1], ref[2], pb["Stf"], pb["opt"], Mask=torch.tensor(mask, device=dev))
        for n, t in zip(mod.NAMES, f):
            setattr(mod, n, torch.nn.Parameter(t))
        assert mod._fusable() == (dev == "cuda")
        lam, mu, den = mod.lame_padded()
        (lam * gl.to(dev)).sum().add((mu * gm.to(dev)).sum()).add((den * gd.to(dev)).sum()).backward()
        outs[dev] = [t.detach().cpu().numpy() for t in (lam, mu, den)] + [getattr(mod, n).grad.cpu().numpy() for n in mod.NAMES]
    if cls_name.endswith("VRH"):
        for k in range(3):
            assert np.array_equal(outs["cuda"][k], outs["cpu"][k]), (cls_name, "forward", k)
    else:
        lam_c, mu_c, den_c = outs["cpu"][:3]
        assert np.array_equal(outs["cuda"][2], den_c)                      # the density has no root in it
        np.testing.assert_allclose(outs["cuda"][1], mu_c, rtol=5e-7)       # rho vs^2: two ulp of vs
        # Lambda = rho (vp^2 - 2 vs^2) / 1e6: two ulp of each velocity, measured against the minuend
        m_c = den_c.astype(np.float64) * (lam_c.astype(np.float64) / den_c + 2 * mu_c.astype(np.float64) / den_c)
        assert np.abs(outs["cuda"][0] - lam_c).max() <= 1e-6 * np.abs(m_c).max()
    tol = 1e-6      # measured: 1.5e-7 (VRH), 4.2e-7 (Gassmann)
    print("%s: fused vs torch-on-CPU, forward max rel. deviation (Lambda, Mu, Den) %s; chain rule max deviation / max entry %s" % (
        cls_name, ["%.1e" % (np.abs(outs["cuda"][k] - outs["cpu"][k]).max() / np.abs(outs["cpu"][k]).max()) for k in range(3)],
        ["%.1e" % (np.abs(outs["cuda"][k] - outs["cpu"][k]).max() / np.abs(outs["cpu"][k]).max()) for k in range(3, 6)]))
    for k in range(3, 6):
        scale = np.abs(outs["cpu"][k]).max()
        assert scale > 0 and np.abs(outs["cuda"][k] - outs["cpu"][k]).max() <= tol * scale, (cls_name, "backward", k, np.abs(outs["cuda"][k] - outs["cpu"][k]).max() / scale)


def test_fused_chain_on_gpu_equals_the_reference_chain_on_cpu_tensors(tmp_path, oracle, hip_ops):
    """The whole iteration (module -> FWIFunction -> HIP propagator -> chain rule): everything resident in HBM with the
    fused maps, against the reference's arrangement -- torch expressions on CPU tensors, model staged over PCIe.  The
    two media are bit-identical, so misfit and raw gradients are too; only the chain rule's summation order differs."""
    from sepfwi import modules as M
    pb = P.make_problem(str(tmp_path), hetero=True, nSteps=150, nshots=2)
    lt, mt, dt_ = pb["lame_true"]
    hip_ops.obscalc(lt, mt, dt_, pb["Stf"], 1, pb["Shot_ids"], pb["para_fname"])
    mask = torch.zeros(pb["nz_pad"], pb["nx_pad"]); mask[pb["nPml"] + 3:pb["nPml"] + 44, pb["nPml"]:pb["nPml"] + 60] = 1.0
    res = {}
    for dev in ("cuda", "cpu"):
        f = [torch.tensor(pb["init"][k], device=dev, requires_grad=True) for k in ("vp", "vs", "rho")]
        fwi = M.FWI(f[0], f[1], f[2], pb["Stf"], pb["opt"], Mask=mask.to(dev))
        assert fwi._fusable() == (dev == "cuda")
        loss = fwi(pb["Shot_ids"], ngpu=1)
        loss.backward()
        res[dev] = [float(loss.detach())] + [p.grad.cpu().numpy() for p in fwi.parameters()]
    assert res["cuda"][0] == res["cpu"][0]
    for a, b in zip(res["cuda"][1:], res["cpu"][1:]):
        assert np.abs(a - b).max() <= 2e-6 * np.abs(b).max()


def test_fused_maps_at_headline_size_and_timing(hip_ops):
    """2000 x 1000: one launch each way instead of the torch chain; prints both times."""
    import time
    from sepfwi import modules as M
    from sepfwi import utils as ft
    nz, nx, nPml = 1000, 2000, 32
    nPad = ft.nPad_for(nz, nPml)
    opt = dict(nz=nz, nx=nx, nz_orig=nz, nx_orig=nx, nPml=nPml, nPad=nPad, para_fname="unused.json")
    g = torch.Generator().manual_seed(0)
    vp = (3000.0 + 500.0 * torch.rand(nz, nx, generator=g)).cuda()
    vs, rho = (vp / 1.8).contiguous(), (2000.0 + 0.2 * vp).contiguous()
    times = {}
    grads = {}
    for fused in (True, False):
        M.USE_FUSED_MAPS = fused
        try:
            f = [t.clone().requires_grad_(True) for t in (vp, vs, rho)]
            fwi = M.FWI(f[0], f[1], f[2], None, opt)
            for rep in range(3):
                for p in fwi.parameters():
                    p.grad = None
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                lam, mu, den = fwi.lame_padded()
                (lam.sum() + 0.5 * mu.sum() + den.sum()).backward()
                torch.cuda.synchronize()
                times[fused] = time.perf_counter() - t0
            grads[fused] = [p.grad.clone() for p in fwi.parameters()]
        finally:
            M.USE_FUSED_MAPS = True
    print("pad + mask + Lame map + chain rule at 2000x1000: fused %.2f ms, torch ops %.2f ms" % (1e3 * times[True], 1e3 * times[False]))
    # all-ones mask: the corner cells sum ~1900 padded cells each way; torch's GPU kernels also divide by 1e6 through a
    # reciprocal multiply -- agreement to float32 summation noise
    dev = [float((a - b).abs().max()) / float(b.abs().max()) for a, b in zip(grads[True], grads[False])]
    print("max deviation fused vs torch ops, relative to the largest gradient entry:", dev)
    assert max(dev) <= 1e-4


@pytest.mark.parametrize("cls_name,key", [("FWI_Rock_Physics_VRH", "vrh"), ("FWI_Rock_Physics_gassmann", "gas")])
def test_rock_physics_modules_on_hip_tensors(tmp_path, hip_ops, cls_name, key):
    """SURVEY.md 8f-1 remainder: the two rock-physics parameterisations (FWI_ops.py:401-619) with every tensor in HBM.
    (i) Their (phi, cc, sw) -> Lambda, Mu, Den maps on HIP tensors against the golden vectors generated by importing the
    reference's fwi_utils.py (tests/golden/rock_physics.npz), float32 accuracy.  (ii) A whole iteration -- module ->
    FWIFunction -> HIP propagator -> autograd chain rule -- on HIP tensors against the same chain on the reference's CPU
    tensors: misfit within 1e-5; d/d(phi, cc, sw) of both against the float64 chain rule (<= 5e-4).  On HIP tensors the module runs the fused maps (kinds 5 / 6, one launch
    each way; test_fused_rock_physics_maps_equal_torch_expressions compares them with the torch expressions)."""
    import os
    from conftest import GOLDEN
    from sepfwi import modules as M
    G = np.load(os.path.join(GOLDEN, "rock_physics.npz"))
    cls = getattr(M, cls_name)
    phi, cc, sw = [torch.tensor(G[k].astype(np.float32), device="cuda") for k in ("phi", "cc", "sw")]
    lam, mu, den = cls.lame(None, phi, cc, sw)
    vp, vs, rho = G[key + "_vp"], G[key + "_vs"], G[key + "_rho"]
    assert lam.is_cuda and mu.is_cuda and den.is_cuda
    np.testing.assert_allclose(den.cpu().numpy(), rho, rtol=2e-6)
    np.testing.assert_allclose(mu.cpu().numpy(), rho * vs ** 2 / 1e6, rtol=2e-5)
    np.testing.assert_allclose(lam.cpu().numpy(), rho * (vp ** 2 - 2 * vs ** 2) / 1e6, rtol=2e-4)   # a difference of two large moduli in float32

    # whole iteration: smooth rock-property fields on the test grid, observed data from a perturbed set
    pb = P.make_problem(str(tmp_path), nz=44, nx=60, nPml=10, nSteps=200, nshots=2, hetero=True)
    rng = np.random.default_rng(5)
    f0 = [P.smooth_random(rng, (44, 60), lo, hi).astype(np.float32) for lo, hi in ((0.10, 0.30), (0.05, 0.45), (0.2, 0.9))]
    f1 = [a.copy() for a in f0]
    f1[0][18:26, 20:30] *= 1.08
    f1[2][24:32, 35:45] *= 0.9
    from sepfwi import utils as ft
    pad = lambda a: torch.tensor(ft.padding_numpy_array(a, pb["nPml"], pb["nPad"]))
    lam_t, mu_t, den_t = cls.lame(None, *[pad(a) for a in f1])
    hip_ops.obscalc(lam_t.contiguous(), mu_t.contiguous(), den_t.contiguous(), pb["Stf"], 1, pb["Shot_ids"], pb["para_fname"])
    mask = torch.zeros(pb["nz_pad"], pb["nx_pad"])
    mask[pb["nPml"] + 3:pb["nPml"] + 44, pb["nPml"]:pb["nPml"] + 60] = 1.0
    res = {}
    for dev in ("cuda", "cpu"):
        prm = [torch.tensor(a, device=dev, requires_grad=True) for a in f0]
        fwi = cls(prm[0], prm[1], prm[2], pb["Stf"], pb["opt"], Mask=mask.to(dev))
        loss = fwi(pb["Shot_ids"], ngpu=1)
        loss.backward()
        assert all(p.grad.device.type == dev for p in fwi.parameters())
        res[dev] = [float(loss.detach())] + [p.grad.cpu().numpy() for p in fwi.parameters()]
    assert res["cpu"][0] > 0
    assert abs(res["cuda"][0] - res["cpu"][0]) <= 1e-5 * res["cpu"][0], (res["cuda"][0], res["cpu"][0])
    # yardstick: the same chain rule in float64 (the raw float32 gradients of the propagator pulled back through a float64 copy
    # of the map).  torch's CPU and GPU float32 kernels round these cancellation-prone expressions differently (the Gassmann
    # map takes square roots and squares them again): each is compared with the float64 result, not with the other.
    p32 = [torch.tensor(a, requires_grad=True) for a in f0]
    m32 = cls(p32[0], p32[1], p32[2], pb["Stf"], pb["opt"], Mask=mask)
    raw = hip_ops.backward(*[t.detach().contiguous() for t in m32.lame_padded()], pb["Stf"], 1, pb["Shot_ids"], pb["para_fname"])
    p64 = [torch.tensor(a.astype(np.float64), requires_grad=True) for a in f0]
    m64 = cls(p64[0], p64[1], p64[2], pb["Stf"], pb["opt"], Mask=mask.double())
    l64, u64, d64 = m64.lame_padded()
    ((l64 * raw[1].double()).sum() + (u64 * raw[2].double()).sum() + (d64 * raw[3].double()).sum()).backward()
    for k, name in enumerate(("PHI", "CC", "SW")):
        ref = list(m64.parameters())[k].grad.numpy()
        assert np.abs(ref).max() > 0
        dev = {d: float(np.abs(res[d][1 + k] - ref).max() / np.abs(ref).max()) for d in ("cuda", "cpu")}
        print("%s d/d%s: float32 chain vs float64 chain rule, max deviation / max |g|: %r" % (cls_name, name, dev))
        assert dev["cuda"] <= 5e-4 and dev["cpu"] <= 5e-4, (cls_name, name, dev)
        # (the yardstick uses the CPU run's media, so "cpu" shows the chain rule's own rounding, ~1e-7; torch's GPU kernels
        # produce media that differ in the last bit, and the propagator's gradient answers a 1-ulp change of the medium with
        # ~1e-5: measured 2.6e-5 (VRH) and 2e-4 (Gassmann) with torch's GPU expressions in round 3 -- float32 conditioning of the
        # problem, on either device; since round 5 the "cuda" chain is the fused map of csrc/param_maps.hip, kinds 5 and 6)


def test_elastic_propagator_equals_the_autograd_module(tmp_path, hip_ops):
    """The reference's second caller (propagator.py:57-226: `ElasticPropagator.apply_forward / apply_gradient`, velocities in
    km/s, chain rule by hand) against the autograd module `FWI` (m/s) on the same survey: same gathers on disk, same misfit,
    gradients equal after the factor 1000 of the velocity unit; and `device=` (everything in HBM) equals the CPU-tensor way."""
    from sepfwi import modules as M
    from sepfwi.propagator import ElasticPropagator, Model, Survey
    pb = P.make_problem(str(tmp_path / "a"), hetero=True, nSteps=150, nshots=2)
    nz, nx, nPml = pb["opt"]["nz"], pb["opt"]["nx"], pb["nPml"]
    sv = pb["survey"]
    survey = Survey(pb["para"]["f0"], np.array([sv["shot%d" % i]["x_src"] for i in range(2)]), np.array([sv["shot%d" % i]["z_src"] for i in range(2)]),
                    np.array(sv["shot0"]["x_rec"]), np.array(sv["shot0"]["z_rec"]))
    km = lambda m: {"vp": m["vp"] / np.float32(1e3), "vs": m["vs"] / np.float32(1e3), "rho": m["rho"]}
    exp = str(tmp_path / "b")
    import os
    os.makedirs(exp)
    mk = lambda m: Model(nx, nz, pb["para"]["dx"], pb["para"]["dz"], pb["nSteps"], pb["para"]["dt"], nPml, m["vp"], m["vs"], m["rho"], exp)
    ElasticPropagator(mk(km(pb["true"])), survey).apply_forward(ngpu=1)
    lt, mt, dt_ = pb["lame_true"]
    hip_ops.obscalc(lt, mt, dt_, pb["Stf"], 1, pb["Shot_ids"], pb["para_fname"])
    for sid in range(2):       # the two callers leave the same observed gathers (moduli differ by float rounding of the unit change)
        a = np.fromfile(os.path.join(exp, "Data", "Shot_ett%d.bin" % sid), dtype=np.float32)
        b = np.fromfile(os.path.join(pb["data_dir"], "Shot_ett%d.bin" % sid), dtype=np.float32)
        assert a.size == b.size and np.linalg.norm(a - b) <= 2e-5 * np.linalg.norm(b)
    f = [torch.tensor(pb["init"][k], requires_grad=True) for k in ("vp", "vs", "rho")]
    fwi = M.FWI(f[0], f[1], f[2], pb["Stf"], pb["opt"])
    loss = fwi(pb["Shot_ids"], ngpu=1)
    loss.backward()
    loss = loss.detach()
    want = [fwi.Vp.grad.numpy() * 1e3, fwi.Vs.grad.numpy() * 1e3, fwi.Den.grad.numpy()]
    outs = {}
    for dev in (None, "cuda"):
        outs[dev] = ElasticPropagator(mk(km(pb["true"])), survey, device=dev).apply_gradient(mk(km(pb["init"])), ngpu=1)
        misfit, gvp, gvs, grho, gstf = outs[dev]
        assert abs(float(misfit[0]) - float(loss)) <= 2e-4 * float(loss) and gstf.shape == (2, pb["nSteps"])
        for g, w in zip((gvp, gvs, grho), want):   # interior cells: autograd folds the padding's gradient onto the edge cells, this caller crops
            assert g.shape == (nz, nx) and np.abs(g - w)[1:-1, 1:-1].max() <= 5e-4 * np.abs(w[1:-1, 1:-1]).max()
    assert outs[None][0] == outs["cuda"][0]
    for a, b in zip(outs[None][1:], outs["cuda"][1:]):
        assert np.abs(a - b).max() <= 2e-6 * np.abs(b).max()


def test_only_one_field_inverted_on_hip_tensors(tmp_path, hip_ops):
    """Main-004-FWI-Rock-Physics.py:117-119 inverts one field of three (the others without requires_grad).  On HIP tensors the fused
    map then has two constant inputs: its gradient for the inverted field is the corresponding block of the all-fields gradient,
    through the real operator, and the SciPy glue sees a vector of that one field."""
    from sepfwi import modules as M
    from sepfwi.obj_wrapper import PyTorchObjective
    pb = P.make_problem(str(tmp_path), hetero=True, nSteps=120, nshots=2)
    lt, mt, dt_ = pb["lame_true"]
    hip_ops.obscalc(lt, mt, dt_, pb["Stf"], 1, pb["Shot_ids"], pb["para_fname"])
    T = lambda k, g: torch.tensor(pb["init"][k], device="cuda", requires_grad=g)
    full = M.FWI(T("vp", True), T("vs", True), T("rho", True), pb["Stf"], pb["opt"])
    assert full._fusable()
    f_full = full(pb["Shot_ids"], ngpu=1)
    f_full.backward()
    for which, name in (("vp", "Vp"), ("vs", "Vs"), ("rho", "Den")):
        part = M.FWI(T("vp", which == "vp"), T("vs", which == "vs"), T("rho", which == "rho"), pb["Stf"], pb["opt"])
        assert part._fusable() and [n for n, _ in part.named_parameters()] == [name]
        obj = PyTorchObjective(part, lambda: part(pb["Shot_ids"], ngpu=1))
        fun, jac = obj.fun, obj.jac
        assert obj.x0.size == pb["init"]["vp"].size and fun(obj.x0) == float(f_full.detach())
        g = jac(obj.x0).reshape(pb["init"]["vp"].shape)
        w = getattr(full, name).grad.cpu().numpy().astype(np.float64)
        assert np.abs(g - w).max() <= 1e-6 * np.abs(w).max(), name


def test_whole_iteration_on_a_side_torch_stream(tmp_path, hip_ops):
    """Module -> fused maps -> propagator -> chain rule with everything issued on a NON-default torch stream (the library then runs
    on that caller's stream and does not order itself behind the default one): same loss and gradients as on the default stream,
    bit for bit, also when the tensors were produced on the side stream just before."""
    from sepfwi import modules as M
    pb = P.make_problem(str(tmp_path), hetero=True, nSteps=150, nshots=3)
    lt, mt, dt_ = pb["lame_true"]
    hip_ops.obscalc(lt, mt, dt_, pb["Stf"], 1, pb["Shot_ids"], pb["para_fname"])
    res = []
    for side in (None, torch.cuda.Stream()):
        ctx = torch.cuda.stream(side) if side is not None else torch.cuda.stream(torch.cuda.default_stream())
        hip_ops.release()
        with ctx:
            f = [torch.tensor(pb["init"][k], device="cuda").mul_(1.0).requires_grad_(True) for k in ("vp", "vs", "rho")]
            fwi = M.FWI(f[0], f[1], f[2], pb["Stf"], pb["opt"])
            loss = fwi(pb["Shot_ids"], ngpu=1)
            loss.backward()
            if side is not None:
                side.synchronize()
        torch.cuda.synchronize()
        res.append([float(loss.detach())] + [p.grad.cpu() for p in fwi.parameters()])
    assert res[0][0] == res[1][0] and res[0][0] > 0
    for a, b in zip(res[0][1:], res[1][1:]):
        assert torch.equal(a, b)
