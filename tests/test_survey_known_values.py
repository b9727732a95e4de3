"""Known values the survey obtained from the reference's OWN kernels (probe results): turned into tests.

  * cpmlInit (Src/utilities.cu:243-359) for N = 40, nPml = 8, dh = 10, f0 = 10, dt = 1e-3 returns b[0] = 0.54789609.
  * one forward el_stress call (Src/el_stress.cu:50-87) on a unit vz impulse gives sum|szz| = (2 * 9/8 + 2/24) / dz * (lambda + 2 mu) * dt
    = 4.66667e6 for (lambda + 2 mu) dt / dz = 2e6: the four taps of the staggered 4th-order difference.
Checked on the CPU oracle, on the product's host code (C ABI, no GPU needed) and -- for the stencil sum -- on the HIP kernels.
"""
import ctypes as C

import numpy as np
import pytest
import torch

import problems as P

B0 = 0.54789609


def test_cpml_b0_of_the_reference_kernels(oracle):
    """The oracle's restatement AND sepfwi_cpml_profiles (csrc/config.cpp, what the session uploads) give the value the
    reference's cpmlInit printed."""
    ref = oracle.cpml_init(40, 8, 10.0, 10.0, 1e-3)
    assert abs(float(ref["b"][0]) - B0) <= 5e-8, float(ref["b"][0])
    from sepfwi import _native
    L = _native.lib()
    arrs = [np.zeros(40, np.float32) for _ in range(6)]
    _native.check(L.sepfwi_cpml_profiles(*[a.ctypes.data for a in arrs], 40, 8, C.c_float(10.0), C.c_float(10.0), C.c_float(1e-3)))
    assert abs(float(arrs[2][0]) - B0) <= 5e-8, float(arrs[2][0])      # order: K, a, b, K_half, a_half, b_half
    assert np.array_equal(arrs[2], ref["b"])


def test_el_stress_on_a_vz_impulse(oracle):
    """ofwi_el_stress (oracle/torchfwi_oracle.c, restating Src/el_stress.cu:50-87) on vz = delta: sum|szz| = 4.66667e6."""
    nz, nx, dz, dt = 24, 20, 10.0, 1e-3
    lam, mu = 1.0e10, 0.5e10                                  # (lambda + 2 mu) dt / dz = 2e6
    n = nz * nx
    f = lambda v=0.0: np.full(n, v, np.float32)
    vz, vx, szz, sxx, sxz = f(), f(), f(), f(), f()
    vz[(nx // 2) * nz + nz // 2] = 1.0                        # internal layout a[x * nz + z] (Src/libCUFD.cu:71-77)
    mem = [f() for _ in range(4)]
    cp = [np.ones(max(nz, nx), np.float32) for _ in range(12)]

    class Cpml(C.Structure):
        _fields_ = [(k, C.POINTER(C.c_float)) for k in ("K_z", "a_z", "b_z", "K_z_half", "a_z_half", "b_z_half",
                                                          "K_x", "a_x", "b_x", "K_x_half", "a_x_half", "b_x_half")]
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    c = Cpml(*[p(a) for a in cp])
    oracle.lib().ofwi_el_stress(p(vz), p(vx), p(szz), p(sxx), p(sxz), *[p(m) for m in mem], p(f(lam)), p(f(mu)), p(f(mu)),
                                C.byref(c), C.c_int(nz), C.c_int(nx), C.c_float(dt), C.c_float(dz), C.c_float(dz), C.c_int(0), C.c_int(0),
                                C.c_int(1), None, None, None, None, None)
    expect = (2 * 9 / 8 + 2 / 24) / dz * (lam + 2 * mu) * dt
    assert abs(expect - 4.66667e6) < 10
    assert abs(float(np.abs(szz).sum()) - expect) <= 2e-6 * expect, float(np.abs(szz).sum())
    assert np.count_nonzero(szz) == 4 and np.count_nonzero(sxz) == 4      # dvz_dz feeds szz / sxx, dvz_dx feeds sxz
    assert abs(float(np.abs(sxx).sum()) - expect * lam / (lam + 2 * mu)) <= 2e-6 * expect


@pytest.mark.gpu
def test_hip_kernels_on_an_impulse(tmp_path, hip_ops):
    """The same four-tap sum on the HIP kernels, read back through sepfwi_debug_field.  A run from rest cannot hold a vz impulse
    (the source acts on the stresses), so the impulse is the first source sample: after the time step that injects it
    szz(src) = sxx(src) = 1500^2 stf dt exactly (add_source, Src/utilities.cu:531-538) and the forward velocity kernel has spread it
    with the same taps: sum|vz| = sum|vx| = (2 * 9/8 + 2/24) / dh * amp * dt / rho  (Src/el_velocity.cu:49-50,64-65,78-80)."""
    nSteps = 3
    stf = np.zeros(nSteps)
    stf[1] = 1.0e7                                            # sample 0 is zeroed by the taper; samples 1, 2 pass unchanged
    pb = P.make_problem(str(tmp_path), hetero=False, nSteps=nSteps, nshots=1, stf=stf, src_z=20, src_x=[30])
    lam, mu, den = pb["lame_init"]
    with P.kernel_options(batch=0):
        hip_ops.obscalc(lam, mu, den, pb["Stf"], 1, pb["Shot_ids"], pb["para_fname"])
        fld = [hip_ops.debug_field(pb["para_fname"], w).numpy() for w in range(5)]
    vz, vx, szz, sxx, sxz = fld
    dt, dh, rho = 1.0e-3, 10.0, 2400.0
    amp = np.float32(np.float32(1500.0 ** 2) * np.float32(1.0e7)) * np.float32(dt)
    zs, xs = 20 + pb["nPml"], 30 + pb["nPml"]
    assert szz[zs, xs] == amp and sxx[zs, xs] == amp and np.count_nonzero(szz) == 1 and np.count_nonzero(sxx) == 1
    assert not sxz.any()
    expect = (2 * 9 / 8 + 2 / 24) / dh * float(amp) * dt / rho
    for v in (vz, vx):
        assert np.count_nonzero(v) == 4
        assert abs(float(np.abs(v.astype(np.float64)).sum()) - expect) <= 2e-6 * expect
    assert np.count_nonzero(vz[:, xs]) == 4 and np.count_nonzero(vx[zs, :]) == 4     # vz along the column, vx along the row
