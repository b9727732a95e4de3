"""Host-side helpers that produce the INPUTS of the propagator (unchanged semantics of the reference's
fwi_utils.py): JSON writers, Ricker source, replicate padding.  Pure numpy / torch."""
from __future__ import annotations

import json
import os

import numpy as np
import torch
import torch.nn.functional as F


def nPad_for(nz: int, nPml: int, multiple: int = 32) -> int:
    """Rows appended below the bottom PML so that nz + 2 nPml + nPad is a multiple of 32
    (Main-001-FWI-Anomaly-Vp-Vs-Den.py:35; yields 32, not 0, when already a multiple)."""
    return int(multiple - np.mod(nz + 2 * nPml, multiple))


def padding_numpy_array(arr: np.ndarray, npml: int, npad: int) -> np.ndarray:
    """Edge-replicate a (nz, nx) array to (nz + 2 npml + npad, nx + 2 npml)   (fwi_utils.py:11-27)."""
    return np.pad(arr, ((npml, npml + npad), (npml, npml)), mode="edge")


def padding(cp, cs, den, nz_orig, nx_orig, nz, nx, nPml, nPad):
    """Bilinear resize to (nz, nx) (identity when sizes agree) then replicate padding; differentiable
    (fwi_utils.py:31-44)."""
    out = []
    for t in (cp, cs, den):
        t4 = t.view(1, 1, nz_orig, nx_orig)
        t4 = F.interpolate(t4, size=(nz, nx), mode="bilinear", align_corners=False)
        t4 = F.pad(t4, pad=(nPml, nPml, nPml, nPml + nPad), mode="replicate")
        out.append(t4.view(nz + 2 * nPml + nPad, nx + 2 * nPml))
    return tuple(out)


def paraGen(nz, nx, dz, dx, nSteps, dt, f0, nPml, nPad, para_fname, survey_fname, data_dir_name,
            if_win=False, filter_para=None, if_src_update=False, scratch_dir_name="", if_cross_misfit=False,
            das_fiber="horizontal", obs_pack_fname=None, conditioning=None, obs_cache_mb=None):
    """Write the one-line parameter JSON (schema of fwi_utils.py:46-83; nz, nx are the PADDED sizes).
    das_fiber (extension): "horizontal" = axial strain exx = vx(x) - vx(x-1), the reference's live
    choice; "vertical" = ezz = vz(z) - vz(z-1) (recording_ezz / res_injection_ezz, Src/utilities.cu:620-641, which the
    reference only reaches by editing libCUFD.cu).  The key is written only when it is not the default, so default
    files stay byte-identical to the reference's."""
    para = {"nz": int(nz), "nx": int(nx), "dz": dz, "dx": dx, "nSteps": int(nSteps), "dt": float(dt),
            "f0": f0, "nPoints_pml": int(nPml), "nPad": int(nPad)}
    if if_win:
        para["if_win"] = True
    if filter_para is not None:
        para["filter"] = filter_para
    if if_src_update:
        para["if_src_update"] = True
    para["survey_fname"] = survey_fname
    para["data_dir_name"] = data_dir_name
    os.makedirs(data_dir_name, exist_ok=True)
    if if_cross_misfit:
        para["if_cross_misfit"] = True
    if das_fiber != "horizontal":
        if das_fiber != "vertical":
            raise ValueError("das_fiber must be 'horizontal' or 'vertical'")
        para["das_fiber"] = das_fiber
    if scratch_dir_name != "":
        para["scratch_dir_name"] = scratch_dir_name
        os.makedirs(scratch_dir_name, exist_ok=True)
    if conditioning is not None:   # "live" (default behaviour of this library) or "reference": if_win / filter / if_src_update / if_cross_misfit
        if conditioning not in ("live", "reference"):   # are parsed and ignored, as the reference's driver does (libCUFD.cu:353-457)
            raise ValueError("conditioning must be 'live' or 'reference'")
        para["conditioning"] = conditioning
    if obs_cache_mb:        # extension: HBM budget [MB] of the session's observed-data store (beyond it: pinned host memory)
        para["obs_cache_mb"] = int(obs_cache_mb)
    if obs_pack_fname:      # extension: one packed file of axial-strain gathers instead of four files per shot
        para["obs_pack_fname"] = obs_pack_fname
    with open(para_fname, "w") as fp:
        json.dump(para, fp)


def surveyGen(z_src, x_src, z_rec, x_rec, survey_fname, Windows=None, Weights=None, Src_Weights=None, Src_rxz=None, Rec_rxz=None,
              Das_sensitivity=None):
    """Write the one-line survey JSON: every shot shares the receiver list; indices are UNPADDED grid
    indices (fwi_utils.py:87-124).
    Das_sensitivity (extension): (nrec, 6) directional sensitivities of the DAS channels in the layout of
    the reference's Numba solver (DAS_Waveform_Modeling/src/elasticSolver.py:152-153,276): column 0 weighs exx, column 3
    ezz, column 1 exz -- ett = s0 exx + s3 ezz + s1 exz, for shaped / dipping fibres.  Omitted: a straight fibre along x
    (or z, parameter key das_fiber), the reference's CUDA behaviour; the key is only written when given."""
    z_src = np.asarray(z_src).tolist()
    x_src = np.asarray(x_src).tolist()
    z_rec = np.asarray(z_rec).tolist()
    x_rec = np.asarray(x_rec).tolist()
    survey = {"nShots": len(x_src)}
    for i in range(len(x_src)):
        shot = {"z_src": int(z_src[i]), "x_src": int(x_src[i]), "nrec": len(x_rec),
                "z_rec": [int(v) for v in z_rec], "x_rec": [int(v) for v in x_rec]}
        if Windows is not None:
            shot["win_start"] = [float(v) for v in Windows["shot%d" % i]["start"]]
            shot["win_end"] = [float(v) for v in Windows["shot%d" % i]["end"]]
        if Weights is not None:
            shot["weights"] = [float(v) for v in Weights["shot%d" % i]["weights"]]
        if Src_Weights is not None:
            shot["src_weight"] = Src_Weights[i]
        if Src_rxz is not None:
            shot["src_rxz"] = Src_rxz[i]
        if Rec_rxz is not None:
            shot["rec_rxz"] = np.asarray(Rec_rxz).tolist()
        if Das_sensitivity is not None:
            ds = np.asarray(Das_sensitivity, dtype=float)
            if ds.shape != (len(x_rec), 6):
                raise ValueError("Das_sensitivity must be (nrec, 6): exx, exz, -, ezz, -, - (elasticSolver.py:152-153)")
            shot["das_sensitivity"] = ds.tolist()
        survey["shot%d" % i] = shot
    with open(survey_fname, "w") as fp:
        json.dump(survey, fp)


def sourceGene(f, nStep, delta_t):
    """Ricker wavelet, delay 1.2/f, amplitude 1e7, float64 (fwi_utils.py:127-140)."""
    e = np.pi * np.pi * f * f
    tau = delta_t * np.arange(nStep) - 1.2 / f
    return (1.0 - 2.0 * e * tau ** 2) * np.exp(-e * tau ** 2) * 1.0e7


def read_shot_gather(data_dir, comp, shot_id, nSteps):
    """Shot_{pr|vx|vz|ett}{id}.bin -> (nrec, nSteps) float32 (libCUFD.cu:755-769)."""
    return np.fromfile(os.path.join(data_dir, "Shot_%s%d.bin" % (comp, shot_id)), dtype=np.float32).reshape(-1, nSteps)


PACK_MAGIC = b"SEPFWIP1"


def pack_observed(data_dir, shot_ids, nSteps, pack_fname):
    """One file for a whole survey's observed axial-strain gathers instead of the reference's four files per
    shot, of which only Shot_ett{id}.bin enters misfit and adjoint source (Src/libCUFD.cu:216-223,427,607).  Layout, little
    endian:  8 bytes magic "SEPFWIP1" | int32 nEntries | int32 nSteps | nEntries x (int32 shot_id, int32 nrec, int64 byte offset)
    | the gathers, float32 [nrec][nSteps] each, exactly the bytes of the Shot_ett files.  Named by the parameter key
    "obs_pack_fname" (paraGen(..., obs_pack_fname=)); shots missing from the pack are still read from their own files."""
    ids = [int(i) for i in shot_ids]
    gathers = [read_shot_gather(data_dir, "ett", i, nSteps) for i in ids]
    head = 8 + 8 + 16 * len(ids)
    off = head
    with open(pack_fname, "wb") as fp:
        fp.write(PACK_MAGIC)
        fp.write(np.array([len(ids), nSteps], dtype="<i4").tobytes())
        for i, g in zip(ids, gathers):
            fp.write(np.array([i, g.shape[0]], dtype="<i4").tobytes())
            fp.write(np.array([off], dtype="<i8").tobytes())
            off += g.size * 4
        for g in gathers:
            fp.write(np.ascontiguousarray(g, dtype="<f4").tobytes())
    return pack_fname


def read_packed_gather(pack_fname, shot_id):
    """-> (nrec, nSteps) float32 of one shot of a pack_observed file."""
    with open(pack_fname, "rb") as fp:
        if fp.read(8) != PACK_MAGIC:
            raise ValueError("%s is not a packed observed-data file" % pack_fname)
        n, nSteps = np.frombuffer(fp.read(8), dtype="<i4")
        for _ in range(int(n)):
            sid, nrec = np.frombuffer(fp.read(8), dtype="<i4")
            off = int(np.frombuffer(fp.read(8), dtype="<i8")[0])
            if int(sid) == int(shot_id):
                fp.seek(off)
                return np.frombuffer(fp.read(int(nrec) * int(nSteps) * 4), dtype="<f4").reshape(int(nrec), int(nSteps)).copy()
    raise KeyError("shot %d is not in %s" % (shot_id, pack_fname))


# Mineral / fluid constants of the reference's rock-physics maps (fwi_utils.py:156-167,311-322; FWI_ops.py:452-462,573-584):
# quartz, clay, water, hydrocarbon; cs = consolidation parameter of the drained frame (Dupuy et al. 2016).
ROCK = dict(k_q=37.00 * 1e9, k_c=21.00 * 1e9, k_w=2.25 * 1e9, k_h=0.04 * 1e9, mu_q=44.00 * 1e9, mu_c=10.00 * 1e9,
            rho_q=2.65 * 1e3, rho_c=2.55 * 1e3, rho_w=1.00 * 1e3, rho_h=0.10 * 1e3, cs=20.0)


def pcs2dv_vrh(phi, cc, sw):
    """(porosity, clay content, water saturation) -> (vp, vs, rho), Voigt-Reuss-Hill   (fwi_utils.py:154-196)."""
    R = ROCK
    kv = (1 - phi) * (R["k_c"] * cc + R["k_q"] * (1 - cc)) + phi * (R["k_w"] * sw + R["k_h"] * (1 - sw))
    kr_1 = (1 - phi) * (cc / R["k_c"] + (1 - cc) / R["k_q"]) + phi * (sw / R["k_w"] + (1 - sw) / R["k_h"])
    k = 0.5 * (kv + 1 / kr_1)
    mu = 0.5 * ((1 - phi) * (R["mu_c"] * cc + R["mu_q"] * (1 - cc)) + 0)
    rho = (R["rho_w"] * sw + R["rho_h"] * (1 - sw)) * phi + (R["rho_c"] * cc + R["rho_q"] * (1 - cc)) * (1 - phi)
    lam = k - 2. / 3. * mu
    return np.sqrt((lam + 2. * mu) / rho), np.sqrt(mu / rho), rho


def pcs2dv_gassmann(phi, cc, sw):
    """(porosity, clay content, water saturation) -> (vp, vs, rho), Biot-Gassmann with Voigt mineral mixing
    (fwi_utils.py:309-352: weighted_average, vrh(method='Voigt'), drained_moduli, biot_gassmann)."""
    R = ROCK
    rho_f = R["rho_w"] * sw + R["rho_h"] * (1 - sw)
    k_f = R["k_w"] * sw + R["k_h"] * (1 - sw)
    k_s = R["k_c"] * cc + R["k_q"] * (1 - cc)
    mu_s = R["mu_c"] * cc + R["mu_q"] * (1 - cc)
    rho_s = R["rho_c"] * cc + R["rho_q"] * (1 - cc)
    k_d = k_s * ((1 - phi) / (1 + R["cs"] * phi))
    mu_d = mu_s * ((1 - phi) / (1 + 1.5 * R["cs"] * phi))
    delta = ((1 - phi) / phi) * (k_f / k_s) * (1 - (k_d / (k_s - k_s * phi)))
    k_u = (phi * k_d + (1 - (1 + phi) * (k_d / k_s)) * k_f) / (phi * (1 + delta))
    rho = rho_f * phi + rho_s * (1 - phi)
    return np.sqrt((k_u + 0.75 * mu_d) / rho), np.sqrt(mu_d / rho), rho
