#!/usr/bin/env python
"""End-to-end DAS horizontal-fibre FWI on the headline shape (BASELINE.json configs[3] / configs[4]): 2000x1000
Marmousi-style model, 4000 time steps, N shots sharded over the GPUs of one node, SciPy L-BFGS-B with the reference's
options (DAS_Waveform_Inversion/Main-001-...py:126-168).  Prints wall-clock per gradient evaluation and per iteration.

    torchrun --nproc-per-node 8 examples/das_fwi_2000x1000.py --shots 256 --niter 1     # configs[3]
    torchrun --nproc-per-node 8 examples/das_fwi_2000x1000.py --shots 128 --niter 10    # configs[4]
    python examples/das_fwi_2000x1000.py --shots 6 --niter 2                            # one GPU, reduced survey

Synthetic data (no network): observed data are generated from the "true" model, the inversion starts from its smoothed
version (bench.marmousi_style).  Every tensor of the iteration lives in HBM; per evaluation the host sees the flat
float64 parameter / gradient vectors of SciPy only."""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch
from scipy import optimize

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sep-2023_amd")]
import bench                                   # noqa: E402  (model + survey generator of the benchmark)
from sepfwi import dist as fdist               # noqa: E402
from sepfwi import fwi_ops                     # noqa: E402
from sepfwi import modules as M                # noqa: E402
from sepfwi import utils as ft                 # noqa: E402
from sepfwi.obj_wrapper import PyTorchObjective  # noqa: E402


class TimedObjective(PyTorchObjective):
    """PyTorchObjective.cache with a stop-watch on each phase of one evaluation: float64 vector -> HBM parameters (unpack),
    the forward chain (parameterisation maps + propagator call + all-reduce; the propagator's own wall clock comes from
    sepfwi_stats), the autograd chain rule, HBM gradients -> float64 vector (pack).  What is left of an L-BFGS-B iteration is
    SciPy's own work on the flat vectors."""

    def __init__(self, obj, loss, para_fname, gpu_id):
        super().__init__(obj, loss)
        self.para_fname, self.gpu_id = para_fname, gpu_id
        self.phases = []

    def cache(self, x):
        tick = time.perf_counter
        t0 = tick()
        state = self.unpack_parameters(x)
        for name, buf in self.obj.named_buffers():
            state[name] = buf
        self.obj.load_state_dict(state)
        self.cached_x = x
        self.obj.zero_grad()
        torch.cuda.synchronize()
        t1 = tick()
        val = self.loss()
        self.f = val.item()
        torch.cuda.synchronize()
        t2 = tick()
        val.backward()
        torch.cuda.synchronize()
        t3 = tick()
        self.jac = self.pack_grads()
        t4 = tick()
        prop = fwi_ops.stats(self.para_fname, self.gpu_id)["total_ms"] * 1e-3
        self.phases.append(dict(unpack=t1 - t0, forward=t2 - t1, propagator=prop, chain_rule=t3 - t2, pack=t4 - t3, total=t4 - t0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shots", type=int, default=6)
    ap.add_argument("--niter", type=int, default=2)
    ap.add_argument("--nz", type=int, default=1000)
    ap.add_argument("--nx", type=int, default=2000)
    ap.add_argument("--nsteps", type=int, default=4000)
    ap.add_argument("--no-bounds", action="store_true",
                    help="run unconstrained like the reference's experiments (on this synthetic start the first steps then leave the\n"
                         "Courant limit)")
    ap.add_argument("--scipy-minimize", action="store_true",
                    help="go through optimize.minimize() as the reference's scripts do instead of sepfwi.obj_wrapper.minimize_lbfgsb\n"
                         "(same routine, same iterates; 10-20 s more per call on 6 M bounded unknowns)")
    ap.add_argument("--pert", type=float, default=0.1, help="amplitude of the true model's random perturbation (bench.marmousi_style: 0.1)")
    ap.add_argument("--sigma-init", type=float, default=40.0, help="Gaussian smoothing [cells] that makes the initial model from the true one (40)")
    ap.add_argument("--step0", type=float, default=20.0, help="largest model change [m/s, kg/m^3] of L-BFGS-B's first trial step (objective scaling)")
    ap.add_argument("--max-seconds", type=float, default=0.0, help="stop after the iteration that ends beyond this many seconds (0: no limit)")
    ap.add_argument("--no-restart", action="store_true", help="end at the first line-search failure like the reference's scripts instead of restarting L-BFGS-B from the current iterate")
    ap.add_argument("--mask-rows", type=int, default=4, help="rows below the surface kept fixed (the reference's scripts: 4; sources and fibre sit in row 2)")
    ap.add_argument("--files", action="store_true", help="observed data through Shot_*.bin files as the reference does (default: straight into the HBM store)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="collective backend under torchrun (nccl = RCCL)")
    ap.add_argument("--share-gpu", action="store_true", help="every rank on device 0: rehearsal of the N-rank run on a one-GPU box (with --backend gloo)")
    ap.add_argument("--quiet-skip", action="store_true", help="library option quiet_skip: leave out updates whose every input is exactly +0 (same bits, less time)")
    a = ap.parse_args()
    if a.quiet_skip:
        from sepfwi import _native
        _native.check(_native.lib().sepfwi_set_option(b"quiet_skip", 1))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if a.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as td
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            td.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            td.init_process_group(backend="gloo")
    rank = fdist.rank()
    dev = torch.device("cuda", local)
    fwi_ops.device_override = local

    # every rank writes the same parameter / survey files into its own directory (observed data of its own shots only)
    work = tempfile.mkdtemp(prefix="sepfwi_fwi_r%d_" % rank)
    nPml = 32
    nPad = ft.nPad_for(a.nz, nPml)
    pb = bench.setup_problem(work, a.nz, a.nx, a.nsteps, a.shots, pert_amp=a.pert, sigma_init=a.sigma_init)
    true, init = bench.marmousi_style(a.nz, a.nx, pert_amp=a.pert, sigma_init=a.sigma_init)
    Stf = pb["Stf"].to(dev)
    Shot_ids = torch.arange(a.shots, dtype=torch.int32)
    opt = dict(nz=a.nz, nx=a.nx, nz_orig=a.nz, nx_orig=a.nx, nPml=nPml, nPad=nPad, para_fname=pb["para_fname"])

    t0 = time.perf_counter()
    pad = lambda m: torch.tensor(ft.padding_numpy_array(m, nPml, nPad), dtype=torch.float32, device=dev)
    # observed data: modelled from the true model straight into the sessions' HBM store (no Shot_*.bin files: 127 MB per shot)
    M.FWI_obscalc(pad(true[0]), pad(true[1]), pad(true[2]), Stf, pb["para_fname"])(Shot_ids, ngpu=1, to_store=not a.files)
    torch.cuda.synchronize()
    t_obs = time.perf_counter() - t0

    Mask = torch.zeros((pb["nz_pad"], pb["nx_pad"]), dtype=torch.float32, device=dev)
    Mask[nPml + a.mask_rows:nPml + a.nz, nPml:nPml + a.nx] = 1.0     # keep the source / fibre rows fixed (Main-001-...py:62-66: 4 rows)
    T = lambda m: torch.tensor(m, dtype=torch.float32, device=dev, requires_grad=True)
    box = lambda m: (np.full(m.shape, float(m.min()) * 0.8), np.full(m.shape, float(m.max()) * 1.2))   # all three or none
    if a.no_bounds:
        fwi = M.FWI(T(init[0]), T(init[1]), T(init[2]), Stf, opt, Mask=Mask)
    else:
        fwi = M.FWI(T(init[0]), T(init[1]), T(init[2]), Stf, opt, Mask=Mask,
                    Vp_bounds=box(true[0]), Vs_bounds=box(true[1]), Den_bounds=box(true[2]))
    obj = TimedObjective(fwi, lambda: fwi(Shot_ids, ngpu=1), pb["para_fname"], local)
    fun, jac = obj.fun, obj.jac
    evals = []

    def timed_fun(x):
        t, n0 = time.perf_counter(), len(obj.phases)
        f = fun(x)
        torch.cuda.synchronize()
        if len(obj.phases) > n0:      # a real evaluation (cached re-evaluations of the same x cost nothing and are not counted)
            evals.append(time.perf_counter() - t)
        return f

    hist = []
    t0 = time.perf_counter()
    f0 = timed_fun(obj.x0)
    g0 = jac(obj.x0)
    hist.append(f0)
    # L-BFGS-B starts a bounded problem with the full step x - g: scale the objective so that this first step changes
    # the model by at most 20 (m/s, kg/m^3).  (The scaling has to sit here: FWIFunction.backward ignores grad_misfit,
    # as in the reference, FWI_ops.py:54-63.)
    c = a.step0 / float(np.abs(g0).max())
    if rank == 0:
        print("observed data: %.2f s   iterate 0: misfit %.6e  |g|_inf %.3e  (%.2f s per gradient evaluation, %d shots on "
              "%d GPU(s))" % (t_obs, f0, np.abs(g0).max(), evals[-1], a.shots, world), flush=True)

    def cb(x):
        hist.append(obj.f)
        if rank == 0:
            print("iterate %d: misfit %.6e   elapsed %.1f s   (%d gradient evaluations so far)" % (len(hist) - 1, obj.f, time.perf_counter() - t0, len(evals)), flush=True)
        if a.max_seconds > 0 and time.perf_counter() - t0 > a.max_seconds:
            raise StopIteration

    lb_opts = {"gtol": 1e-16, "maxiter": a.niter, "ftol": 1e-12, "maxcor": 5, "maxfun": 1500, "maxls": 6}
    # The reference's options (Main-001-...py:160-168).  With maxls = 6 the line search gives up ("ABNORMAL") when the gradient -- the
    # reference's adjoint, which is not the exact transpose in heterogeneous media -- no longer yields a point that passes
    # the Wolfe tests within six trials.  The reference's scripts then simply end; a production driver does what every L-BFGS user
    # does: it clears the memory and restarts from the current iterate (the first step is steepest descent again) until the requested
    # number of iterations is done.  --no-restart keeps the reference scripts' behaviour.
    x_cur, nit_done, restarts, stopped = obj.x0, 0, 0, False

    def cb_leg(x):
        nonlocal stopped
        try:
            cb(x)
        except StopIteration:
            stopped = True
            raise

    while True:
        lb_opts["maxiter"] = a.niter - nit_done
        if a.scipy_minimize:
            res = optimize.minimize(lambda x: c * timed_fun(x), x_cur, method="L-BFGS-B", jac=lambda x: c * jac(x), bounds=obj.bounds,
                                    tol=None, callback=cb_leg, options=lb_opts)
        else:       # the same compiled L-BFGS-B routine and the same iterates, without SciPy's per-element loops over the bounds
            from sepfwi.obj_wrapper import minimize_lbfgsb
            res = minimize_lbfgsb(lambda x: c * timed_fun(x), x_cur, lambda x: c * jac(x), bounds=obj.bounds, callback=cb_leg, **lb_opts)
        nit_done += res.nit
        x_cur = res.x
        if stopped or a.no_restart or nit_done >= a.niter or res.nit == 0 or "ABNORMAL" not in str(res.message):
            break
        restarts += 1
        if rank == 0:
            print("line search gave up after iterate %d (%s): L-BFGS memory cleared, restarting from there" % (nit_done, str(res.message).strip()), flush=True)
    res.nit = nit_done
    wall = time.perf_counter() - t0
    if rank == 0:
        n_c = pb["n_c"]
        upd = 3.0 * n_c * (a.nsteps - 1) * a.shots
        print("optimizer: %s (%d restart(s) after a failed line search); evaluation times [s]: %s" % (str(res.message).strip(), restarts, " ".join("%.2f" % e for e in evals)))
        print("done: %d iterations, %d gradient evaluations in %.1f s (%.1f s of it inside SciPy's L-BFGS-B on the %d float64 "
              "unknowns); misfit %.4e -> %.4e; mean %.2f s per evaluation = %.1f Gcell-updates/s including the autograd chain, "
              "the all-reduce and the host <-> device copies of the flat vectors" %
              (res.nit, len(evals), wall, wall - float(np.sum(evals)), obj.x0.size, hist[0], hist[-1], float(np.mean(evals)),
               upd / float(np.mean(evals)) / 1e9))
        ph = obj.phases
        mean = lambda k: float(np.mean([p[k] for p in ph]))
        ev = mean("total")
        print("split of one evaluation (mean of %d) [s]: unpack x -> HBM %.3f | forward chain %.3f of which the propagator call "
              "%.3f (maps, all-reduce, misfit read-back: %.3f) | autograd chain rule %.3f | pack gradients -> float64 %.3f | "
              "total %.3f" % (len(ph), mean("unpack"), mean("forward"), mean("propagator"), mean("forward") - mean("propagator"),
                             mean("chain_rule"), mean("pack"), ev))
        scipy_s = wall - float(np.sum([p["total"] for p in ph]))
        print("split of the run [s]: %d evaluations %.1f (propagator %.1f = %.1f %%) | SciPy L-BFGS-B on %d float64 unknowns %.1f "
              "(%.2f per iteration) | wall %.1f  => host share outside the propagator %.1f %%" %
              (len(ph), float(np.sum([p["total"] for p in ph])), float(np.sum([p["propagator"] for p in ph])),
               100.0 * float(np.sum([p["propagator"] for p in ph])) / wall, obj.x0.size, scipy_s, scipy_s / max(res.nit, 1), wall,
               100.0 * (1.0 - float(np.sum([p["propagator"] for p in ph])) / wall)))
    if world > 1:
        import torch.distributed as td
        td.destroy_process_group()
    import shutil
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
