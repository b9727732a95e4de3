#!/usr/bin/env python
"""bench.py -- headline benchmark of the hot path (BASELINE.json: "Gcell-updates/s (fwd+adj),
2000x1000 grid x 4000 steps; 1/2/4/8 GPUs").

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one FWI gradient evaluation (`fwi_ops.backward`: forward + boundary-saving adjoint of
--shots-per-step (default 3) shots per GPU on the 2000x1000 model, 4000 time steps, plus -- for N > 1 -- the single
RCCL all-reduce of [gLambda|gMu|gDen|misfit]).  Three shots per GPU per step because the session overlaps the forward
passes of up to three shots on three streams.  Weak scaling: every rank owns the same number of shots per step.  Inputs (model, source,
observed data) are resident in HBM when the timed region starts.  Prints ONE JSON line on rank 0.

cell-update = one grid cell advanced one time step by one propagator; a fwd+adj shot is
3 * N_c * (nSteps-1) updates with N_c = (nz+2nPml)*(nx+2nPml).
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "sep-2023_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
BYTES_PER_UPDATE_FWDADJ = 184.0 / 3.0   # 60 B fwd + 124 B bwd per cell per step = 61.33 B / cell-update
BYTES_FWD = 60.0
# Dominant kernel of the sweep: k_bwd_persist, the persistent backward time loop -- ONE launch per shot and backward pass, whose time
# step IS the backward step of SURVEY.md 8(d): 124 algorithmic bytes per cell (5 forward fields r/w 40, 5 adjoint fields r/w 40,
# 5 coefficients 20, 3 gradients r/w 24).  `roofline.achieved` = N_c * 124 B / (HIP-event time of the passes / their time steps).
# Where the loop is not in use (option bwd_fuse=2, a busy GPU) the two-launch step's k_bwd_b is sampled with HIP events instead and
# charged the arrays it read-modify-writes: szz,sxx,sxz 24 + adjoint vz,vx 16 + lambda,mu,ave_mu 12 + grad lambda,mu 16 = 68 B per cell.
BYTES_K_BWD_STRESS = 68.0
# PMC bytes / algorithmic bytes are quoted per STEP only (`roofline.traffic_ratio`: fwd_step against 60 B, bwd_step against 124 B per
# cell): splitting a step's bytes between its two kernels is an apportioning, not a measurement of either.
FWD_KERNELS, BWD_KERNELS = ("k_stress_fwd_save", "k_velocity_fwd"), ("k_bwd_a", "k_bwd_b")


def marmousi_style(nz, nx, seed=2023, pert_amp=0.1, sigma_init=40.0):
    """SURVEY.md 8(d) C2/C3 synthetic: 1-D gradient Vp 1500->4500 + Gaussian-filtered N(0,1) perturbation
    (sigma 8 cells, +-10 %), Vs = Vp/1.732, rho = 310 Vp^0.25; initial = Gaussian-smoothed (sigma 40).  The two keyword
    arguments are for the end-to-end example only (a gentler inverse problem); the bench uses the defaults."""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    base = np.linspace(1500.0, 4500.0, nz)[:, None] * np.ones((1, nx))
    pert = gaussian_filter(rng.standard_normal((nz, nx)), 8.0)
    pert = pert_amp * pert / np.abs(pert).max()
    vp = base * (1.0 + pert)
    vp0 = gaussian_filter(vp, sigma_init)
    mk = lambda v: (v.astype(np.float32), (v / 1.732).astype(np.float32), (310.0 * v ** 0.25).astype(np.float32))
    return mk(vp), mk(vp0)


def setup_problem(workdir, nz, nx, nSteps, n_shots_total, nPml=32, dh=10.0, dt=1.0e-3, f0=10.0, rec_stride=1, **model_kw):
    from sepfwi import utils as ft
    nPad = ft.nPad_for(nz, nPml)
    nz_pad, nx_pad = nz + 2 * nPml + nPad, nx + 2 * nPml
    para_fname = os.path.join(workdir, "para_file.json")
    survey_fname = os.path.join(workdir, "survey_file.json")
    ft.paraGen(nz_pad, nx_pad, dh, dh, nSteps, dt, f0, nPml, nPad, para_fname, survey_fname, os.path.join(workdir, "Data"))
    src_x = (10 + np.round(np.arange(n_shots_total) * (nx - 21) / max(n_shots_total - 1, 1))).astype(int)
    rec_x = np.arange(10, nx - 10, rec_stride).astype(int)      # rec_stride > 1 (scripts/ab_bench.py): channels that are NOT a line of consecutive cells
    ft.surveyGen(np.full(src_x.shape, 2), src_x, np.full(rec_x.shape, 2), rec_x, survey_fname)
    true, init = marmousi_style(nz, nx, **model_kw)

    def lame(m):
        vp, vs, rho = [torch.tensor(ft.padding_numpy_array(a, nPml, nPad)) for a in m]
        return ((vp ** 2 - 2.0 * vs ** 2) * rho / 1e6).contiguous(), (vs ** 2 * rho / 1e6).contiguous(), rho.contiguous()

    Stf = torch.tensor(ft.sourceGene(f0, nSteps, dt), dtype=torch.float32).repeat(n_shots_total, 1)
    return dict(para_fname=para_fname, lame_true=lame(true), lame_init=lame(init), Stf=Stf, nPad=nPad, nPml=nPml,
                nz_pad=nz_pad, nx_pad=nx_pad, n_c=(nz + 2 * nPml) * nx_pad, nrec=int(rec_x.size))


def effective_cores():
    """Host cores this process may really use: the smaller of the scheduler affinity and the cgroup CPU quota (a GPU box shows its
    host's 256 hardware threads in os.cpu_count() but may grant a share of them)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().split()[0])
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline(nz, nx, seconds=12.0):
    """The reference's CPU propagator is the Numba solver (DAS_Waveform_Modeling/src/elasticSolver.py);
    numba cannot travel, so its C restatement (oracle/numba_oracle.c, pinned bit-for-bit to the reference
    by tests/golden/numba_*.npz) is timed: one shot per host core, like Pool(min(nsrc, cpu_count))
    (elasticSolver.py:163-166), on the SAME 2000x1000 grid for a bounded number of steps -> `value`.  Extra keys
    (BASELINE.md section 3): the same solver on ONE core, and the float32 restatement of the TorchFWI fwd+adj path (the
    parity oracle, oracle/torchfwi_oracle.c) with one shot per core -- the like-for-like figure of the GPU metric."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    O.build()
    host_cores = os.cpu_count() or 1
    cores = max(1, min(effective_cores(), 32))
    ndamp = 32
    vp = np.full((nx, nz), 3000.0)
    vs = vp / 1.732
    rho = np.full((nx, nz), 2400.0)
    src = np.array([[nx // 2 * 10.0, nz // 2 * 10.0]])
    rec = np.array([[nx // 3 * 10.0, nz // 3 * 10.0]])

    def run(nt):
        t0 = time.perf_counter()
        O.numba_forward(nx, nz, ndamp, 10.0, 10.0, 1.0e-3, nt, 10.0, vp, vs, rho, src, rec, rec, np.zeros((1, 6)))
        return time.perf_counter() - t0

    t_probe = run(4)
    nt1 = int(max(4, min(400, 0.4 * seconds / max(t_probe / 4.0, 1e-6))))
    t_one = run(nt1)                                        # one shot on one core, nothing else running
    cells = (nx + 2 * ndamp) * (nz + 2 * ndamp)
    one_core = cells * nt1 / t_one / 1e9
    nt = int(max(4, min(400, seconds / max(t_one / nt1, 1e-6))))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        list(ex.map(lambda _: run(nt), range(cores)))
    el = time.perf_counter() - t0
    val = cores * cells * nt / el / 1e9
    out = {"value": round(val, 5), "unit": "Gcell-updates/s", "cores": cores, "kind": "port",
           "sample": "float64 C restatement of elasticSolver.py (velocity+stress = 1 cell-update), %dx%d grid + %d sponge, "
                     "%d steps, %d shots in parallel (one per core: the smaller of scheduler affinity, cgroup quota and 32; os.cpu_count() = %d), forward only, %.1f s"
                     % (nx, nz, ndamp, nt, cores, host_cores, el),
           "one_core": {"value": round(one_core, 5), "unit": "Gcell-updates/s", "cores": 1,
                        "sample": "the same solver, one shot on one core, %d steps, %.1f s" % (nt1, t_one)}}
    try:
        out["torchfwi_f32_fwdadj"] = cpu_baseline_fwdadj(nz, nx, cores, seconds)
    except Exception as e:      # a reported extra: never let it take the bench line down
        out["torchfwi_f32_fwdadj"] = {"error": repr(e)[:200]}
    return out


def cpu_baseline_fwdadj(nz, nx, cores, seconds):
    """The float32 CPU restatement of the TorchFWI propagator (the parity oracle) on the bench problem's own padded grid:
    forward + boundary-saving adjoint of `cores` shots, one per core (its OpenMP loop over shots), for a bounded number of
    time steps.  Same cell-update count as the GPU metric: 3 * N_c * (nSteps - 1) per shot."""
    from oracle import oracle as O
    from sepfwi import utils as ft
    import json
    nPml = 32
    nPad = ft.nPad_for(nz, nPml)
    nz_pad, nx_pad = nz + 2 * nPml + nPad, nx + 2 * nPml
    n_c = (nz + 2 * nPml) * nx_pad
    lam = np.full((nz_pad, nx_pad), 2400.0 * (3000.0 ** 2 - 2.0 * 1732.0 ** 2) / 1e6, np.float32)
    mu = np.full((nz_pad, nx_pad), 2400.0 * 1732.0 ** 2 / 1e6, np.float32)
    den = np.full((nz_pad, nx_pad), 2400.0, np.float32)

    def run(nt, nshots):
        d = tempfile.mkdtemp(prefix="sepfwi_cpu_")
        try:
            pf, sf = os.path.join(d, "p.json"), os.path.join(d, "s.json")
            ft.paraGen(nz_pad, nx_pad, 10.0, 10.0, nt, 1.0e-3, 10.0, nPml, nPad, pf, sf, os.path.join(d, "Data"))
            rx = np.arange(10, nx - 10).astype(int)
            sx = np.linspace(10, nx - 11, nshots).round().astype(int)
            ft.surveyGen(np.full(sx.shape, 2), sx, np.full(rx.shape, 2), rx, sf)
            para, survey = json.load(open(pf)), json.load(open(sf))
            stf = np.tile(ft.sourceGene(10.0, nt, 1.0e-3).astype(np.float32), (nshots, 1))
            obs = np.zeros((nshots, 4, rx.size, nt), np.float32)
            t0 = time.perf_counter()
            O.cufd(lam, mu, den, stf, 1, list(range(nshots)), para, survey, obs=obs)
            return time.perf_counter() - t0
        finally:
            shutil.rmtree(d, ignore_errors=True)

    try:    # one OpenMP thread per shot: libgomp would otherwise start one per hardware thread of the HOST (256) and let the idle ones spin
        import ctypes
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(cores))
    except OSError:
        pass
    # Differential timing: allocation, the model averages, file set-up and the thread start-up are the same for a short and a
    # long run, so the rate is (work of the extra time steps) / (extra time); the long run is repeated and both figures are
    # reported (the hosts are shared: the spread between two identical runs is part of the answer).
    nt_a = 9
    t_a = run(nt_a, cores)        # the oracle's OpenMP loop over shots: one shot per thread, `cores` shots -> `cores` threads busy
    per_step = max(t_a / (nt_a - 1), 1e-6)
    nt_b = nt_a + int(max(100, min(400, seconds / per_step)))          # >= 100 more time steps
    t_b = [run(nt_b, cores) for _ in range(2)]
    rates = [cores * 3.0 * n_c * (nt_b - nt_a) / max(t - t_a, 1e-9) / 1e9 for t in t_b]
    return {"value": round(float(np.mean(rates)), 5), "unit": "Gcell-updates/s (fwd+adj)", "cores": cores, "kind": "port",
            "runs": [round(r, 5) for r in rates],
            "sample": "float32 C restatement of the reference's cufd (forward + boundary-saving adjoint + imaging), padded %dx%d "
                      "grid, %d shots in parallel (one per core); differential: (%d - %d) time steps in (%.1f, %.1f) - %.1f s, "
                      "set-up and allocation cancel" % (nx_pad, nz_pad, cores, nt_b, nt_a, t_b[0], t_b[1], t_a)}


def kernel_source_digest():
    """sha256 over the files that define the field kernels: profiles/traffic.json carries the digest of the version its PMC
    counters were collected on (scripts/make_traffic_json.py)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("kernels.hip", "kernels_device.hpp", "kernels_bodies.hpp", "kernels_quiet.hpp", "kernels_step.hpp", "kernels_persist.hpp", "kernels_aux.hpp",
              "device_common.hpp", "fwi_types.hpp"):
        with open(os.path.join(ROOT, "sep-2023_amd", "csrc", f), "rb") as fp:
            h.update(fp.read())
    return h.hexdigest()


def launcher_cmd(n_gpus, argv, port=None):
    """`python bench.py --gpus N` without torchrun: the command line of the N-rank job this process turns itself into
    (one rank per GPU over RCCL, rendezvous on 127.0.0.1) -- the same line the driver uses for N > 1."""
    import socket
    if port is None:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def call32(workdir, args, dev, local, n_shots=32):
    """BASELINE.json configs[2] as one call: 32 shots, forward + boundary-saving adjoint gradient, one `fwi_ops.backward`.
    Own survey (32 sources across the model), observed data modelled into the session's HBM store beforehand (untimed)."""
    from sepfwi import fwi_ops
    os.makedirs(workdir, exist_ok=True)
    pb = setup_problem(workdir, args.nz, args.nx, args.nsteps, n_shots)
    lam_t, mu_t, den_t = [t.to(dev) for t in pb["lame_true"]]
    lam, mu, den = [t.to(dev) for t in pb["lame_init"]]
    ids = torch.arange(n_shots, dtype=torch.int32)
    fwi_ops._cufd(3, local, lam_t, mu_t, den_t, pb["Stf"], ids, pb["para_fname"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fwi_ops.backward(lam, mu, den, pb["Stf"], 1, ids, pb["para_fname"])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    st = fwi_ops.stats(pb["para_fname"], local)
    upd = n_shots * 3.0 * pb["n_c"] * (args.nsteps - 1)
    return {"value": round(upd / el / 1e9, 4), "unit": "Gcell-updates/s", "ms": round(el * 1e3, 1), "shots": n_shots,
            "device_ms": round(st["fwd_ms"] + st["bwd_ms"], 1),
            "what": "one fwi_ops.backward over %d shots (configs[2]), wall time of the call, after the timed region" % n_shots}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nz", type=int, default=1000)
    ap.add_argument("--nx", type=int, default=2000)
    ap.add_argument("--nsteps", type=int, default=4000, help="time steps per shot")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="fwdadj", choices=["fwdadj", "fwd"])
    ap.add_argument("--shots-per-step", type=int, default=3,
                    help="shots each GPU processes per step (3: the forward passes of the three overlap on three streams)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="collective backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--no-call32", action="store_true", help="skip the one 32-shot call of configs[2] reported as `call32` at N = 1")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE", help="library option for this run (sepfwi_set_option), e.g. bwd_fuse=2")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks on device 0 (rehearsal on a one-GPU box; use with --backend gloo)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher of N ranks (a child process; its exit code is ours) instead of
        # silently measuring one GPU.  This parent never initialises the HIP runtime.
        import subprocess
        # (the parent stays GPU-free: no device query here -- a rank without a device fails with its own message below)
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(launcher_cmd(args.gpus, sys.argv[1:]), env=env))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as td
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if args.share_gpu:      # rehearsal of the N-rank path on a one-GPU box: every rank drives device 0, gloo reduces on the host
        local = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if local >= torch.cuda.device_count():
            raise SystemExit("bench.py: rank %d needs HIP device %d but only %d are visible" % (rank, local, torch.cuda.device_count()))
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            td.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            td.init_process_group(backend=args.backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product path has no CPU fallback)")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from sepfwi import fwi_ops
    fwi_ops.device_override = local
    K, W = args.steps, args.warmup
    spr = max(args.shots_per_step, 1)
    per_rank_steps = max(K, 1)
    n_total = world * per_rank_steps * spr
    workdir = tempfile.mkdtemp(prefix="sepfwi_bench_r%d_" % rank)
    try:
        pb = setup_problem(workdir, args.nz, args.nx, args.nsteps, n_total)
        lam_t, mu_t, den_t = [t.to(dev) for t in pb["lame_true"]]
        lam, mu, den = [t.to(dev) for t in pb["lame_init"]]
        Stf = pb["Stf"]
        # shot ids: step s uses the block [s*world*spr, (s+1)*world*spr); rank r owns the r-th contiguous group of spr shots
        my_ids = [s * world * spr + rank * spr + j for s in range(per_rank_steps) for j in range(spr)]
        # observed data for my shots (untimed set-up): modelled from the "true" model straight into the session's HBM store
        # (SEPFWI_CALC_OBSERVE_TO_STORE: no Shot_*.bin files -- the file route would write and read back 127 MB per shot and rank)
        from sepfwi import dist as _dist
        _cufd = fwi_ops._cufd
        for k in range(0, len(my_ids), spr):
            _cufd(3, local, lam_t, mu_t, den_t, Stf, torch.tensor(my_ids[k:k + spr], dtype=torch.int32), pb["para_fname"])
        del lam_t, mu_t, den_t

        def step(s):
            ids = torch.arange(s * world * spr, (s + 1) * world * spr, dtype=torch.int32)
            if args.mode == "fwd":
                return fwi_ops.forward(lam, mu, den, Stf, local, ids[rank * spr:(rank + 1) * spr], pb["para_fname"])
            return fwi_ops.backward(lam, mu, den, Stf, world, ids, pb["para_fname"])

        from sepfwi import _native
        _native.check(_native.lib().sepfwi_set_option(b"probe", 61))   # HIP-event timestamps on every 61st k_bwd_b launch (two-launch step only; the loop is timed as a whole)
        for kv in args.option:
            k_, v_ = kv.split("=")
            _native.check(_native.lib().sepfwi_set_option(k_.encode(), int(v_)))
        for w in range(W):
            step(w % per_rank_steps)
        torch.cuda.synchronize()
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()
        _dist.enable_collective_timing(True)
        _dist.collective_stats(reset=True)
        t0 = time.perf_counter()
        fwd_ms = bwd_ms = call_ms = 0.0
        probe_us, probe_n = 0.0, 0
        persist_steps = bwd_steps = 0
        for s in range(K):
            step(s)
            st = fwi_ops.stats(pb["para_fname"], local)
            fwd_ms += st["fwd_ms"]
            bwd_ms += st["bwd_ms"]
            call_ms += st["total_ms"]       # this rank's own propagator call, without the collective and its wait for the slowest rank
            n_launch = st["launches"]       # kernel launches of the last call (every shot, both time loops, set-up and finalisation)
            probe_us += st["probe_kernel_us"] * st["probe_calls"]
            probe_n += st["probe_calls"]
            persist_steps += st.get("persist_steps", 0)
            bwd_steps += st["bwd_steps"]
        torch.cuda.synchronize()
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        coll = _dist.collective_stats(reset=True) if args.mode == "fwdadj" else None
        rank_ms = [call_ms / max(K, 1)]
        rank_ar = [coll["allreduce_ms"] if coll and coll["allreduce_ms"] is not None else 0.0]
        # which share of this rank's backward time steps ran inside the persistent loop, and -- where not all of them did -- why not: a
        # rank that fell back to per-step launches (its GPU shared with another job, say) is ~10 % slower and must not hide inside
        # rank_ms_per_step.max
        rank_loop = [persist_steps / float(bwd_steps) if bwd_steps else None]
        rank_why = [fwi_ops.loop_status(pb["para_fname"], local) if args.mode == "fwdadj" else ""]
        if world > 1:
            if td.get_world_size() != args.gpus:
                raise SystemExit("bench.py: the %s group has %d ranks, --gpus says %d" % (td.get_backend(), td.get_world_size(), args.gpus))
            cdev = dev if args.backend == "nccl" else "cpu"
            t = torch.tensor([el], dtype=torch.float64, device=cdev)
            td.all_reduce(t, op=td.ReduceOp.MAX)
            el = float(t.item())
            # per-rank figures (untimed): each rank's own propagator time per step (load imbalance) and its mean all-reduce time
            mine = torch.tensor([rank_ms[0], rank_ar[0]], dtype=torch.float64, device=cdev)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            td.all_gather(allr, mine)
            rank_ms = [float(a[0]) for a in allr]
            rank_ar = [float(a[1]) for a in allr]
            # every rank's loop share and reason, as fixed-size tensors through the same all_gather as above (no pickling on the way)
            enc = (rank_why[0].encode("utf-8", "replace")[:240]).ljust(240, b" ")
            mine2 = torch.tensor([-1.0 if rank_loop[0] is None else rank_loop[0]] + [float(c) for c in enc], dtype=torch.float64, device=cdev)
            all2 = [torch.zeros_like(mine2) for _ in range(world)]
            td.all_gather(all2, mine2)
            rank_loop = [None if float(a[0]) < 0 else float(a[0]) for a in all2]
            rank_why = [bytes(int(v) for v in a[1:].tolist()).decode("utf-8", "replace").strip() for a in all2]

        passes = 3 if args.mode == "fwdadj" else 1
        updates_per_shot = passes * pb["n_c"] * (args.nsteps - 1)
        value = world * K * spr * updates_per_shot / el / 1e9
        if rank == 0:
            # roofline of the dominant kernel group, measured live with HIP events on the session stream
            # (sepfwi_stats.fwd_ms / bwd_ms): algorithmic bytes per time step / measured time per time step.
            nst = K * spr * (args.nsteps - 1)
            traffic = traffic_ratio = traffic_p = None
            tf = os.path.join(ROOT, "profiles", "traffic.json")   # PMC-derived HBM bytes per launch
            if os.path.exists(tf) and args.nz == 1000 and args.nx == 2000:
                tj = json.load(open(tf))
                # the counters were collected on ONE version of the kernels: a changed kernel file makes them stale -> null
                if tj.get("kernel_source_sha256") == kernel_source_digest():
                    traffic = tj.get("k_bwd_b_bytes_per_launch")
                    traffic_p = tj.get("k_bwd_persist_bytes_per_time_step")     # persistent loop: PMC bytes of a launch / its time steps
                    # PMC bytes / algorithmic bytes per time step (1.0 = every array touched exactly once)
                    pmc = {k: tj[k + "_bytes_per_launch"] for k in FWD_KERNELS + BWD_KERNELS if k + "_bytes_per_launch" in tj}
                    traffic_ratio = {}
                    if all(k in pmc for k in BWD_KERNELS):
                        traffic_ratio["bwd_step"] = round(sum(pmc[k] for k in BWD_KERNELS) / (124.0 * pb["n_c"]), 3)
                    if all(k in pmc for k in FWD_KERNELS):
                        traffic_ratio["fwd_step"] = round(sum(pmc[k] for k in FWD_KERNELS) / (BYTES_FWD * pb["n_c"]), 3)
            if args.mode == "fwdadj" and probe_n > 0:
                per_step_us = probe_us / probe_n
                ach = pb["n_c"] * BYTES_K_BWD_STRESS / (per_step_us * 1e-6) / 1e9
                kern = "k_bwd_b (%d launches sampled with HIP events in the timed region)" % probe_n
            elif args.mode == "fwdadj":
                per_step_us = bwd_ms * 1e3 / nst
                ach = pb["n_c"] * 124.0 / (per_step_us * 1e-6) / 1e9
                kern = "whole backward time step"
                if persist_steps > 0:      # the backward pass ran as ONE persistent launch per shot: that kernel IS the backward step (124 B per cell)
                    kern = "k_bwd_persist (one launch per shot and backward pass; %d time steps by HIP events around the passes)" % persist_steps
                    traffic = traffic_p
                    if traffic_ratio is not None and traffic_p is not None:
                        traffic_ratio["bwd_step"] = round(traffic_p / (124.0 * pb["n_c"]), 3)
            else:
                per_step_us = fwd_ms * 1e3 / nst
                ach = pb["n_c"] * BYTES_FWD / (per_step_us * 1e-6) / 1e9
                kern = "whole forward time step (k_stress<FWD> + k_velocity<FWD>)"
                traffic = traffic_ratio = None
            out = {
                "metric": "Gcell-updates/s (fwd+adj), %dx%d grid x %d steps" % (args.nx, args.nz, args.nsteps) if args.mode == "fwdadj" else "Gcell-updates/s (fwd)",
                "value": round(value, 4), "unit": "Gcell-updates/s", "n_gpus": world, "steps": K, "warmup": W,
                "ms_per_step": round(el * 1e3 / max(K, 1), 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": ("configs[2]" if args.mode == "fwdadj" else "configs[1]") + " shape: %dx%d model (padded %dx%d), %d time steps, %d DAS channels, "
                                       "%d shot(s) per GPU per step, %s" % (args.nx, args.nz, pb["nx_pad"], pb["nz_pad"], args.nsteps,
                                                                        pb["nrec"], spr, "forward + boundary-saving adjoint gradient" if args.mode == "fwdadj" else "forward only"),
                           "cell_updates_per_shot": updates_per_shot, "shots_per_gpu_per_step": spr, "parallelism": "shots x%d" % world,
                           **({"options": list(args.option)} if args.option else {})},   # a run with non-default library options says so
                "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                             "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_ratio": traffic_ratio, "kernel": kern,
                             "avg_us": round(per_step_us, 2),
                             "whole_job_frac": round(value * (BYTES_PER_UPDATE_FWDADJ if args.mode == "fwdadj" else BYTES_FWD) / world / HBM_PEAK_GBPS, 4),
                             # un-apportioned cross-checks from the session's own HIP-event timing of the time loops: 60 B per cell
                             # and forward step, 124 B per cell and backward step
                             "fwd_step_frac": round(pb["n_c"] * BYTES_FWD / (fwd_ms * 1e3 / nst * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4) if fwd_ms > 0 else None,
                             "bwd_step_frac": round(pb["n_c"] * 124.0 / (bwd_ms * 1e3 / nst * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4) if bwd_ms > 0 else None},
                "fwd_ms_per_shot": round(fwd_ms / (K * spr), 2), "bwd_ms_per_shot": round(bwd_ms / (K * spr), 2),
                "fwd_us_per_time_step": round(fwd_ms * 1e3 / nst, 2), "bwd_us_per_time_step": round(bwd_ms * 1e3 / nst, 2),
                # SURVEY.md 8(d) "what else to report": device time of the two time loops (HIP events) beside the wall time of a step,
                # and launches per shot and time step (the reference: 12 forward + 12 backward; here 2 + 2 on a line of channels)
                "device_ms_per_step": round((fwd_ms + bwd_ms) / max(K, 1), 3),
                "launches_per_shot_time_step": round(n_launch / float(spr * (args.nsteps - 1)), 3),
            }
            # did the collective backend really see N ranks, and what did the one all-reduce per step cost?  `allreduce_ms` is the mean
            # HIP-event time around the collective on the rank that waited least (the last to arrive: the collective itself); `_max`
            # the rank that waited longest (collective + waiting for the slowest rank).  rank_ms_per_step: each rank's own propagator call.
            if world > 1 and coll is not None:
                out["rccl"] = {"ranks": td.get_world_size(), "backend": td.get_backend(), "calls_per_rank": coll["calls"],
                               "bytes": coll["bytes"], "allreduce_ms": round(min(rank_ar), 4), "allreduce_ms_max": round(max(rank_ar), 4),
                               "staged_copies": coll["staged"], "devices": "shared device 0 (rehearsal)" if args.share_gpu else "one per rank"}
            out["rank_ms_per_step"] = {"min": round(min(rank_ms), 3), "max": round(max(rank_ms), 3)}
            if args.mode == "fwdadj":
                fr = [f for f in rank_loop if f is not None]
                out["persistent_loop"] = {"share_of_bwd_steps_min": round(min(fr), 4) if fr else None,
                                          "share_of_bwd_steps_per_rank": [None if f is None else round(f, 4) for f in rank_loop],
                                          "why_not": {str(r): w for r, w in enumerate(rank_why) if w}}
            if world == 1 and args.mode == "fwdadj" and not args.no_call32 and args.nz == 1000 and args.nx == 2000:
                # configs[2] literally: ONE fwi_ops.backward call over 32 shots on this GPU (outside the timed region above)
                out["call32"] = call32(os.path.join(workdir, "call32"), args, dev, local)
            if not args.no_cpu_baseline and world == 1:
                out["cpu_baseline"] = cpu_baseline(args.nz, args.nx)
            print(json.dumps(out))
    finally:
        shutil.rmtree(workdir, ignore_errors=True)
        if world > 1:
            td.destroy_process_group()


if __name__ == "__main__":
    main()
