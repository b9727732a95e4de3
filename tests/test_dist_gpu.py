"""Two ranks (gloo) sharing the one GPU of the test box: the real HIP operator under torch.distributed.
Checks that the sharded run (each rank its contiguous block of shots, one all-reduce) returns on every rank what a
single process computes for all shots."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import problems as P

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, workdir, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "sep-2023_amd"), os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
    import problems as P2
    from sepfwi import fwi_ops
    fwi_ops.device_override = 0
    pb = P2.make_problem(os.path.join(workdir, "rank%d" % rank), hetero=True, nSteps=180, nshots=3)
    # both ranks read the SAME observed data directory (written by the parent)
    import json
    para = dict(pb["para"]); para["data_dir_name"] = os.path.join(workdir, "Data")
    json.dump(para, open(pb["para_fname"], "w"))
    lam, mu, den = [t.cuda() for t in pb["lame_init"]]
    m, gL, gM, gD, gS = fwi_ops.backward(lam, mu, den, pb["Stf"], world, pb["Shot_ids"], pb["para_fname"])
    q.put((rank, float(m), gL.cpu().numpy(), gM.cpu().numpy(), gD.cpu().numpy(), gS.numpy()))
    td.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_on_one_gpu_match_single_process(tmp_path, oracle, hip_ops):
    import json
    work = str(tmp_path)
    pb = P.make_problem(os.path.join(work, "single"), hetero=True, nSteps=180, nshots=3)
    para = dict(pb["para"]); para["data_dir_name"] = os.path.join(work, "Data")
    json.dump(para, open(pb["para_fname"], "w"))
    pb["data_dir"] = para["data_dir_name"]
    os.makedirs(para["data_dir_name"], exist_ok=True)
    lt, mt, dt_ = pb["lame_true"]
    hip_ops.obscalc(lt, mt, dt_, pb["Stf"], 1, pb["Shot_ids"], pb["para_fname"])
    lam, mu, den = pb["lame_init"]
    ref = hip_ops.backward(lam, mu, den, pb["Stf"], 1, pb["Shot_ids"], pb["para_fname"])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, work, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for r in res:
        assert abs(r[1] - float(ref[0])) <= 1e-5 * abs(float(ref[0]))
        for k in (2, 3, 4):
            assert P.rel_l2(r[k], ref[k - 1].numpy()) <= 1e-5
    # gStf: rank 0 holds its own block (shot 0), rows by local position; other ranks zeros
    assert P.rel_l2(res[0][5][:1], ref[4].numpy()[:1]) <= 1e-6
    assert np.all(res[1][5] == 0.0)
