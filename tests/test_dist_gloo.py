"""N > 1 path on CPU: two processes, gloo backend.  The propagator itself needs a GPU, so each rank's
single-device call `_cufd` is replaced by a deterministic stand-in that depends on the shot ids it was
given; the test checks the shot partition (reference split rule) and the ONE fused all-reduce."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shot_grad(shape, sid):
    g = torch.arange(shape[0] * shape[1], dtype=torch.float32).reshape(shape)
    return g * (sid + 1) * 1e-3, -g * (sid + 2) * 1e-3, g * 0 + sid


def _worker(rank, world, port, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "sep-2023_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
    from sepfwi import dist
    dist.enable_collective_timing(True)
    from sepfwi.ops import fwi_ops
    seen = []
    ncoll = [0]
    real_allreduce = td.all_reduce

    def counting_allreduce(*a, **k):
        ncoll[0] += 1
        return real_allreduce(*a, **k)
    td.all_reduce = counting_allreduce

    def fake_cufd(calc_id, gpu_id, Lambda, Mu, Den, Stf, shot_ids, para_fname, out_device=None):
        ids = [int(i) for i in shot_ids]
        seen.extend(ids)
        gL, gM, gD = [torch.zeros(Lambda.shape) for _ in range(3)]
        for sid in ids:
            a, b, c = _shot_grad(Lambda.shape, sid)
            gL += a; gM += b; gD += c
        gS = torch.stack([torch.full((Stf.shape[1],), float(s)) for s in ids])
        return torch.tensor([float(sum(ids)) + 0.5 * len(ids)]), gL, gM, gD, gS
    fwi_ops._cufd = fake_cufd
    lam = torch.ones(6, 5)
    Stf = torch.zeros(7, 11)
    ids = torch.arange(7, dtype=torch.int32)
    m, gL, gM, gD, gS = fwi_ops.backward(lam, lam, lam, Stf, world, ids, "unused.json")
    first = (rank, seen[:], float(m), gL.numpy().copy(), gM.numpy().copy(), gD.numpy().copy(), gS.numpy().copy(), ncoll[0], dist.my_block(7))

    # the production layout: ops._cufd hands out views of ONE buffer [gL | gM | gD | misfit]; the collective must act on
    # that buffer itself (no staging copy), still exactly once
    bufs = []

    def fused_cufd(calc_id, gpu_id, Lambda, Mu, Den, Stf, shot_ids, para_fname, out_device=None):
        m0, a, b, c, gS0 = fake_cufd(calc_id, gpu_id, Lambda, Mu, Den, Stf, shot_ids, para_fname)
        n = Lambda.numel()
        fused = torch.cat([a.reshape(-1), b.reshape(-1), c.reshape(-1), m0.reshape(-1)])
        bufs.append(fused)
        return fused[3 * n:3 * n + 1], fused[0:n].view(Lambda.shape), fused[n:2 * n].view(Lambda.shape), fused[2 * n:3 * n].view(Lambda.shape), gS0
    fwi_ops._cufd = fused_cufd
    ncoll[0] = 0
    m2, gL2, gM2, gD2, gS2 = fwi_ops.backward(lam, lam, lam, Stf, world, ids, "unused.json")
    zero_copy = (dist.fused_view(m2, gL2, gM2, gD2) is not None and gL2.untyped_storage().data_ptr() == bufs[0].untyped_storage().data_ptr()
                 and ncoll[0] == 1 and torch.equal(gL2, torch.from_numpy(first[3])) and torch.equal(gD2, torch.from_numpy(first[5]))
                 and float(m2) == first[2])
    cs = dist.collective_stats(reset=True)      # two operator calls so far: one staged (separate tensors), one on the buffer itself
    q.put(first + (zero_copy, cs, dist.collective_stats()))
    td.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_ranks_partition_and_single_allreduce():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=100) for _ in range(world)], key=lambda t: t[0])
    [p.join(30) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    # contiguous blocks, reference rule int(linspace(0, 7, 3)) = [0, 3, 7]
    assert res[0][1] == [0, 1, 2] and res[1][1] == [3, 4, 5, 6]
    assert res[0][8] == (0, 3) and res[1][8] == (3, 7)
    eL = sum(_shot_grad((6, 5), s)[0] for s in range(7)).numpy()
    eM = sum(_shot_grad((6, 5), s)[1] for s in range(7)).numpy()
    eD = sum(_shot_grad((6, 5), s)[2] for s in range(7)).numpy()
    for r in res:
        assert r[7] == 1                                    # exactly one collective per operator call
        assert r[9]                                         # fused buffer reduced in place: same storage, same numbers, one collective
        cs, after = r[10], r[11]                            # the record bench.py prints as "rccl": {...}
        assert cs["ranks"] == 2 and cs["backend"] == "gloo" and cs["calls"] == 2 and cs["staged"] == 1
        assert cs["bytes"] == 4 * (3 * 6 * 5 + 1) and cs["allreduce_ms"] is not None and cs["allreduce_ms"] >= 0.0
        assert after["calls"] == 0 and after["allreduce_ms"] is None
        assert abs(r[2] - (21 + 3.5)) < 1e-5                # misfit summed over ranks
        np.testing.assert_allclose(r[3], eL, rtol=1e-6)
        np.testing.assert_allclose(r[4], eM, rtol=1e-6)
        np.testing.assert_allclose(r[5], eD, rtol=1e-6)
    # gStf: rank 0's block only, rows by local position (Torch_Fwi.cpp:102-103)
    assert res[0][6].shape == (7, 11) and np.all(res[0][6][:3, 0] == [0, 1, 2]) and np.all(res[0][6][3:] == 0)
