"""The N>1 path on CPU: world_size-2 gloo processes exercise the distributed four-step NTT's single all-to-all and its
layout bookkeeping (galois_amd/_dist.py) with oracle-backed stand-ins for the two per-rank HIP kernels, plus the
batch-sharding helper used by bench.py.  The HIP kernels themselves are covered by the -m gpu tests."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, order, n1, n2, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import galois_amd as ga
        from galois_amd import _lib as L
        from galois_amd import dist as gdist
        from oracle import gf_oracle as O

        GF = ga.GF(order)
        F = O.OracleField(order, 1, None, GF._primitive_element_int)
        n = n1 * n2
        x = np.random.default_rng(7).integers(0, min(order, 2**62), n, dtype=np.uint64) % np.uint64(order)
        omega = GF._root_of_unity_int(n)

        # oracle-backed stand-ins with the SAME contracts as gfa_ntt_columns / gfa_ntt (include/galois_amd.h)
        def column_pass(field, local, n1_, cols, col0, n_total, om):
            a = local.numpy().view(np.uint64).reshape(n1_, cols)
            w_n1 = int(F.pow([om], [n_total // n1_])[0])
            out = np.empty_like(a)
            for c in range(cols):
                col = F.ntt(a[:, c].copy(), omega=w_n1)
                tw = F.pow(np.full(n1_, om, dtype=np.uint64), ((col0 + c) * np.arange(n1_)) % n_total)
                out[:, c] = F.mul(col, tw)
            return torch.from_numpy(out.view(np.int64))

        def row_pass(field, rows, n2_, om2):
            a = rows.numpy().view(np.uint64)
            out = np.stack([F.ntt(r.copy(), omega=om2) for r in a])
            return torch.from_numpy(out.view(np.int64))

        def column_pass_inv(field, local, n1_, cols, col0, n_total, om_inv, scaled):
            a = local.numpy().view(np.uint64).reshape(n1_, cols)
            w_n1 = int(F.pow([om_inv], [n_total // n1_])[0])
            ninv = pow(n_total % order, order - 2, order)
            out = np.empty_like(a)
            for c in range(cols):
                tw = F.pow(np.full(n1_, om_inv, dtype=np.uint64), ((col0 + c) * np.arange(n1_)) % n_total)
                col = F.ntt(F.mul(a[:, c].copy(), tw), omega=w_n1)
                out[:, c] = F.mul(col, np.full(n1_, ninv, dtype=np.uint64)) if scaled else col
            return torch.from_numpy(out.view(np.int64))

        local = torch.from_numpy(gdist.columns_to_local(x, rank, world, n1, n2).view(np.int64))
        # the column pass in two sub-blocks, each with its own grouped exchange (the overlapped form of the device path)
        mine = gdist.ntt_four_step_distributed(GF, local, n1, n2, omega=omega, column_pass=column_pass, row_pass=row_pass,
                                               nsub=2 if (n2 // world) % 2 == 0 else 1)
        one = gdist.ntt_four_step_distributed(GF, local, n1, n2, omega=omega, column_pass=column_pass, row_pass=row_pass, nsub=1)
        assert torch.equal(mine, one)
        # the inverse consumes the row-block layout directly and returns the column-block layout: two exchanges in total
        back = gdist.intt_four_step_distributed(GF, mine, n1, n2, omega=omega, row_pass=row_pass, column_pass_inv=column_pass_inv)
        round_trip_ok = bool(torch.equal(back, local))
        flags = [None] * world
        dist.all_gather_object(flags, round_trip_ok)
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        if rank == 0:
            full = gdist.local_to_natural([g.numpy().view(np.uint64) for g in gathered], n1, n2)
            want = F.ntt(x, omega=omega)
            q.put(bool(np.array_equal(full, want)) and all(flags))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("order,n1,n2", [(65537, 16, 32), (2**64 - 2**32 + 1, 8, 16), (7340033, 32, 8)])
def test_distributed_four_step_ntt_two_ranks(order, n1, n2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, order, n1, n2, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _reduce_worker(rank, world, port, order, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import galois_amd as ga
        from galois_amd import _lib as L
        from galois_amd import dist as gdist
        from oracle import gf_oracle as O

        GF = ga.GF(order)
        F = O.OracleField(GF.characteristic, GF.degree, int(GF.irreducible_poly) if GF.degree > 1 else None, GF._primitive_element_int)
        fold = {L.OP_ADD: F.add, L.OP_SUB: F.sub, L.OP_MUL: F.mul, L.OP_DIV: F.div}

        def local_reduce(field, flat, op):  # same contract as gfa_reduce on one row: a left fold
            a = flat.numpy().view(np.uint64)
            acc = np.array([a[0]], dtype=np.uint64)
            for v in a[1:]:
                acc = fold[op](acc, np.array([v], dtype=np.uint64))
            return torch.from_numpy(acc.view(np.int64))

        n = 1001  # uneven shards: 501 + 500
        x = np.random.default_rng(11).integers(1, order, n, dtype=np.uint64)
        lo, hi = gdist.shard_range(n, rank, world)
        ok = True
        for op in (L.OP_ADD, L.OP_MUL, L.OP_SUB, L.OP_DIV):
            got = gdist.reduce_sharded(GF, torch.from_numpy(x[lo:hi].copy().view(np.int64)), op, local_reduce=local_reduce)
            want = np.array([x[0]], dtype=np.uint64)
            for v in x[1:]:
                want = fold[op](want, np.array([v], dtype=np.uint64))
            ok = ok and int(got.numpy().view(np.uint64)[0]) == int(want[0])
        # an empty shard on the last rank contributes the identity
        got = gdist.reduce_sharded(GF, torch.from_numpy((x[:7] if rank == 0 else x[:0]).copy().view(np.int64)), L.OP_MUL, local_reduce=local_reduce)
        want = np.array([x[0]], dtype=np.uint64)
        for v in x[1:7]:
            want = F.mul(want, np.array([v], dtype=np.uint64))
        ok = ok and int(got.numpy().view(np.uint64)[0]) == int(want[0])
        flags = [None] * world
        dist.all_gather_object(flags, ok)
        if rank == 0:
            q.put(all(flags))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("order", [2**8, 7340033, 3**5])
def test_reduce_of_an_array_sharded_over_two_ranks(order):
    """add / multiply / subtract / divide .reduce with one slice per rank: local folds, an all-gather of the two partials,
    one final fold -- against the oracle's fold of the whole array (left folds for subtract and divide)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reduce_worker, args=(r, 2, port, order, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_layout_helpers_roundtrip():
    sys.path.insert(0, ROOT)
    from galois_amd import dist as gdist

    n1, n2, world = 8, 16, 4
    x = np.arange(n1 * n2, dtype=np.uint64)
    parts = [gdist.columns_to_local(x, r, world, n1, n2) for r in range(world)]
    assert all(p.shape == (n1, n2 // world) for p in parts)
    assert np.array_equal(np.hstack(parts).reshape(-1), x)
    # output layout: rank g holds X[k1 + n1*k2] for its k1 block
    X = np.arange(n1 * n2, dtype=np.uint64)
    blocks = [np.array([[X[k1 + n1 * k2] for k2 in range(n2)] for k1 in range(g * n1 // world, (g + 1) * n1 // world)])
              for g in range(world)]
    assert np.array_equal(gdist.local_to_natural(blocks, n1, n2), X)


def test_choose_split():
    from galois_amd import _dist as D

    assert D.choose_split(1 << 26, 8) == (1 << 10, 1 << 16)
    assert D.choose_split(1 << 20, 2) == (1 << 10, 1 << 10)
    assert D.choose_split(1 << 8, 2) == (1 << 7, 2)
    assert D.choose_split(1 << 32, 8) == (1 << 12, 1 << 20)
    for n, w in ((1 << 26, 8), (1 << 16, 4), (1 << 12, 2)):
        n1, n2 = D.choose_split(n, w)
        assert n1 * n2 == n and n1 % w == 0 and n2 % w == 0
    with pytest.raises(ValueError):
        D.choose_split(1000, 8)
