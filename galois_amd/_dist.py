"""
Multi-GPU use of the engine: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

New design -- the reference has no distributed path at all:
  * element-wise ufuncs, batched NTTs and Reed-Solomon codewords are independent units: `shard_range` hands each
    rank a contiguous slice, and there is NO data-path collective;
  * one sequence too long for a single GPU uses the four-step decomposition with exactly one all-to-all
    (`ntt_four_step_distributed`), with documented distributed input / output layouts.
"""
from __future__ import annotations

from typing import Callable

import numpy as np
import torch

from . import _lib as L


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [start, stop) slice of `total` independent units owned by `rank` (sizes differ by at most one)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} out of range for world size {world}")
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


# ---------------------------------------------------------------------------------------------------------------------
# ufunc.reduce over an array sharded across the ranks ("local reduce, then combine G partials")
# ---------------------------------------------------------------------------------------------------------------------
_DUAL_OP = {L.OP_ADD: L.OP_ADD, L.OP_MUL: L.OP_MUL, L.OP_SUB: L.OP_ADD, L.OP_DIV: L.OP_MUL}


def _device_reduce(field, flat: torch.Tensor, op: int) -> torch.Tensor:
    """One rank's fold of a 1-D shard with gfa_reduce (left fold for SUB / DIV); returns a 1-element tensor."""
    from ._array import _GFA_DTYPE, _ptr, _stream

    out = torch.empty(1, dtype=flat.dtype, device=flat.device)
    err = torch.zeros(1, dtype=torch.int32, device=flat.device) if op == L.OP_DIV else None
    L.check(L.lib().gfa_reduce(field._handle, op, _ptr(flat), _ptr(out), 1, flat.numel(), _GFA_DTYPE[flat.element_size()], _stream(),
                               _ptr(err) if err is not None else None), "gfa_reduce")
    if err is not None and int(err.item()) & L.DEVERR_ZERO_DIVISION:
        raise ZeroDivisionError("Cannot compute the multiplicative inverse of 0 in a Galois field.")
    return out


def _all_gather(t: torch.Tensor, group=None) -> list[torch.Tensor]:
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if t.is_cuda and dist.get_backend(group) == "gloo":  # plumbing rigs only (see _all_to_all)
        parts = [torch.empty(t.shape, dtype=t.dtype) for _ in range(world)]
        dist.all_gather(parts, t.cpu(), group=group)
        return [p.to(t.device) for p in parts]
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t.contiguous(), group=group)
    return parts


def reduce_sharded(field, local: torch.Tensor, op: int = L.OP_ADD, group=None, local_reduce: Callable | None = None) -> torch.Tensor:
    """ufunc.reduce (reference dispatch: _domains/_ufunc.py:686-689, kernels :180-198) of ONE array whose elements are spread
    over the ranks in rank order -- rank g holds the g-th contiguous slice, e.g. `shard_range`'s -- as a tensor in the field's
    device storage dtype.  Every rank folds its own slice (gfa_reduce), the G one-element partials are all-gathered (G * 8 bytes:
    the only traffic between GPUs) and every rank folds those, so all ranks return the same 1-element tensor.

    op: L.OP_ADD / OP_MUL, or the reference's left folds OP_SUB / OP_DIV (a0 - a1 - a2 - ...): rank 0 folds with the op itself,
    the other ranks with its dual (+ for -, * for /), and the final fold over [p0, p1, ...] uses the op again.  Rank 0's slice
    must not be empty for SUB / DIV (it holds a0).  `local_reduce(field, flat, op)` defaults to the HIP kernel; the CPU (gloo)
    tests inject an oracle-backed stand-in."""
    import torch.distributed as dist

    if op not in _DUAL_OP:
        raise ValueError("reduce is defined for add, subtract, multiply and divide only")
    rank = dist.get_rank(group)
    local_reduce = local_reduce or _device_reduce
    flat = local.reshape(-1).contiguous()
    mine_op = op if rank == 0 else _DUAL_OP[op]
    if flat.numel() == 0:
        if rank == 0 and op in (L.OP_SUB, L.OP_DIV):
            raise ValueError("rank 0 must hold at least one element for subtract.reduce / divide.reduce")
        part = torch.full((1,), 0 if mine_op == L.OP_ADD else 1, dtype=flat.dtype, device=flat.device)  # identity of the fold
    else:
        part = local_reduce(field, flat, mine_op)
    parts = _all_gather(part.reshape(1), group)
    return local_reduce(field, torch.cat(parts), op)


def reduce_distributed(x, ufunc=np.add, group=None):
    """`ufunc.reduce(x)` for a FieldArray whose flattened elements are this rank's slice of a larger array (device front end
    of `reduce_sharded`): returns a 0-d array of x's field, identical on every rank."""
    ops = {np.add: L.OP_ADD, np.subtract: L.OP_SUB, np.multiply: L.OP_MUL, np.true_divide: L.OP_DIV}
    if ufunc not in ops:
        raise ValueError("reduce is defined for np.add, np.subtract, np.multiply and np.true_divide only")
    cls = type(x)
    if getattr(cls, "_limbed", False):
        raise NotImplementedError("reduce_distributed: fields of order >= 2^64 are not supported yet")
    out = reduce_sharded(cls, x._t, ops[ufunc], group)
    return cls._wrap(out.reshape(()), x._np_dtype)


# ---------------------------------------------------------------------------------------------------------------------
# distributed four-step NTT
# ---------------------------------------------------------------------------------------------------------------------
# View the length-N input as an (n1 x n2) row-major matrix x[j1*n2 + j2] (N = n1*n2, powers of two).
#   input layout  ("column blocks"): rank g holds columns j2 in [g*n2/G, (g+1)*n2/G) as a local (n1 x n2/G) array.
#   output layout ("row blocks of the transposed result"): rank g holds X[k1 + n1*k2] for k1 in [g*n1/G, (g+1)*n1/G),
#                 all k2, as a local (n1/G x n2) array indexed [k1_local][k2].
# Steps: (1) local length-n1 transforms of the owned columns and multiplication by w^(j2*k1)  [gfa_ntt_columns]
#        (2) ONE all-to-all that turns column blocks into row blocks                          [RCCL over xGMI]
#        (3) local length-n2 transforms of the owned rows                                     [gfa_ntt, batched]
# `columns_to_local` / `local_to_natural` define the layouts for tests and for users who hold the data on one host.

def columns_to_local(x_full: np.ndarray, rank: int, world: int, n1: int, n2: int) -> np.ndarray:
    cols = n2 // world
    return np.ascontiguousarray(x_full.reshape(n1, n2)[:, rank * cols:(rank + 1) * cols])


def local_to_natural(parts: list[np.ndarray], n1: int, n2: int) -> np.ndarray:
    """Reassembles the natural-order spectrum from every rank's (n1/G x n2) output block."""
    world = len(parts)
    rows = n1 // world
    out = np.empty(n1 * n2, dtype=parts[0].dtype)
    view = out.reshape(n2, n1)  # X[k1 + n1*k2] -> view[k2, k1]
    for g, blk in enumerate(parts):
        view[:, g * rows:(g + 1) * rows] = blk.reshape(rows, n2).T
    return out


def _device_column_pass(field, local: torch.Tensor, n1: int, cols: int, col0: int, n_total: int, omega: int, in_pitch: int = 0) -> torch.Tensor:
    """`local`: an (n1, cols) array, or -- with in_pitch -- a view of `cols` adjacent columns of a wider row-major array whose
    rows are in_pitch elements apart (read in place).  The result is a compact (n1, cols) array."""
    from ._array import _GFA_DTYPE, _ptr, _stream

    out = torch.empty((n1, cols), dtype=local.dtype, device=local.device)
    L.check(L.lib().gfa_ntt_columns_pitched(field._handle, _ptr(local), in_pitch or cols, _ptr(out), cols, n1, cols, col0, n_total, omega,
                                            _GFA_DTYPE[local.element_size()], _stream()), "gfa_ntt_columns")
    return out


def _device_row_pass(field, rows: torch.Tensor, n2: int, omega_n2: int) -> torch.Tensor:
    from ._array import _GFA_DTYPE, _ptr, _stream

    out = torch.empty_like(rows)
    L.check(L.lib().gfa_ntt(field._handle, _ptr(rows), _ptr(out), n2, rows.numel() // n2, omega_n2, 0,
                            _GFA_DTYPE[rows.element_size()], _stream()), "gfa_ntt")
    return out


def _device_row_pass_from_chunks(field, recv: torch.Tensor, n2: int, omega_n2: int) -> torch.Tensor | None:
    """Row pass reading the all-to-all receive buffer recv[s][k1_local][c] in place (row k1_local, element s*cols + c): the
    kernel addresses the per-peer chunks itself, so the (rows, n2) re-layout copy disappears.  None if unsupported."""
    from ._array import _GFA_DTYPE, _ptr, _stream

    world, rows, cols = recv.shape
    out = torch.empty((rows, n2), dtype=recv.dtype, device=recv.device)
    rc = L.lib().gfa_ntt_chunked(field._handle, _ptr(recv), _ptr(out), n2, rows, omega_n2, 0, cols, rows * cols, cols, 0, 0, 0,
                                 _GFA_DTYPE[recv.element_size()], _stream())
    if rc == L.ERR_UNSUPPORTED:
        return None
    L.check(rc, "gfa_ntt_chunked")
    return out


def _device_row_pass_to_chunks(field, rows_in: torch.Tensor, n2: int, omega_n2: int, world: int) -> torch.Tensor | None:
    """Row pass of the inverse writing straight into the all-to-all SEND buffer send[s][k1_local][c]."""
    from ._array import _GFA_DTYPE, _ptr, _stream

    rows = rows_in.shape[0]
    cols = n2 // world
    send = torch.empty((world, rows, cols), dtype=rows_in.dtype, device=rows_in.device)
    rc = L.lib().gfa_ntt_chunked(field._handle, _ptr(rows_in), _ptr(send), n2, rows, omega_n2, 0, 0, 0, 0, cols, rows * cols, cols,
                                 _GFA_DTYPE[rows_in.element_size()], _stream())
    if rc == L.ERR_UNSUPPORTED:
        return None
    L.check(rc, "gfa_ntt_chunked")
    return send


def _device_column_pass_inv(field, local: torch.Tensor, n1: int, cols: int, col0: int, n_total: int, omega_inv: int,
                            scaled: bool) -> torch.Tensor:
    from ._array import _GFA_DTYPE, _ptr, _stream

    out = torch.empty_like(local)
    L.check(L.lib().gfa_ntt_columns_inv(field._handle, _ptr(local), _ptr(out), n1, cols, col0, n_total, omega_inv, 1 if scaled else 0,
                                        _GFA_DTYPE[local.element_size()], _stream()), "gfa_ntt_columns_inv")
    return out


def choose_split(n_total: int, world: int) -> tuple[int, int]:
    """(n1, n2) with n1 * n2 == n_total for ntt_four_step_distributed.  The column pass runs strided transforms of length
    n1 and is fastest when n1 fits the register-blocked kernel (n1 <= 2^10); n2 = n_total / n1 may then be up to 2^20 (the
    row pass is an ordinary batched transform).  Measured for 2^26 Goldilocks points over 8 ranks
    (tools/c5_local_bench.py): 1024 x 65536 -> 0.20 ms of kernels per rank, 8192 x 8192 -> 0.38 ms."""
    if n_total & (n_total - 1) or world & (world - 1):
        raise ValueError("n_total and the number of ranks must be powers of two")
    n1 = min(1 << 10, n_total // max(world, 2))
    while n_total // n1 > (1 << 20):
        n1 *= 2
    n2 = n_total // n1
    if n1 % world or n2 % world or n1 > (1 << 13):
        raise ValueError(f"no supported split of {n_total} points over {world} ranks")
    return n1, n2


def _all_to_all(recv: torch.Tensor, send: torch.Tensor, group=None) -> None:
    """The one collective of the path.  RCCL ("nccl" backend) moves device buffers directly over xGMI.  Under the gloo
    backend (CPU test rigs, or several ranks sharing ONE GPU for plumbing runs -- RCCL refuses duplicate devices) device
    tensors are staged through host memory; that route is for testing the N > 1 control flow, never for measurements."""
    import torch.distributed as dist

    if send.is_cuda and dist.get_backend(group) == "gloo":
        r = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(r.view(-1), send.reshape(-1).cpu(), group=group)
        recv.copy_(r)
    else:
        dist.all_to_all_single(recv.view(-1), send.reshape(-1), group=group)


class _Stamps:
    """Optional per-stage GPU timing (torch events on the current stream) for bench.py: stages in issue order."""

    def __init__(self, sink: dict | None):
        self.sink = sink
        self.events = []

    def mark(self, name: str):
        if self.sink is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.events.append((name, e))

    def finish(self):
        if self.sink is not None and len(self.events) > 1:
            self.events[-1][1].synchronize()
            for (_, e0), (name, e1) in zip(self.events[:-1], self.events[1:]):
                self.sink.setdefault(name, []).append(e0.elapsed_time(e1))


_SIDE_STREAMS: dict = {}


def _side_stream(device) -> "torch.cuda.Stream":
    key = str(device)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


def _all_to_all_lists(outs: list[torch.Tensor], ins: list[torch.Tensor], group=None) -> None:
    """List form of the exchange (one contiguous block per peer on either side).  gloo with device tensors: through the host."""
    import torch.distributed as dist

    if ins[0].is_cuda and dist.get_backend(group) == "gloo":
        r = [torch.empty(o.shape, dtype=o.dtype) for o in outs]
        _gloo_all_to_all(r, [i.contiguous().cpu() for i in ins], group)
        for o, h in zip(outs, r):
            o.copy_(h)
    elif dist.get_backend(group) == "gloo":
        r = [torch.empty(o.shape, dtype=o.dtype) for o in outs]
        _gloo_all_to_all(r, [i.contiguous() for i in ins], group)
        for o, h in zip(outs, r):
            o.copy_(h)
    else:
        dist.all_to_all(outs, [i.contiguous() for i in ins], group=group)


def _gloo_all_to_all(outs: list[torch.Tensor], ins: list[torch.Tensor], group=None) -> None:
    """gloo has no list all_to_all: one flat all_to_all_single of equal blocks instead (test rigs only)."""
    import torch.distributed as dist

    flat_in = torch.cat([i.reshape(-1) for i in ins])
    flat_out = torch.empty_like(flat_in)
    dist.all_to_all_single(flat_out, flat_in, group=group)
    n = ins[0].numel()
    for k, o in enumerate(outs):
        o.copy_(flat_out[k * n:(k + 1) * n].reshape(o.shape))


def ntt_four_step_distributed(field, local_cols: torch.Tensor, n1: int, n2: int, omega: int | None = None, group=None,
                              column_pass: Callable | None = None, row_pass: Callable | None = None,
                              timings: dict | None = None, nsub: int | None = None) -> torch.Tensor:
    """
    One rank's part of a single length-(n1*n2) NTT over `field` spread over the ranks of `group`.

    local_cols: (n1, n2/G) tensor in the field's device storage dtype (column-block input layout above).
    Returns the (n1/G, n2) row-block output layout.  `column_pass` / `row_pass` default to the HIP kernels; the
    CPU (gloo) tests inject oracle-backed stand-ins to exercise the exchange and the layout bookkeeping.
    """
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_total = n1 * n2
    cols = n2 // world
    rows = n1 // world
    if n2 % world or n1 % world:
        raise ValueError("n1 and n2 must be divisible by the number of ranks")
    if tuple(local_cols.shape) != (n1, cols):
        raise ValueError(f"local_cols must have shape {(n1, cols)}, not {tuple(local_cols.shape)}")
    if omega is None:
        omega = field._root_of_unity_int(n_total)
    column_pass = column_pass or _device_column_pass
    row_pass = row_pass or _device_row_pass
    st = _Stamps(timings)
    st.mark("start")
    # (1) columns: A[k1][c] = w^((col0+c)*k1) * sum_j1 x[j1][c] * w_n1^(j1*k1), in `nsub` sub-blocks of adjacent columns;
    # (2) the one exchange, issued per sub-block: rank r receives, from every rank p, rows [r*rows, (r+1)*rows) of p's sub-block s
    #     at recv[p][s] -- on a side stream under RCCL, so that sub-block s travels while sub-block s+1 is being transformed (the
    #     kernels and the exchange cost about the same in C5).  Still ONE logical all-to-all, in nsub grouped send / recv halves.
    nsub = int(nsub) if nsub else (2 if cols % 2 == 0 and cols >= 64 else 1)
    if cols % nsub:
        raise ValueError("the number of local columns must be divisible by nsub")
    csub = cols // nsub
    local_cols = local_cols.contiguous()
    recv = torch.empty((world, nsub, rows, csub), dtype=local_cols.dtype, device=local_cols.device)
    overlap = local_cols.is_cuda and dist.get_backend(group) == "nccl" and nsub > 1
    side = _side_stream(local_cols.device) if overlap else None
    for sb in range(nsub):
        if column_pass is _device_column_pass:
            a = _device_column_pass(field, local_cols[:, sb * csub:(sb + 1) * csub], n1, csub, rank * cols + sb * csub, n_total, omega,
                                    in_pitch=cols)
        else:
            a = column_pass(field, local_cols[:, sb * csub:(sb + 1) * csub].contiguous(), n1, csub, rank * cols + sb * csub, n_total, omega)
        outs = [recv[p, sb] for p in range(world)]
        ins = [a[p * rows:(p + 1) * rows] for p in range(world)]
        if overlap:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                dist.all_to_all(outs, ins, group=group)
            a.record_stream(side)
        else:
            _all_to_all_lists(outs, ins, group)
    if overlap:
        torch.cuda.current_stream().wait_stream(side)
    st.mark("columns_and_exchange_ms")
    # (3) rows: X[k1 + n1*k2] = sum_j2 A[k1][j2] * w_n2^(j2*k2),  w_n2 = w^n1.  recv[s][k1_local][c] is column s*cols + c of
    # row k1_local: the device row pass reads those per-peer chunks in place; stand-ins (and chunk sizes the kernel does not
    # take) get the (rows, n2) row-major copy
    omega_n2 = field._scalar(L.OP_POW, omega, n1)
    out = _device_row_pass_from_chunks(field, recv.view(world * nsub, rows, csub), n2, omega_n2) if row_pass is _device_row_pass and recv.is_cuda else None
    if out is None:
        mine = recv.permute(2, 0, 1, 3).reshape(rows, n2).contiguous()
        st.mark("relayout_ms")
        out = row_pass(field, mine, n2, omega_n2)
    st.mark("row_pass_ms")
    st.finish()
    return out


def intt_four_step_distributed(field, local_rows: torch.Tensor, n1: int, n2: int, omega: int | None = None, group=None,
                               scaled: bool = True, row_pass: Callable | None = None,
                               column_pass_inv: Callable | None = None, timings: dict | None = None) -> torch.Tensor:
    """
    Inverse of `ntt_four_step_distributed` (ifft_jit semantics, reference _domains/_function.py:387-392): consumes the
    (n1/G, n2) row-block layout that the forward transform produces (rank g holds X[k1 + n1*k2] at [k1 - g*n1/G][k2]) and
    returns the (n1, n2/G) column-block layout it consumes -- so intt(ntt(x)) costs two exchanges in total and no
    re-layout in between.  `omega` is the FORWARD root of unity (default: the field's); `scaled` divides by n1*n2.

    Steps (the forward ones backwards, with w' = w^-1):
      (1) local length-n2 transforms of the owned rows, root w'^n1                              [gfa_ntt, batched]
      (2) ONE all-to-all that turns row blocks into column blocks                                [RCCL over xGMI]
      (3) multiply (k1, c) by w'^(k1 * (col0 + c)), length-n1 transforms of the owned columns,
          scale by 1/(n1*n2)                                                                     [gfa_ntt_columns_inv]
    """
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_total = n1 * n2
    cols = n2 // world
    rows = n1 // world
    if n2 % world or n1 % world:
        raise ValueError("n1 and n2 must be divisible by the number of ranks")
    if tuple(local_rows.shape) != (rows, n2):
        raise ValueError(f"local_rows must have shape {(rows, n2)}, not {tuple(local_rows.shape)}")
    if omega is None:
        omega = field._root_of_unity_int(n_total)
    omega_inv = field._scalar(L.OP_RECIP, omega, 0)
    row_pass = row_pass or _device_row_pass
    column_pass_inv = column_pass_inv or _device_column_pass_inv
    st = _Stamps(timings)
    st.mark("start")
    # (1) rows: B[k1][o] = sum_k2 Y[k1][k2] * (w'^n1)^(k2*o), written straight into the send buffer send[s][k1_local][c]
    # (rank s receives columns [s*cols, (s+1)*cols) of every rank's row block) when the device kernel takes that layout
    w_rows = field._scalar(L.OP_POW, omega_inv, n1)
    send = _device_row_pass_to_chunks(field, local_rows.contiguous(), n2, w_rows, world) if row_pass is _device_row_pass and local_rows.is_cuda else None
    if send is None:
        b = row_pass(field, local_rows.contiguous(), n2, w_rows)
        st.mark("row_pass_ms")
        send = b.view(rows, world, cols).permute(1, 0, 2).contiguous()
        st.mark("relayout_ms")
    else:
        st.mark("row_pass_ms")
    # (2) the one exchange
    recv = torch.empty((world, rows, cols), dtype=send.dtype, device=send.device)
    _all_to_all(recv, send, group)
    st.mark("all_to_all_ms")
    # recv[s][k1_local][c] is row s*rows + k1_local of this rank's column block: already (n1, cols) row-major
    out = column_pass_inv(field, recv.view(n1, cols), n1, cols, rank * cols, n_total, omega_inv, scaled)
    st.mark("column_pass_ms")
    st.finish()
    return out
