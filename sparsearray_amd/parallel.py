"""Multi-GPU sharding of the hot path.

One process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests and in the same-device rehearsal).  The reference has no multi-device
path ("Collective call sites: none"); the rule used here: split the
largest operand, move the smallest tensor.

=====================================  ==========================  ===========================
op                                     partition                   collective
=====================================  ==========================  ===========================
crossprod(A, Y), Y tall and dense      rows (contracted dim) of    all-reduce of the ncol x K
                                       A and of Y                  result
colSums/colVars/... (col stats)        leaves (columns) by nnz     all-gather of the scalars
colSums on a ROW-sharded operand       rows                        all-reduce of ncol scalars
rowsum(A, group)                       leaves (columns) by nnz     all-gather of ngroup x ncol/N
A %*% Y (tall result)                  rows of A = rows of result  none (each rank owns rows)
crossprod(A) / crossprod(A, B) sparse  leaves of A (block of result all-gather of the other sparse
                                       rows per rank)              operand, once (ragged CSC)
colsum(A, group)                       leaves (columns) by nnz     all-reduce of nrow x ngroup
rowSums(A), 2-D                        rows (result owned)         all-gather of nrow / N sums
                                       or leaves                   or all-reduce of nrow sums
=====================================  ==========================  ===========================

Everything here works on device-resident operands (`DeviceCSC`) and calls the HIP library for
the local compute; `local=` hooks let the CPU tests plug the oracle in under the same sharding
and collective logic.  bench.py drives `ShardedCrossprod` for its N > 1 runs.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


# --------------------------------------------------------------------------------------------
# partitions
# --------------------------------------------------------------------------------------------
def row_block(nrow: int, rank: int, world: int, align: int = 1) -> Tuple[int, int]:
    """Contiguous row block of `rank`.  Sizes differ by at most `align` rows; with align > 1
    every boundary but the last is a multiple of it (128 = the row panels of the product kernel)."""
    units = (nrow + align - 1) // align
    base, rem = divmod(units, world)
    u0 = rank * base + min(rank, rem)
    u1 = u0 + base + (1 if rank < rem else 0)
    return min(u0 * align, nrow), min(u1 * align, nrow)


def row_shard_csc(col_ptr, row_idx, val, r0: int, r1: int):
    """Leaves restricted to rows [r0, r1), offsets rebased to r0.  Offsets are
    ascending inside a leaf (src/leaf_utils.h:12-15) so this is a filter that
    keeps order.  Works on numpy arrays or torch tensors (any device)."""
    if isinstance(row_idx, torch.Tensor):
        keep = (row_idx >= r0) & (row_idx < r1)
        csum = torch.zeros(row_idx.numel() + 1, dtype=torch.int64, device=row_idx.device)
        csum[1:] = torch.cumsum(keep.to(torch.int64), 0)
        new_ptr = csum[col_ptr]
        return new_ptr, (row_idx[keep] - r0).to(torch.int32), val[keep]
    keep = (row_idx >= r0) & (row_idx < r1)
    csum = np.zeros(len(row_idx) + 1, dtype=np.int64)
    csum[1:] = np.cumsum(keep)
    return csum[np.asarray(col_ptr)], (row_idx[keep] - r0).astype(np.int32), val[keep]


def col_blocks_by_nnz(col_ptr, world: int) -> List[Tuple[int, int]]:
    """Leaf ranges [c0, c1) per rank with ~equal nnz (prefix sums of col_ptr)."""
    cp = col_ptr.cpu().numpy() if isinstance(col_ptr, torch.Tensor) else np.asarray(col_ptr)
    ncol, nnz = len(cp) - 1, int(cp[-1])
    cuts = [0]
    for g in range(1, world):
        cuts.append(int(np.searchsorted(cp, nnz * g / world, side="left")))
    cuts.append(ncol)
    cuts = np.maximum.accumulate(np.clip(cuts, 0, ncol))
    return [(int(cuts[g]), int(cuts[g + 1])) for g in range(world)]


def col_shard_csc(col_ptr, row_idx, val, c0: int, c1: int):
    """Leaves [c0, c1) as a CSC of their own (col_ptr rebased); views, no copy of the payload."""
    k0, k1 = int(col_ptr[c0]), int(col_ptr[c1])
    return col_ptr[c0:c1 + 1] - k0, row_idx[k0:k1], val[k0:k1]


# --------------------------------------------------------------------------------------------
# collectives
# --------------------------------------------------------------------------------------------
def _world(group=None) -> int:
    return dist.get_world_size(group) if dist.is_initialized() else 1


_FORCE = False


def force_collectives(on: bool = True) -> None:
    """Run the collectives even in a group of ONE rank (the all-reduce of one rank is the identity): what a one-GPU
    box can tell about the RCCL path -- that the library loads beside libsvt_hip.so, that the asynchronous all-reduce
    and the two result buffers of `ShardedCrossprod` work against the real backend -- and nothing about scaling
    (tests/test_rccl_one_rank.py; `bench.py --force-collectives`)."""
    global _FORCE
    _FORCE = bool(on)


def _reduces(group=None) -> bool:
    """Is there a collective to run?  More than one rank, or one rank with force_collectives()."""
    return dist.is_initialized() and (dist.get_world_size(group) > 1 or _FORCE)


def _rank(group=None) -> int:
    return dist.get_rank(group) if dist.is_initialized() else 0


def sharded_crossprod(local_crossprod: Callable[[], torch.Tensor], group=None) -> torch.Tensor:
    """`local_crossprod()` returns this rank's ncol x K partial (any layout, the
    same on every rank); the sum over ranks is returned on every rank."""
    part = local_crossprod()
    if _reduces(group):
        dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group)
    return part


def gather_columns(local: torch.Tensor, sizes: Sequence[int], group=None) -> torch.Tensor:
    """Concatenate per-rank result slices along dim 0 (col stats: one scalar per leaf; rowsum: one
    row of `ngroup` sums per leaf).  `sizes[r]` = leading extent of rank r's slice."""
    if not _reduces(group):
        return local
    tail = tuple(local.shape[1:])
    outs = [torch.empty((int(n),) + tail, dtype=local.dtype, device=local.device) for n in sizes]
    if len(set(int(n) for n in sizes)) == 1:
        dist.all_gather(outs, local.contiguous(), group=group)
    else:
        _all_gather_ragged(outs, local.contiguous(), group)
    return torch.cat(outs)


def _all_gather_ragged(outs, local, group):
    """Ragged all-gather as one broadcast per member.  `dist.broadcast(src=)` takes a GLOBAL
    rank: inside a sub-group the member index must be translated."""
    world = dist.get_world_size(group)
    me = dist.get_rank(group)
    for member in range(world):
        buf = outs[member]
        if me == member:
            buf.copy_(local)
        src = dist.get_global_rank(group, member) if group is not None else member
        dist.broadcast(buf, src=src, group=group)


# --------------------------------------------------------------------------------------------
# device-resident sharded operands (HIP path)
# --------------------------------------------------------------------------------------------
def shard_rows_device(A, rank: int, world: int, align: int = 128):
    """Row block `rank` of a resident operand as a DeviceCSC of its own (device-side filter)."""
    from .device import DeviceCSC
    r0, r1 = row_block(A.nrow, rank, world, align)
    cp, ri, v = row_shard_csc(A.col_ptr, A.row_idx, A.val, r0, r1)
    return DeviceCSC(r1 - r0, cp, ri, v), (r0, r1)


def shard_cols_device(A, rank: int, world: int):
    """Leaf block `rank` (balanced by nnz) of a resident operand; returns (shard, blocks)."""
    from .device import DeviceCSC
    blocks = col_blocks_by_nnz(A.col_ptr, world)
    c0, c1 = blocks[rank]
    cp, ri, v = col_shard_csc(A.col_ptr, A.row_idx, A.val, c0, c1)
    return DeviceCSC(A.nrow, cp.contiguous(), ri, v), blocks


def shard_axis_device(A, dim, axis: int, rank: int, world: int):
    """An N-d array (extents `dim`, dim[0] == A.nrow, leaves in R's order: axis 1 fastest) cut along
    `axis` >= 1: rank r keeps the leaves whose coordinate on that axis lies in its block.  Returns
    (shard as a DeviceCSC, its extents, (lo, hi) on the axis).  BASELINE config 5: shard axis 1 (the
    2e4 columns) of the 2e4 x 2e4 x 64 array, so that every rank owns whole output cells of
    `rowSums(dims=2)` (a sum over axis 2) and whole leaves for the column statistics: no collective
    except the gather of the results."""
    from .device import DeviceCSC
    dim = [int(d) for d in dim]
    assert 1 <= axis < len(dim) and dim[0] == A.nrow
    lo, hi = row_block(dim[axis], rank, world)
    dev = A.val.device
    nleaves = A.ncol
    inner = int(np.prod(dim[1:axis], dtype=np.int64)) if axis > 1 else 1      # leaves per step of `axis`
    j = torch.arange(nleaves, device=dev)
    keep = ((j // inner) % dim[axis] >= lo) & ((j // inner) % dim[axis] < hi)
    counts = (A.col_ptr[1:] - A.col_ptr[:-1])[keep]
    new_ptr = torch.zeros(int(keep.sum()) + 1, dtype=torch.int64, device=dev)
    new_ptr[1:] = torch.cumsum(counts, 0)
    starts = A.col_ptr[:-1][keep]
    # source position of every kept nonzero: start of its leaf + offset inside the leaf
    leaf_of = torch.repeat_interleave(torch.arange(counts.numel(), device=dev), counts)
    src = starts[leaf_of] + (torch.arange(int(new_ptr[-1]), device=dev) - new_ptr[:-1][leaf_of])
    new_dim = list(dim)
    new_dim[axis] = hi - lo
    return DeviceCSC(A.nrow, new_ptr, A.row_idx[src], A.val[src]), tuple(new_dim), (lo, hi)


def gather_axis(local: torch.Tensor, out_dim, axis: int, blocks, group=None) -> torch.Tensor:
    """Inverse of the cut for a result laid out like the array (R order over `out_dim`, rank r holding
    the slab blocks[r] of `axis`): all-gather of the slabs, then every slab goes to its place."""
    if not _reduces(group):
        return local
    out_dim = [int(d) for d in out_dim]
    inner = int(np.prod(out_dim[:axis], dtype=np.int64)) if axis > 0 else 1
    outer = int(np.prod(out_dim[axis + 1:], dtype=np.int64)) if axis + 1 < len(out_dim) else 1
    sizes = [inner * (b[1] - b[0]) * outer for b in blocks]
    flat = gather_columns(local.reshape(-1), sizes, group)
    out = torch.empty(inner * out_dim[axis] * outer, dtype=local.dtype, device=local.device).view(outer, out_dim[axis], inner)
    off = 0
    for (lo, hi), n in zip(blocks, sizes):
        out[:, lo:hi, :] = flat[off:off + n].view(outer, hi - lo, inner)
        off += n
    return out.reshape(-1)


class PeerReducer:
    """Sum of one (K, ncol) partial per rank on every rank WITHOUT a collective kernel: every rank copies
    its partial into a staging slot it owns in every peer's memory (peer-to-peer copies: the copy engines
    on a GPU node, no CU taken from the product kernel, which fills every CU's registers and LDS so that
    an RCCL kernel cannot run beside it -- DESIGN.md section 5), and every rank adds up its `world` slots
    with one small local kernel.  10 MB per peer and step at BASELINE config 2a, over all 7 xGMI links at once.

    Opt-in (`ShardedCrossprod(..., reducer="peer")`, `bench.py --reduce peer`): the default stays RCCL's
    all-reduce.  Only the control flow can be exercised on the boxes this was written on (one GPU:
    `tests/workers/dist_gpu_worker.py` runs two ranks on one device; `tests/test_distributed_cpu.py` runs it on
    CPU over shared memory); that the copies take the copy engines and overlap the product is what an 8-GPU
    node has to show.

    Windows: `staging[b][p]` (this rank's memory, written by rank p) is mapped into every peer through
    torch's CUDA IPC (shared memory on CPU).  Ordering between processes: per buffer an interprocess event
    `pushed` (recorded behind this rank's copies) and one `consumed` (behind this rank's sum), plus two
    counters per buffer in host shared memory that say the record has been ISSUED -- a wait on an
    interprocess event sees the latest record issued at the time of the wait, so the waiter first spins on
    the counter (host side, microseconds: all ranks run the same loop), then lets its stream wait for the event.
    Per step and rank: world copies, one sum, 2 event records, 2 * (world - 1) event waits."""

    def __init__(self, shape, dtype, device, group=None, nbuf: int = 2, timeout_s: float = 60.0):
        import pickle
        from multiprocessing.reduction import ForkingPickler
        import torch.multiprocessing as tmp
        self.group, self.nbuf, self.timeout_s = group, nbuf, timeout_s
        self.world, self.me = _world(group), _rank(group)
        self.cuda = torch.device(device).type == "cuda"
        self.shape = tuple(shape)
        # (handles travel as bytes through all_gather_object: host tensors are shared by file name, not by descriptor --
        # the process-wide strategy is switched only while the windows are made and opened, and put back)
        strategy = tmp.get_sharing_strategy()
        tmp.set_sharing_strategy("file_system")
        try:
            self._open_windows(shape, dtype, device, group, nbuf, pickle, ForkingPickler)
        finally:
            tmp.set_sharing_strategy(strategy)
        self.gen = [0] * nbuf          # generation of the last push / sum of every buffer
        self.pending = [False] * nbuf
        self.closed = False

    def _open_windows(self, shape, dtype, device, group, nbuf, pickle, ForkingPickler):
        self.staging = torch.zeros((nbuf, self.world) + self.shape, dtype=dtype, device=device)
        self.flags = torch.zeros((2, nbuf), dtype=torch.int64)          # [pushed | consumed][buffer], host memory
        if not self.cuda:
            self.staging.share_memory_()
        self.flags.share_memory_()
        mine = {"staging": bytes(ForkingPickler.dumps(self.staging)), "flags": bytes(ForkingPickler.dumps(self.flags))}
        if self.cuda:
            self.copy_stream = torch.cuda.Stream(device=device)
            self.pushed_ev = [torch.cuda.Event(enable_timing=False, interprocess=True) for _ in range(nbuf)]
            self.consumed_ev = [torch.cuda.Event(enable_timing=False, interprocess=True) for _ in range(nbuf)]
            self.part_ready = [torch.cuda.Event() for _ in range(nbuf)]
            for ev in self.pushed_ev + self.consumed_ev:
                ev.record()                                              # (an event must have been recorded to be shared)
            mine["pushed"] = [ev.ipc_handle() for ev in self.pushed_ev]
            mine["consumed"] = [ev.ipc_handle() for ev in self.consumed_ev]
            mine["device"] = torch.device(device).index
        self.pushes_done = [None] * nbuf
        handles = [None] * self.world
        dist.all_gather_object(handles, mine, group=group)
        self.peer_staging, self.peer_flags, self.peer_pushed, self.peer_consumed = [], [], [], []
        for p, h in enumerate(handles):
            if p == self.me:
                self.peer_staging.append(self.staging); self.peer_flags.append(self.flags)
                self.peer_pushed.append(getattr(self, "pushed_ev", None)); self.peer_consumed.append(getattr(self, "consumed_ev", None))
                continue
            self.peer_staging.append(pickle.loads(h["staging"]))
            self.peer_flags.append(pickle.loads(h["flags"]))
            if self.cuda:
                dev = torch.device("cuda", h["device"])
                self.peer_pushed.append([torch.cuda.Event.from_ipc_handle(dev, x) for x in h["pushed"]])
                self.peer_consumed.append([torch.cuda.Event.from_ipc_handle(dev, x) for x in h["consumed"]])
            else:
                self.peer_pushed.append(None); self.peer_consumed.append(None)
        if self.world > 1:
            dist.barrier(group=group)          # every rank has opened every window before anyone's tensors can go away

    def close(self):
        """Collective: every rank drops its views of the peers' windows (the IPC mappings and the shared-memory
        files behind the host counters go with them), after a barrier that says nobody is still copying."""
        if getattr(self, "closed", True):
            return
        if self.cuda:
            torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier(group=self.group)
        self.peer_staging = self.peer_flags = self.peer_pushed = self.peer_consumed = None
        self.staging = self.flags = None
        self.closed = True

    def _spin(self, flags, row, b, want):
        import time
        t0 = time.monotonic()
        while int(flags[row, b]) < want:
            if time.monotonic() - t0 > self.timeout_s:
                raise RuntimeError(f"PeerReducer: rank {self.me} waited {self.timeout_s} s for a peer (flag {row}, buffer {b})")
            time.sleep(0)

    def push(self, b: int, part: torch.Tensor):
        """Send `part` (this rank's partial of the step that uses buffer b) to every rank's staging slot."""
        self.gen[b] += 1
        g = self.gen[b]
        if self.cuda:
            self.part_ready[b].record()                                   # behind the product on the current stream
            with torch.cuda.stream(self.copy_stream):
                self.copy_stream.wait_event(self.part_ready[b])
                for p in range(self.world):
                    if p != self.me:
                        # the peer has summed what this slot held (generation g - 1) before it is overwritten
                        self._spin(self.peer_flags[p], 1, b, g - 1)
                        self.copy_stream.wait_event(self.peer_consumed[p][b])
                    self.peer_staging[p][b, self.me].copy_(part, non_blocking=True)
                self.pushed_ev[b].record(self.copy_stream)
                self.pushes_done[b] = torch.cuda.Event()
                self.pushes_done[b].record(self.copy_stream)
        else:
            for p in range(self.world):
                if p != self.me:
                    self._spin(self.peer_flags[p], 1, b, g - 1)
                self.peer_staging[p][b, self.me].copy_(part)
        self.flags[0, b] = g                                              # "my record of pushed[b] has been issued"
        self.pending[b] = True

    def before_overwrite(self, b: int):
        """Call before the product writes the partial of buffer b again: its copies must have left."""
        if self.cuda and self.pushes_done[b] is not None:
            torch.cuda.current_stream().wait_event(self.pushes_done[b])

    def finish(self, b: int, out: torch.Tensor):
        """Sum of the `world` partials of buffer b into `out` (on the current stream)."""
        if not self.pending[b]:
            return
        g = self.gen[b]
        for p in range(self.world):
            if p != self.me:
                self._spin(self.peer_flags[p], 0, b, g)
                if self.cuda:
                    torch.cuda.current_stream().wait_event(self.peer_pushed[p][b])
        if self.cuda and self.pushes_done[b] is not None:
            torch.cuda.current_stream().wait_event(self.pushes_done[b])      # my own slot
        torch.sum(self.staging[b], dim=0, out=out)
        if self.cuda:
            self.consumed_ev[b].record()
        self.flags[1, b] = g                                              # "my record of consumed[b] has been issued"
        self.pending[b] = False


class ShardedCrossprod:
    """crossprod(A, Y) with A and Y sharded on rows.  Every rank holds its row block of A (with
    the panel-blocked layout of that block, built once) and of Y; a step computes the rank's
    ncol x K partial with the product kernels and sums the partials over the ranks.

    reducer = "rccl" (default): all-reduce (`torch.distributed`, RCCL on the GPU box).  Two result buffers:
    the all-reduce of step i runs on the collective's stream while the product of step i + 1 runs on the
    compute stream; step i + 2 waits for it before it reuses the buffer.
    reducer = "peer": `PeerReducer` -- peer-to-peer copies of the partials and one local sum, no collective
    kernel; the sum of step i is taken at the start of step i + 1 (its copies fly during the product in between).
    `result()` returns the last finished buffer, laid out (K, ncol) C-contiguous = the column-major ncol x K
    matrix R would get.
    spare_cus = n: the product kernel leaves n CUs idle (process-wide setting of the library) so that the
    collective's kernels can start beside it; None = leave the setting alone.
    """

    def __init__(self, A_local, K: int, group=None, cbw: int = 0, wpb: int = 0, logr: int = 0,
                 reducer: str = "rccl", spare_cus=None):
        from .device import PbcPlan, set_spare_cus
        self.A, self.K, self.group = A_local, int(K), group
        if spare_cus is not None:
            set_spare_cus(spare_cus)
        self.plan = PbcPlan(A_local, K, cbw, wpb, logr)
        dev = A_local.val.device
        world = _world(group)
        nbuf = 2 if _reduces(group) else 1
        self.outs = [torch.zeros((self.K, A_local.ncol), dtype=torch.float64, device=dev) for _ in range(nbuf)]
        self.pending = [None] * nbuf
        self.stepno = 0
        self.peer = None
        if reducer == "peer" and world > 1:
            self.peer = PeerReducer((self.K, A_local.ncol), torch.float64, dev, group, nbuf)
            self.parts = [torch.zeros_like(self.outs[0]) for _ in range(nbuf)]
        elif reducer not in ("rccl", "peer"):
            raise ValueError("reducer must be 'rccl' or 'peer'")

    def _pick(self) -> int:
        i = self.stepno % len(self.outs)
        self.stepno += 1
        if self.pending[i] is not None:
            self.pending[i].wait()           # stream-level wait: buffer i is free again
            self.pending[i] = None
        return i

    def step(self, Y_local: torch.Tensor, events=None) -> int:
        """One product + reduction; Y_local is this rank's (K, rows) block.  `events` = a pair of
        torch events recorded around the dominant kernel."""
        i = self._pick()
        ld = Y_local.shape[1]
        target = self.outs[i]
        if self.peer is not None:
            prev = (i - 1) % len(self.outs)
            self.peer.finish(prev, self.outs[prev])          # the previous step's sum (its copies had a whole product to land)
            self.peer.before_overwrite(i)
            target = self.parts[i]
        if events is not None:
            events[0].record()
        self.plan.run_phase(1, Y_local, ld, target)
        if events is not None:
            events[1].record()
        self.plan.run_phase(2, Y_local, ld, target)
        if self.peer is not None:
            self.peer.push(i, target)
        elif _reduces(self.group):
            self.pending[i] = dist.all_reduce(self.outs[i], op=dist.ReduceOp.SUM, group=self.group,
                                              async_op=True)
        return i

    def wait(self):
        if self.peer is not None:
            for i in range(len(self.outs)):
                self.peer.finish(i, self.outs[i])
            return
        for i, w in enumerate(self.pending):
            if w is not None:
                w.wait()
                self.pending[i] = None

    def result(self) -> torch.Tensor:
        self.wait()
        return self.outs[(self.stepno - 1) % len(self.outs)]

    def close(self):
        """Collective when the peer reducer is in use: finishes what is pending and closes its windows."""
        self.wait()
        if self.peer is not None:
            self.peer.close()
            self.peer = None


def sharded_colsums_rows(A_local, group=None, local: Optional[Callable] = None) -> torch.Tensor:
    """colSums of a ROW-sharded operand: per-rank partial sums of every column, all-reduced
    (ncol doubles on the wire; BASELINE config 4 runs it beside the row-sharded crossprod)."""
    if local is not None:
        part = local()
    else:
        from .device import colstats
        part, _ = colstats(A_local, "sum")
    if _reduces(group):
        dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group)
    return part


def sharded_colstats(A_cols, blocks, op: str, na_rm=False, group=None,
                     local: Optional[Callable] = None) -> torch.Tensor:
    """A column statistic of a LEAF-sharded operand: every leaf is whole on one rank, so the
    values are the single-GPU ones; the scalars are all-gathered."""
    if local is not None:
        loc = local()
    else:
        from .device import colstats
        loc, _ = colstats(A_cols, op, na_rm)
    return gather_columns(loc, [b[1] - b[0] for b in blocks], group)


def sharded_rowsum(A_cols, blocks, grp: torch.Tensor, ngroup: int, na_rm=False, group=None,
                   local: Optional[Callable] = None) -> torch.Tensor:
    """rowsum(A, group) of a LEAF-sharded operand (columns are independent:
    src/rowsum_methods.c:86-103): every rank computes the ngroup sums of its own leaves, the
    (ncol_local, ngroup) slabs are all-gathered into the (ncol, ngroup) result = column-major
    ngroup x ncol.  `grp` (one int per row) is replicated."""
    if local is not None:
        loc = local()
    else:
        from .device import rowsum
        loc = rowsum(A_cols, grp, ngroup, na_rm)
    return gather_columns(loc, [b[1] - b[0] for b in blocks], group)


def sharded_matmul(A_rows_t_plan, Y2: torch.Tensor, out_local: torch.Tensor):
    """A %*% Y2 with A sharded on rows: rank g owns rows [r0, r1) of the result, computed as
    crossprod(t(A_g), Y2) with the panel-blocked layout of t(A_g); Y2 is replicated and there is
    no collective."""
    A_rows_t_plan.run(Y2, Y2.shape[1], out_local)
    return out_local


def sharded_matmul_sparse(A_rows, B, out_local=None):
    """A %*% B, both sparse (BASELINE config 3), with A sharded on rows and B (small) replicated: rank g owns
    rows [r0, r1) of the result and computes them with the row-panel kernel on its block of A
    (`device.matmul_csc_csc`); no collective.  Returns (out_local, not_finite): a nonzero flag on ANY rank
    means the product has to be redone by the dense route on every rank (the caller reduces the flags with
    `dist.all_reduce(flag, op=MAX)` when it needs the decision to be collective)."""
    from .device import matmul_csc_csc
    return matmul_csc_csc(A_rows, B, out=out_local)


# --------------------------------------------------------------------------------------------
# the rows of SURVEY.md section 8e that move a sparse operand or reduce a dense one
# --------------------------------------------------------------------------------------------
def allgather_csc(col_ptr, row_idx, val, blocks, group=None):
    """Leaf blocks (one per rank, `blocks[r]` = its leaf range) -> the whole operand on every rank:
    ragged all-gather of the leaf lengths, the offsets and the values; col_ptr rebuilt by a prefix sum.
    This is the one transfer of unary crossprod(A) / crossprod(A, B): every rank needs all leaves of the
    other operand (6 GB once at BASELINE config 4; src/SparseMatrix_mult.c:827-908, 1037-1101 walk them
    in place)."""
    lens = (col_ptr[1:] - col_ptr[:-1]).contiguous()
    ncols = [b[1] - b[0] for b in blocks]
    all_lens = gather_columns(lens, ncols, group)
    nnzs = []
    off = 0
    for n in ncols:
        nnzs.append(int(all_lens[off:off + n].sum()))
        off += n
    all_idx = gather_columns(row_idx.contiguous(), nnzs, group)
    all_val = gather_columns(val.contiguous(), nnzs, group)
    full_ptr = torch.zeros(all_lens.numel() + 1, dtype=torch.int64, device=all_lens.device)
    full_ptr[1:] = torch.cumsum(all_lens, 0)
    return full_ptr, all_idx, all_val


def sharded_crossprod_sparse(local_product: Callable, B_block, blocks_B, group=None, gather_result=True,
                             blocks_A=None):
    """crossprod(A, B), both sparse (B = A for the unary form): rank r holds leaf block r of A and of B.
    `B_block` = (col_ptr, row_idx, val) of the rank's block of B; `local_product(B_full)` returns the rank's
    block of result rows, (ncol_A_local, ncol_B), from its block of A and the gathered B.  With
    `gather_result` the row blocks are all-gathered into the ncol_A x ncol_B matrix (`blocks_A` = leaf
    ranges of A per rank)."""
    B_full = allgather_csc(*B_block, blocks_B, group)
    part = local_product(B_full)
    if not gather_result or not _reduces(group):
        return part
    return gather_columns(part.contiguous(), [b[1] - b[0] for b in (blocks_A or blocks_B)], group)


def device_sparse_crossprod_block(A_block):
    """`local_product` for sharded_crossprod_sparse() on the device: the rank's leaf block of A (a DeviceCSC) against
    the gathered B through the sparse-aware kernel (svt_dev_crossprod_csc_csc on t(A_block), built once here).  The
    rank's rows of the result come back as the (ncol_A_local, ncol_B) tensor gather_columns() concatenates; the
    kernel's not-finite flag of the last product is kept in `.flag` (nonzero: that product has to be redone by
    the dense-buffer route, svt_dev_crossprod_csc_csc_dense_buffer)."""
    from .device import DeviceCSC, crossprod_csc_csc
    At = A_block.t()

    def local_product(B_full):
        cp, ri, v = B_full
        B = DeviceCSC(A_block.nrow, cp, ri, v)
        out, flag = crossprod_csc_csc(At, B)          # (ncol_B, ncol_A_local) = column-major ncol_A_local x ncol_B
        local_product.flag = flag
        return out.t().contiguous()
    local_product.flag = None
    return local_product


def sharded_colsum(local_colsum: Callable, group=None) -> torch.Tensor:
    """colsum(A, group) with the leaves sharded: a group of columns spans ranks, so every rank adds the
    leaves it holds into its own nrow x ngroup partial (src/rowsum_methods.c:204-255) and the partials
    are all-reduced.  (Integer input: int64 partials, NA / overflow flags reduced with MAX by the caller.)"""
    part = local_colsum()
    if _reduces(group):
        dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group)
    return part


def sharded_rowsums_2d(local_rowsums: Callable, blocks=None, group=None) -> torch.Tensor:
    """rowSums(A) of a 2-D operand.  `blocks` = row ranges per rank: the operand is sharded on rows, every
    rank owns its nrow / N sums and they are all-gathered (no reduction).  `blocks` = None: sharded on
    leaves, the nrow partial sums are all-reduced (8 MB at BASELINE config 2)."""
    part = local_rowsums()
    if not _reduces(group):
        return part
    if blocks is not None:
        return gather_columns(part.contiguous(), [b[1] - b[0] for b in blocks], group)
    dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group)
    return part
