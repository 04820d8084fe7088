"""
Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI;
"gloo" in the CPU tests).  Tiles are independent (decomposition.py:790-838), so the tile grid is
split into contiguous runs of tiles (k-outer order, so a run is a band of tile rows) and every rank
decomposes its run; the only exchange is the gather of the per-tile results before the global
recombination.  The Gaussian test matrix of tile b is keyed by b (counter-based RNG), so the result
does not depend on the number of ranks.
"""
from typing import List, Tuple


def tile_partition(n_tiles: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced [lo, hi) runs of tiles, one per rank (earlier ranks get the remainder)."""
    base, rem = divmod(int(n_tiles), int(world))
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


class Dist:
    """Thin view of the default process group (or a single-process stand-in)."""

    def __init__(self, enabled: bool = False, group=None):
        self.group = group
        self.enabled = False
        self.rank, self.world = 0, 1
        if enabled:
            import torch.distributed as dist

            if not dist.is_initialized():
                raise RuntimeError("distributed=True needs an initialised torch.distributed process group")
            self.rank = dist.get_rank(group)
            self.world = dist.get_world_size(group)
            self.enabled = self.world > 1

    def gather_runs(self, tensor, bounds):
        """Every rank owns rows [bounds[r][0], bounds[r][1]) of ``tensor`` (dim 0) and has filled them;
        after the call every rank holds all rows.  ONE all-gather (RCCL: every link carries 1 / world of the data once);
        runs may differ in length, so every rank contributes a block padded to the longest run."""
        if not self.enabled:
            return
        import torch.distributed as dist

        lens = [max(0, hi - lo) for lo, hi in bounds]
        longest = max(lens)
        if longest == 0:
            return
        lo, hi = bounds[self.rank]
        send = tensor.new_empty((longest,) + tuple(tensor.shape[1:]))
        send[:hi - lo] = tensor[lo:hi]
        recv = tensor.new_empty((self.world * longest,) + tuple(tensor.shape[1:]))
        dist.all_gather(list(recv.split(longest, dim=0)), send, group=self.group)
        for r, (rlo, rhi) in enumerate(bounds):
            if r != self.rank and rhi > rlo:
                tensor[rlo:rhi] = recv[r * longest:r * longest + (rhi - rlo)]

    def exchange_rows(self, tensor, bounds, needs):
        """Halo exchange.  Rank r owns rows [bounds[r][0], bounds[r][1]) of ``tensor``; needs[r] is a list of foreign row
        ranges (lo, hi) rank r must also hold (every rank passes the same ``bounds`` and ``needs``).  Afterwards every
        rank holds its needs; nothing else moves.  RCCL: one batch of point-to-point sends / receives between the ranks
        involved; other backends (gloo cannot send device tensors point to point): one broadcast per requested range."""
        if not self.enabled:
            return
        import torch.distributed as dist

        def owner(row):
            for r, (lo, hi) in enumerate(bounds):
                if lo <= row < hi:
                    return r
            raise ValueError("row {} has no owner".format(row))

        def grank(r):
            return dist.get_global_rank(self.group, r) if self.group is not None else r

        # split every requested range at ownership boundaries: (destination, source, lo, hi)
        moves = []
        for dst, ranges in enumerate(needs):
            for lo, hi in ranges:
                while lo < hi:
                    src = owner(lo)
                    top = min(hi, bounds[src][1])
                    if src != dst:
                        moves.append((dst, src, lo, top))
                    lo = top
        if dist.get_backend(self.group) == "nccl":
            ops = []
            for dst, src, lo, hi in moves:
                if src == self.rank:
                    ops.append(dist.P2POp(dist.isend, tensor[lo:hi], grank(dst), self.group))
                elif dst == self.rank:
                    ops.append(dist.P2POp(dist.irecv, tensor[lo:hi], grank(src), self.group))
            for req in (dist.batch_isend_irecv(ops) if ops else []):
                req.wait()
        else:
            # a broadcast is collective: every rank takes part; a rank that did not ask for a range receives it into
            # a scratch copy (its own rows of that range, if any, stay untouched)
            for src, lo, hi in sorted(set((src, lo, hi) for _, src, lo, hi in moves)):
                wanted = self.rank == src or any(d == self.rank and (s_, l_, h_) == (src, lo, hi) for d, s_, l_, h_ in moves)
                buf = tensor[lo:hi] if wanted else tensor[lo:hi].clone()
                dist.broadcast(buf, src=grank(src), group=self.group)

    def broadcast_object(self, obj):
        """Rank 0's value of a small Python object on every rank (seeds, the random frame sample)."""
        if not self.enabled:
            return obj
        import torch.distributed as dist

        box = [obj]
        dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        return box[0]

    def all_reduce(self, tensor):
        """In-place sum over the ranks (RCCL all-reduce on GPUs)."""
        if not self.enabled:
            return
        import torch.distributed as dist

        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group)

    def gather_rows_to_root(self, tensor, bounds, on_block=None):
        """Rank r has filled rows [bounds[r][0], bounds[r][1]) of ``tensor``; afterwards rank 0 holds all rows.
        ``on_block(lo, hi)`` runs on rank 0 as each remote block has arrived (e.g. to start its download).
        RCCL: point-to-point sends into rank 0, all receives posted at once; other backends (gloo cannot send
        device tensors): one broadcast per owner."""
        if not self.enabled:
            return
        import torch.distributed as dist

        def grank(r):
            return dist.get_global_rank(self.group, r) if self.group is not None else r

        if dist.get_backend(self.group) == "nccl":
            if self.rank == 0:
                todo = [(r, lo, hi) for r, (lo, hi) in enumerate(bounds) if r != 0 and hi > lo]
                ops = [dist.P2POp(dist.irecv, tensor[lo:hi], grank(r), self.group) for r, lo, hi in todo]
                reqs = dist.batch_isend_irecv(ops) if ops else []
                for req in reqs:
                    req.wait()
                for _, lo, hi in todo:
                    if on_block is not None:
                        on_block(lo, hi)
            else:
                lo, hi = bounds[self.rank]
                if hi > lo:
                    for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, tensor[lo:hi], grank(0), self.group)]):
                        req.wait()
        else:
            for r, (lo, hi) in enumerate(bounds):
                if r != 0 and hi > lo:
                    dist.broadcast(tensor[lo:hi], src=grank(r), group=self.group)
                    if self.rank == 0 and on_block is not None:
                        on_block(lo, hi)

    def gather_blocks_to_root(self, block, shapes):
        """Every rank holds one contiguous 2-D block (``shapes[r]`` = its shape on rank r; a dimension may be 0); rank 0
        returns the list of all blocks (its own included), the other ranks return None.  RCCL: point-to-point sends into
        rank 0, all receives posted at once; other backends: one broadcast per owner."""
        if not self.enabled:
            return [block]
        import torch.distributed as dist

        def grank(r):
            return dist.get_global_rank(self.group, r) if self.group is not None else r

        nonempty = [r for r, sh in enumerate(shapes) if sh[0] * sh[1] > 0]
        out = None
        if self.rank == 0:
            out = [block if r == 0 else block.new_empty(tuple(sh)) for r, sh in enumerate(shapes)]
        if dist.get_backend(self.group) == "nccl":
            if self.rank == 0:
                ops = [dist.P2POp(dist.irecv, out[r], grank(r), self.group) for r in nonempty if r != 0]
                for req in (dist.batch_isend_irecv(ops) if ops else []):
                    req.wait()
            elif self.rank in nonempty:
                for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, block.contiguous(), grank(0), self.group)]):
                    req.wait()
        else:
            for r in nonempty:
                if r == 0:
                    continue
                buf = out[r] if self.rank == 0 else (block.contiguous() if self.rank == r else block.new_empty(tuple(shapes[r])))
                dist.broadcast(buf, src=grank(r), group=self.group)
        return out

    def barrier(self):
        if self.enabled:
            import torch.distributed as dist

            dist.barrier(self.group)
