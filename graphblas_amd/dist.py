"""Row-block sharding of C = A*B over several GPUs (one process per GPU).

Rows of C depend on the same rows of A and on all of B, C(I,:) = A(I,:)*B, so the rows are
split into contiguous blocks balanced by flop count (the count GB_AxB_saxpy3_flopcount produces,
Source/GB_AxB_saxpy3_flopcount.c:81; RMAT rows follow a power law, so equal row counts would
not balance).  The only exchange is the all-gather of the right-hand operand when it is itself
distributed by rows (C = A*A): every rank contributes the entries of its row block.

The other products of the path shard the same way, by row blocks of the output: y(I) = A(I,:) x needs the
all-gather of x (VectorBlockExchange), a traversal step w(I)<!visited> = A(I,:) q the exchange of the
frontier (FrontierExchange: index lists while it is small, a dense reduce once it is large).

When B does not fit whole on every rank it is exchanged one column panel at a time instead
(ColumnPanelExchange): C(I, J) = A(I,:)*B(:, J), the panels of C are disjoint, nothing is reduced.

The helpers work on any torch.distributed backend ("nccl" is RCCL on ROCm; "gloo" on CPU).
"""
import numpy as np


def row_flops(Ap, Ai, Bp=None):
    """Multiply-adds per row i of A*B for CSR operands: sum over k in A(i,:) of nnz B(k,:)."""
    Bp = Ap if Bp is None else Bp
    per_entry = np.diff(Bp)[Ai]
    cs = np.concatenate([[0], np.cumsum(per_entry)])
    return (cs[Ap[1:]] - cs[Ap[:-1]]).astype(np.int64)


def split_by_flops(prefix, lo, hi, nparts):
    """Boundaries of nparts contiguous pieces of rows [lo,hi) with ~equal flops
    (prefix = exclusive cumsum of the row flops, length nrows+1)."""
    base, total = prefix[lo], prefix[hi] - prefix[lo]
    cuts = [lo]
    for r in range(1, nparts):
        k = int(np.searchsorted(prefix, base + total * r / nparts, side="left"))
        cuts.append(min(max(k, cuts[-1]), hi))
    cuts.append(hi)
    return cuts


def chunks_by_flops(prefix, lo, hi, max_flops):
    """Contiguous row chunks of [lo,hi), each with at most max_flops multiply-adds (at least one row)."""
    out, start = [], lo
    while start < hi:
        k = int(np.searchsorted(prefix, prefix[start] + max_flops, side="right")) - 1
        k = min(max(k, start + 1), hi)
        out.append((start, k))
        start = k
    return out


def _all_gather_padded(dist, torch, send, pad, world):
    """One collective: every rank contributes `pad` elements, the result is the flat [world * pad] buffer.
    gloo moves host memory, so device tensors are staged through the host there (CPU rehearsals of the
    path only; RCCL takes the device tensors as they are)."""
    flat = torch.empty(world * pad, dtype=send.dtype, device=send.device)
    if dist.get_backend() == "gloo" and send.is_cuda:
        host = torch.empty(world * pad, dtype=send.dtype)
        dist.all_gather_into_tensor(host, send.cpu())
        flat.copy_(host)
    else:
        dist.all_gather_into_tensor(flat, send)
    return flat


def all_gather_in_place(dist, torch, mine, out, offs, counts, rank):
    """All-gather of pieces of unequal length straight into their places: out[offs[r] : offs[r] + counts[r]]
    receives the piece of rank r, `mine` is this rank's.  Nothing is padded and nothing is copied afterwards:
    the slices of `out` are the receive buffers.  RCCL takes the list of unequal slices as it is (it moves each
    piece as a broadcast inside one group); gloo wants equal shares, so there every piece is one broadcast, staged
    through the host for device tensors."""
    world = len(counts)
    pieces = [out[offs[r]: offs[r] + counts[r]] for r in range(world)]
    if dist.get_backend() != "gloo":
        try:
            dist.all_gather(pieces, mine)
            return
        except (RuntimeError, ValueError, TypeError):       # a build that refuses unequal pieces: one broadcast each
            pass
    staged = dist.get_backend() == "gloo" and out.is_cuda
    for r in range(world):
        if counts[r] == 0:
            continue
        if staged:
            host = mine.cpu() if r == rank else torch.empty(counts[r], dtype=out.dtype)
            dist.broadcast(host, src=r)
            pieces[r].copy_(host)
        else:
            if r == rank:
                pieces[r].copy_(mine)
            dist.broadcast(pieces[r], src=r)


class RowBlockExchange:
    """All-gather of the (i, x) arrays of a matrix distributed by row blocks.

    part[r]..part[r+1] are the rows of rank r and Ap the (replicated) row pointers, so entry
    ranges are known everywhere and only indices and values travel: one all-gather per array, every
    block received at its own length straight into its place in the full array (flop-balanced blocks of a
    power-law matrix hold unequal numbers of entries; padding them to the longest would move
    world * max instead of the sum)."""

    def __init__(self, dist, torch, Ap, part, rank, device, idx_dtype, val_dtype, collective=None):
        self.dist, self.torch = dist, torch
        self.world = len(part) - 1
        self.rank = rank
        # collective=True keeps the collective in the loop even for one rank (a process group must be up)
        self.collective = (self.world > 1) if collective is None else bool(collective)
        self.offs = [int(Ap[part[r]]) for r in range(self.world)]
        self.counts = [int(Ap[part[r + 1]] - Ap[part[r]]) for r in range(self.world)]
        self.moved = sum(self.counts)                       # entries a gather moves; padded it was world * max
        self.padded = self.world * max(max(self.counts), 1)

    def gather(self, my_i, my_x, out_i, out_x):
        """my_i/my_x: this rank's entries; out_i/out_x: full-length arrays to fill."""
        if not self.collective:
            out_i.copy_(my_i); out_x.copy_(my_x)
            return
        for mine, out in ((my_i, out_i), (my_x, out_x)):
            all_gather_in_place(self.dist, self.torch, mine, out, self.offs, self.counts, self.rank)


class ColumnPanelExchange:
    """The right-hand operand when it does not fit whole on every rank: B, distributed by row blocks, is exchanged
    one column panel at a time.  The columns are cut into contiguous ranges J_0 .. J_{q-1} that hold about equal
    numbers of entries; every rank keeps its row block cut the same way (pieces B(K_r, J_c): CSR, global column
    indices), panel c of the whole matrix is the all-gather of the pieces c, each received at its own length
    straight into its place, and C(I, J_c) = A(I,:) * B(:, J_c).  The panels of C are disjoint column ranges, so
    nothing is accumulated or reduced, and a rank holds one panel (1/q of B) beside its own block.

    Set-up needs the rank's own block only (host arrays): one SUM all-reduce of the column counts fixes the cuts,
    one all-gather per panel of the per-row counts gives every rank the row pointers of the gathered panel.
      perm          permutation of the block's entries that lays the pieces one after the other (rows ascending
                    inside a piece); the caller keeps my_i[perm], my_x[perm]
      piece[c]      (start, end) of piece c in the permuted arrays
      panel_p[c]    row pointers (n + 1, int64) of the gathered panel c;  panel_nnz[c] its entries"""

    def __init__(self, dist, torch, part, rank, ncols, my_p, my_i, npanels, device, collective=None):
        self.dist, self.torch, self.rank = dist, torch, rank
        self.world = len(part) - 1
        self.collective = (self.world > 1) if collective is None else bool(collective)
        self.part = [int(v) for v in part]
        self.npanels = int(npanels)
        my_p = np.asarray(my_p, np.int64) - int(my_p[0])
        my_i = np.asarray(my_i, np.int64)
        rows = len(my_p) - 1
        assert rows == self.part[rank + 1] - self.part[rank]
        col_counts = np.bincount(my_i, minlength=ncols).astype(np.int64)
        col_counts = self._reduce_sum(col_counts, device)
        prefix = np.concatenate([[0], np.cumsum(col_counts)])
        cuts = [0]
        for c in range(1, self.npanels):
            k = int(np.searchsorted(prefix, prefix[-1] * c / self.npanels, side="left"))
            cuts.append(min(max(k, cuts[-1]), ncols))
        cuts.append(ncols)
        self.cuts = cuts
        pid = np.searchsorted(np.asarray(cuts[1:-1], np.int64), my_i, side="right")       # panel of every entry
        self.perm = np.argsort(pid, kind="stable")
        ends = np.cumsum(np.bincount(pid, minlength=self.npanels))
        self.piece = [(int(ends[c - 1]) if c else 0, int(ends[c])) for c in range(self.npanels)]
        row_of = np.repeat(np.arange(rows, dtype=np.int64), np.diff(my_p))
        n = self.part[-1]
        self.panel_p, self.panel_nnz, self.offs, self.counts = [], [], [], []
        blocks = [self.part[r + 1] - self.part[r] for r in range(self.world)]
        for c in range(self.npanels):
            mine = np.bincount(row_of[pid == c], minlength=rows).astype(np.int64)
            whole = torch.empty(n, dtype=torch.int64, device=device if self._on_device() else "cpu")
            if self.collective:
                all_gather_in_place(dist, torch, torch.from_numpy(mine).to(whole.device), whole, self.part[:-1], blocks, rank)
            else:
                whole.copy_(torch.from_numpy(mine))
            pp = np.concatenate([[0], np.cumsum(whole.cpu().numpy())]).astype(np.int64)
            self.panel_p.append(pp)
            self.panel_nnz.append(int(pp[-1]))
            self.offs.append([int(pp[self.part[r]]) for r in range(self.world)])
            self.counts.append([int(pp[self.part[r + 1]] - pp[self.part[r]]) for r in range(self.world)])
            assert self.counts[c][rank] == self.piece[c][1] - self.piece[c][0]
        self.moved = sum(self.panel_nnz)                    # entries one pass over all panels moves: nnz(B)

    def _on_device(self):
        return self.collective and self.dist.get_backend() == "nccl"

    def _reduce_sum(self, a, device):
        if not self.collective:
            return a
        t = self.torch.from_numpy(a).to(device if self._on_device() else "cpu")
        self.dist.all_reduce(t)
        return t.cpu().numpy()

    def gather(self, c, my_i_perm, my_x_perm, out_i, out_x):
        """Panel c: my_*_perm are the block's entries in piece order, out_* receive panel_nnz[c] entries."""
        a, b = self.piece[c]
        nz = self.panel_nnz[c]
        for mine, out in ((my_i_perm[a:b], out_i[:nz]), (my_x_perm[a:b], out_x[:nz])):
            if not self.collective:
                out.copy_(mine)
            else:
                all_gather_in_place(self.dist, self.torch, mine, out, self.offs[c], self.counts[c], self.rank)


class VectorBlockExchange:
    """A dense vector distributed by the same row blocks: y(I) is computed by the rank that owns rows I
    (y(I) = A(I,:) x needs all of x), and the blocks are all-gathered, each at its own length and straight into
    its place, into the full vector that the next product reads -- the exchange of a PageRank-shaped iteration
    x <- A x (GrB_mxv, Source/GrB_mxv.c:57-66)."""

    def __init__(self, dist, torch, part, rank, device, dtype, collective=None):
        self.dist, self.torch = dist, torch
        self.world, self.rank = len(part) - 1, rank
        self.collective = (self.world > 1) if collective is None else bool(collective)
        self.part = [int(v) for v in part]
        self.counts = [self.part[r + 1] - self.part[r] for r in range(self.world)]

    def gather(self, my_block, out):
        """my_block: this rank's entries y(I); out: the full-length vector to fill."""
        if not self.collective:
            out.copy_(my_block)
            return
        all_gather_in_place(self.dist, self.torch, my_block, out, self.part[:-1], self.counts, self.rank)


class FrontierExchange:
    """The frontier of a level-synchronous traversal (BFS parents with ANY_SECONDI, SSSP distances with
    MIN_PLUS), distributed by row blocks of the OUTPUT: rank r computes w(I)<!visited(I)> = A(I,:) q for its
    rows I -- GrB_mxv with the full frontier q as the second operand; on a symmetric graph the same step as
    the push form q'A of GrB_vxm (Source/GrB_vxm.c:63-72) -- and the pieces are exchanged so that every rank
    holds the whole next frontier.

    sparse(): a small frontier travels as (index, value) lists: the counts first, then ONE padded
    all-gather per array.  dense_min() / dense_or(): a large frontier travels as a dense n-vector reduced
    with MIN (distances) or as a bitmap reduced with MAX (presence).  No reduction of w itself is needed:
    the row blocks are disjoint."""

    def __init__(self, dist, torch, rank, world, device, collective=None):
        self.dist, self.torch, self.rank, self.world, self.device = dist, torch, rank, world, device
        self.collective = (world > 1) if collective is None else bool(collective)

    def sparse(self, my_idx, my_val):
        """my_idx: global indices of the entries this rank found (ascending), my_val their values.
        Returns the concatenation over ranks, in rank order (= ascending index order for row blocks)."""
        torch = self.torch
        if not self.collective:
            return my_idx, my_val
        cnt = torch.tensor([my_idx.numel()], dtype=torch.int64, device=self.device)
        counts = _all_gather_padded(self.dist, torch, cnt, 1, self.world).cpu().tolist()
        offs = [0]
        for c in counts[:-1]:
            offs.append(offs[-1] + c)
        out = []
        for mine in (my_idx, my_val):
            whole = torch.empty(sum(counts), dtype=mine.dtype, device=self.device)
            all_gather_in_place(self.dist, torch, mine, whole, offs, counts, self.rank)
            out.append(whole)
        return out[0], out[1]

    def dense_min(self, d):
        """d: a full-length vector that holds this rank's block and the monoid identity (+inf) elsewhere."""
        if self.collective:
            self._all_reduce(d, self.dist.ReduceOp.MIN)
        return d

    def dense_or(self, b):
        """b: a full-length int8 bitmap that is zero outside this rank's block."""
        if self.collective:
            self._all_reduce(b, self.dist.ReduceOp.MAX)
        return b

    def _all_reduce(self, t, op):
        if self.dist.get_backend() == "gloo" and t.is_cuda:
            h = t.cpu()
            self.dist.all_reduce(h, op=op)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, op=op)


class ScalarReduce:
    """The reduction that ends a sharded masked dot product: for C<L> = L*L' (triangle counting, dot3) the mask is
    cut into row blocks, every rank multiplies C(I,:)<L(I,:)> = L(I,:)*L' against the whole of L (replicated, or
    gathered from the blocks with RowBlockExchange) and only the per-rank sums of C travel."""

    def __init__(self, dist, torch, world, device, collective=None):
        self.dist, self.torch, self.world, self.device = dist, torch, world, device
        self.collective = (world > 1) if collective is None else bool(collective)

    def sum_int64(self, value):
        on_device = self.collective and self.dist.get_backend() == "nccl"
        t = self.torch.tensor([int(value)], dtype=self.torch.int64, device=self.device if on_device else "cpu")
        if self.collective:
            self.dist.all_reduce(t)
        return int(t.item())
