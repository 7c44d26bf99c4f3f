"""Row-block sharding of C = A*B over several GPUs (one process per GPU).

Rows of C depend on the same rows of A and on all of B, C(I,:) = A(I,:)*B, so the rows are
split into contiguous blocks balanced by flop count (the count GB_AxB_saxpy3_flopcount produces,
Source/GB_AxB_saxpy3_flopcount.c:81; RMAT rows follow a power law, so equal row counts would
not balance).  The only exchange is the all-gather of the right-hand operand when it is itself
distributed by rows (C = A*A): every rank contributes the entries of its row block.

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


class RowBlockExchange:
    """All-gather of the (i, x) arrays of a matrix distributed by row blocks.

    part[r]..part[r+1] are the rows of rank r and Ap the (replicated) row pointers, so entry
    ranges are known everywhere and only indices and values travel.  Blocks are padded to the
    longest one because all_gather wants equal shapes."""

    def __init__(self, dist, torch, Ap, part, rank, device, idx_dtype, val_dtype):
        self.dist, self.torch = dist, torch
        self.world = len(part) - 1
        self.rank = rank
        self.offs = [int(Ap[part[r]]) for r in range(self.world)]
        self.counts = [int(Ap[part[r + 1]] - Ap[part[r]]) for r in range(self.world)]
        self.pad = max(self.counts) if self.world > 1 else 0
        if self.world > 1:
            self.gi = [torch.empty(self.pad, dtype=idx_dtype, device=device) for _ in range(self.world)]
            self.gx = [torch.empty(self.pad, dtype=val_dtype, device=device) for _ in range(self.world)]
            self.si = torch.zeros(self.pad, dtype=idx_dtype, device=device)
            self.sx = torch.zeros(self.pad, dtype=val_dtype, device=device)

    def gather(self, my_i, my_x, out_i, out_x):
        """my_i/my_x: this rank's entries; out_i/out_x: full-length arrays to fill."""
        if self.world == 1:
            out_i.copy_(my_i); out_x.copy_(my_x)
            return
        c = self.counts[self.rank]
        self.si[:c] = my_i; self.sx[:c] = my_x
        if self.dist.get_backend() == "gloo" and self.si.is_cuda:
            # gloo moves host memory: device tensors are staged through the host (CPU rehearsals of the path only;
            # RCCL takes the device tensors as they are)
            for send, recv in ((self.si, self.gi), (self.sx, self.gx)):
                host = [self.torch.empty(self.pad, dtype=send.dtype) for _ in range(self.world)]
                self.dist.all_gather(host, send.cpu())
                for r in range(self.world):
                    recv[r].copy_(host[r])
        else:
            self.dist.all_gather(self.gi, self.si)
            self.dist.all_gather(self.gx, self.sx)
        for r in range(self.world):
            o, n = self.offs[r], self.counts[r]
            out_i[o: o + n] = self.gi[r][:n]
            out_x[o: o + n] = self.gx[r][:n]
