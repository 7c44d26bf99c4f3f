"""Synthetic inputs: Graph500 RMAT and Erdos-Renyi graphs as host Matrix objects.

PRNG: numpy Generator(PCG64(seed)); default seed 42.  RMAT: per edge, `scale`
levels, quadrant probabilities (a,b,c,d) = (0.57,0.19,0.19,0.05), edge factor
16, no vertex permutation; duplicate edges are merged by summing their values
(GrB_Matrix_build with GrB_PLUS), so values are never all equal and the
reference would not switch to its iso kernels on such input.

rmat_lcg is a second RMAT generator with a different PRNG (a 64-bit LCG): it rebuilds, edge for edge, the
inputs of the recorded runs of the reference library (tests/golden/reference_run_checksums.json).
"""
import numpy as np
import scipy.sparse as sp

from .binding import NP_TYPES, Type
from .host import Matrix


def rmat_edges(scale, edge_factor=16, seed=42, a=0.57, b=0.19, c=0.19):
    rng = np.random.default_rng(seed)
    n = 1 << scale
    m = edge_factor * n
    rows = np.zeros(m, np.int64)
    cols = np.zeros(m, np.int64)
    ab, abc = a + b, a + b + c
    for level in range(scale):
        r = rng.random(m, dtype=np.float32)
        # quadrant: a -> (0,0), b -> (0,1), c -> (1,0), d -> (1,1)
        row_bit = (r >= ab)
        col_bit = ((r >= a) & (r < ab)) | (r >= abc)
        rows |= row_bit.astype(np.int64) << (scale - 1 - level)
        cols |= col_bit.astype(np.int64) << (scale - 1 - level)
    return n, rows, cols, rng


def rmat(scale, edge_factor=16, seed=42, type=Type.FP64, values="uniform", by_row=True, symmetric=False,
         drop_diagonal=False):
    """n-by-n RMAT matrix.  values: 'uniform' in (0,1], 'ones', or 'int' in [1,255]."""
    n, rows, cols, rng = rmat_edges(scale, edge_factor, seed)
    if drop_diagonal:
        keep = rows != cols
        rows, cols = rows[keep], cols[keep]
    if symmetric:
        rows, cols = np.concatenate([rows, cols]), np.concatenate([cols, rows])
    m = len(rows)
    if values == "uniform":
        v = 1.0 - rng.random(m)
    elif values == "int":
        v = rng.integers(1, 256, m).astype(np.float64)
    else:
        v = np.ones(m)
    S = sp.coo_matrix((v, (rows, cols)), shape=(n, n)).tocsr()      # duplicates summed
    S.sum_duplicates()
    S.sort_indices()
    if values == "ones":
        S.data[:] = 1
    return Matrix.from_scipy(S.astype(NP_TYPES[type]), type=type, by_row=by_row)


def rmat_lcg_edges(scale, edge_factor=16, seed=42, a=0.57, b=0.19, c=0.19):
    """The RMAT edge list of the inputs the reference library was run on when its results for this path were
    recorded (RMAT-10/12/16/18: entry counts, checksums, triangle counts, frontier sizes).  PRNG: the 64-bit
    linear congruential generator x <- x * 6364136223846793005 + 1442695040888963407 (mod 2^64) started at
    x = seed; each new state gives one uniform number, its top 53 bits / 2^53.  One draw per level per edge,
    edges in order, the first level is the most significant bit; quadrants as in rmat_edges."""
    n = 1 << scale
    m = edge_factor * n
    N = m * scale
    mul, inc = np.uint64(6364136223846793005), np.uint64(1442695040888963407)
    # all N states at once: x_k = A_k * seed + C_k with A_k = mul^k, C_k = inc * (mul^(k-1) + ... + 1),
    # built by doubling (A_{s+t} = A_s * A_t, C_{s+t} = A_t * C_s + C_t), arithmetic mod 2^64
    A, Cc = np.empty(N, np.uint64), np.empty(N, np.uint64)
    A[0], Cc[0] = mul, inc
    filled = 1
    with np.errstate(over="ignore"):
        while filled < N:
            k = min(filled, N - filled)
            Ak, Ck = A[filled - 1], Cc[filled - 1]
            A[filled:filled + k] = A[:k] * Ak
            Cc[filled:filled + k] = A[:k] * Ck + Cc[:k]
            filled += k
        state = A * np.uint64(seed) + Cc
    del A, Cc
    u = ((state >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))).reshape(m, scale)
    del state
    ab, abc = a + b, a + b + c
    rows, cols = np.zeros(m, np.int64), np.zeros(m, np.int64)
    for level in range(scale):
        r = u[:, level]
        rows |= (r >= ab).astype(np.int64) << (scale - 1 - level)
        cols |= (((r >= a) & (r < ab)) | (r >= abc)).astype(np.int64) << (scale - 1 - level)
    return n, rows, cols


def rmat_lcg(scale, edge_factor=16, seed=42, type=Type.FP64, by_row=True, symmetric=False, drop_diagonal=False,
             values="multiplicity"):
    """n-by-n RMAT matrix from rmat_lcg_edges.  values 'multiplicity': every edge carries 1 and duplicates are
    summed (GrB_Matrix_build with GrB_PLUS), so A(i,j) is the number of times the edge was drawn; 'ones': pattern
    only.  symmetric: A + A' as a pattern."""
    n, rows, cols = rmat_lcg_edges(scale, edge_factor, seed)
    if drop_diagonal:
        keep = rows != cols
        rows, cols = rows[keep], cols[keep]
    if symmetric:
        rows, cols = np.concatenate([rows, cols]), np.concatenate([cols, rows])
    S = sp.coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(n, n)).tocsr()
    S.sum_duplicates()
    S.sort_indices()
    if values == "ones" or symmetric:
        S.data[:] = 1
    return Matrix.from_scipy(S.astype(NP_TYPES[type]), type=type, by_row=by_row)


def erdos_renyi(n, nnz, seed=42, type=Type.FP64, by_row=True, ncols=None, values="uniform"):
    rng = np.random.default_rng(seed)
    ncols = n if ncols is None else ncols
    rows = rng.integers(0, n, nnz)
    cols = rng.integers(0, ncols, nnz)
    v = (1.0 - rng.random(nnz)) if values == "uniform" else rng.integers(1, 100, nnz).astype(np.float64)
    S = sp.coo_matrix((v, (rows, cols)), shape=(n, ncols)).tocsr()
    S.sum_duplicates()
    S.sort_indices()
    return Matrix.from_scipy(S.astype(NP_TYPES[type]), type=type, by_row=by_row)


def tril_strict(M):
    """L = tril(A,-1) of a host Matrix, same storage orientation."""
    S = sp.tril(M.to_scipy(), k=-1).tocsr()
    return Matrix.from_scipy(S.astype(NP_TYPES[M.type]), type=M.type, by_row=not M.is_csc)
