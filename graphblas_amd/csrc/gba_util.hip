// Finishing passes around the multiply kernels:
//   matrix_wait   kill zombies + sort every vector   (Source/GB_wait.c, Source/GB_unjumble.c)
//   transpose     AT = A', bucket method             (Source/GB_transpose_bucket.c, GB_transpose_cast.c:1-38)
//   partition     flop-balanced 1-D split            (Source/GB_pslice.c:116)
#include "gba_internal.hpp"
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <vector>

namespace gba {

// --------------------------------------------------------------- count ones --
// Eight bitmap bytes per load; one atomic per workgroup (atomics on one address serialise at the
// memory side, about 12 ns each, so a launch keeps them to a few hundred).
__device__ inline int nonzero_bytes(uint64_t w) {
    const uint64_t low7 = 0x7f7f7f7f7f7f7f7fULL;
    return __popcll((w | ((w & low7) + low7)) & ~low7);
}

__global__ void __launch_bounds__(256)
count_ones_kernel(const int8_t *__restrict__ b, int64_t n, unsigned long long *__restrict__ out) {
    __shared__ unsigned long long part[4];
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t head = (int64_t)((8 - (reinterpret_cast<uintptr_t>(b) & 7)) & 7);   // bytes before the first aligned word
    if (head > n) head = n;
    const int64_t nwords = (n - head) >> 3;
    const uint64_t *w = reinterpret_cast<const uint64_t *>(b + head);
    unsigned long long c = 0;
    for (int64_t q = tid; q < nwords; q += stride) c += nonzero_bytes(w[q]);
    const int64_t tail = head + (nwords << 3);
    if (tid < head) c += (b[tid] != 0);
    if (tid < n - tail) c += (b[tail + tid] != 0);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        c = part[0] + part[1] + part[2] + part[3];
        if (c) atomicAdd(out, c);
    }
}

gba_info dev_count_ones(const int8_t *b, int64_t n, int64_t *out) {
    *out = 0;
    if (n <= 0) return GBA_SUCCESS;
    hipStream_t st = ctx().stream;
    DevBuf o; GBA_OK(o.alloc(8));
    GBA_HIP(hipMemsetAsync(o.p, 0, 8, st));
    int64_t blocks = ceil_div(n, 256 * 8 * 8); if (blocks > 512) blocks = 512; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(count_ones_kernel, dim3((unsigned)blocks), dim3(256), 0, st, b, n, o.as<unsigned long long>());
    GBA_HIP(hipMemcpyAsync(out, o.p, 8, hipMemcpyDeviceToHost, st));
    GBA_HIP(hipStreamSynchronize(st));
    return GBA_SUCCESS;
}

// ------------------------------------------------------------ kill zombies --
__global__ void __launch_bounds__(256)
count_live_kernel(int64_t nvec, const int64_t *__restrict__ Cp, const int64_t *__restrict__ Ci, int64_t *__restrict__ cnt) {
    const int lane = threadIdx.x & 63;
    int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (; j < nvec; j += stride) {
        const int64_t p1 = Cp[j + 1];
        int c = 0;
        for (int64_t p = Cp[j] + lane; p < p1; p += 64) c += (Ci[p] >= 0);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
        if (lane == 0) cnt[j] = c;
    }
}

// order-preserving compaction, one wavefront per vector
__global__ void __launch_bounds__(256)
compact_live_kernel(int64_t nvec, const int64_t *__restrict__ Cp, const int64_t *__restrict__ Ci,
                    const uint8_t *__restrict__ Cx, int esize, const int64_t *__restrict__ Np,
                    int64_t *__restrict__ Ni, uint8_t *__restrict__ Nx) {
    const int lane = threadIdx.x & 63;
    int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (; j < nvec; j += stride) {
        const int64_t p0 = Cp[j], p1 = Cp[j + 1];
        int64_t out = Np[j];
        for (int64_t base = p0; base < p1; base += 64) {
            const int64_t p = base + lane;
            const bool live = (p < p1) && (Ci[p] >= 0);
            const unsigned long long m = __ballot(live);
            if (live) {
                const int64_t q = out + __builtin_popcountll(m & ((1ull << lane) - 1));
                Ni[q] = Ci[p];
                for (int b = 0; b < esize; b++) Nx[q * esize + b] = Cx[p * esize + b];
            }
            out += __builtin_popcountll(m);
        }
    }
}

// ------------------------------------------------------------- vector sort --
// bitonic sort of one vector in LDS: (index, original position) pairs, then the
// values are permuted through LDS as well.  n <= SORT_MAX.
static constexpr int SORT_MAX = 4096;
static constexpr int SORT_THREADS = 256;

__global__ void __launch_bounds__(SORT_THREADS)
sort_vectors_lds_kernel(int64_t nlist, const int64_t *__restrict__ list, const int64_t *__restrict__ Cp,
                        int64_t *__restrict__ Ci, uint8_t *__restrict__ Cx, int esize) {
    extern __shared__ __align__(16) unsigned char smem[];
    int64_t *key = reinterpret_cast<int64_t *>(smem);                        // SORT_MAX
    int32_t *pos = reinterpret_cast<int32_t *>(smem + (size_t)SORT_MAX * 8);  // SORT_MAX
    uint8_t *val = smem + (size_t)SORT_MAX * 12;                              // SORT_MAX * 8
    for (int64_t li = blockIdx.x; li < nlist; li += gridDim.x) {
        const int64_t j = list[li];
        const int64_t p0 = Cp[j];
        const int n = (int)(Cp[j + 1] - p0);
        int np2 = 1; while (np2 < n) np2 <<= 1;
        for (int t = threadIdx.x; t < np2; t += SORT_THREADS) {
            key[t] = (t < n) ? Ci[p0 + t] : INT64_MAX;
            pos[t] = t;
        }
        for (int t = threadIdx.x; t < n * esize; t += SORT_THREADS) val[t] = Cx[p0 * esize + t];
        __syncthreads();
        for (int ksz = 2; ksz <= np2; ksz <<= 1) {
            for (int jsz = ksz >> 1; jsz > 0; jsz >>= 1) {
                for (int t = threadIdx.x; t < np2; t += SORT_THREADS) {
                    const int u = t ^ jsz;
                    if (u > t) {
                        const bool up = ((t & ksz) == 0);
                        const int64_t a = key[t], b = key[u];
                        if ((a > b) == up) { key[t] = b; key[u] = a; const int32_t q = pos[t]; pos[t] = pos[u]; pos[u] = q; }
                    }
                }
                __syncthreads();
            }
        }
        for (int t = threadIdx.x; t < n; t += SORT_THREADS) {
            Ci[p0 + t] = key[t];
            const int src = pos[t];
            for (int b = 0; b < esize; b++) Cx[(p0 + t) * esize + b] = val[src * esize + b];
        }
        __syncthreads();
    }
}

// classify vectors: 0 already sorted / short, 1 LDS sort, 2 too long for LDS
__global__ void __launch_bounds__(256)
classify_sort_kernel(int64_t nvec, const int64_t *__restrict__ Cp, const int64_t *__restrict__ Ci, int lds_max,
                     unsigned int *__restrict__ counters, int64_t *__restrict__ lds_list, int64_t *__restrict__ big_list) {
    const int lane = threadIdx.x & 63;
    int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (; j < nvec; j += stride) {
        const int64_t p0 = Cp[j], p1 = Cp[j + 1];
        int bad = 0;
        for (int64_t p = p0 + 1 + lane; p < p1; p += 64) bad |= (Ci[p - 1] > Ci[p]);
        bad = __any(bad);
        if (bad && lane == 0) {
            if (p1 - p0 <= lds_max) { unsigned int q = atomicAdd(&counters[0], 1u); lds_list[q] = j; }
            else { unsigned int q = atomicAdd(&counters[1], 1u); big_list[q] = j; }
        }
    }
}

// Vectors longer than the LDS sort (saxpy3 emits heavy vectors panel by panel, each panel in table
// order): segmented radix sort of (index, position) pairs on compact copies, a bounded number of
// entries per batch, then the values follow their positions.
__global__ void __launch_bounds__(256)
long_len_kernel(int64_t n, const int64_t *__restrict__ list, const int64_t *__restrict__ Cp, int64_t *__restrict__ len) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q <= n) len[q] = (q < n) ? Cp[list[q] + 1] - Cp[list[q]] : 0;
}

__global__ void __launch_bounds__(256)
long_gather_kernel(int64_t q0, const int64_t *__restrict__ list, const int64_t *__restrict__ Cp,
                   const int64_t *__restrict__ cbeg, const int64_t *__restrict__ Ci,
                   int64_t *__restrict__ keys, int64_t *__restrict__ pos, int64_t *__restrict__ seg) {
    const int64_t q = q0 + blockIdx.x;
    const int64_t base = cbeg[q0], c0 = cbeg[q] - base, p0 = Cp[list[q]], n = cbeg[q + 1] - cbeg[q];
    if (threadIdx.x == 0) { seg[blockIdx.x] = c0; if (blockIdx.x == gridDim.x - 1) seg[gridDim.x] = c0 + n; }
    for (int64_t t = threadIdx.x; t < n; t += blockDim.x) { keys[c0 + t] = Ci[p0 + t]; pos[c0 + t] = p0 + t; }
}

__global__ void __launch_bounds__(256)
long_values_kernel(int64_t total, const int64_t *__restrict__ pos, const uint8_t *__restrict__ Cx, int esize,
                   uint8_t *__restrict__ vals) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; c < total; c += stride) {
        const int64_t src = pos[c];
        for (int b = 0; b < esize; b++) vals[c * esize + b] = Cx[src * esize + b];
    }
}

__global__ void __launch_bounds__(256)
long_scatter_kernel(int64_t q0, const int64_t *__restrict__ list, const int64_t *__restrict__ Cp,
                    const int64_t *__restrict__ cbeg, const int64_t *__restrict__ keys,
                    const uint8_t *__restrict__ vals, int esize, int64_t *__restrict__ Ci, uint8_t *__restrict__ Cx) {
    const int64_t q = q0 + blockIdx.x;
    const int64_t c0 = cbeg[q] - cbeg[q0], p0 = Cp[list[q]], n = cbeg[q + 1] - cbeg[q];
    for (int64_t t = threadIdx.x; t < n; t += blockDim.x) {
        Ci[p0 + t] = keys[c0 + t];
        for (int b = 0; b < esize; b++) Cx[(p0 + t) * esize + b] = vals[(c0 + t) * esize + b];
    }
}

static gba_info sort_long_vectors(gba_matrix *C, const int64_t *list, int64_t n, int esize) {
    hipStream_t st = ctx().stream;
    DevBuf cbeg;
    GBA_OK(cbeg.alloc((size_t)(n + 1) * 8));
    hipLaunchKernelGGL(long_len_kernel, dim3((unsigned)ceil_div(n + 1, 256)), dim3(256), 0, st, n, list, C->p, cbeg.as<int64_t>());
    GBA_HIP(hipGetLastError());
    GBA_OK(dev_cumsum_i64(cbeg.as<int64_t>(), n, st));
    std::vector<int64_t> hb((size_t)n + 1);
    GBA_HIP(hipMemcpyAsync(hb.data(), cbeg.p, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost, st));
    GBA_HIP(hipStreamSynchronize(st));
    const int64_t budget = (int64_t)1 << 27;            // entries per batch (the sort takes 32-bit counts)
    int end_bit = 1;
    while (end_bit < 63 && ((int64_t)1 << end_bit) < C->vlen) end_bit++;
    for (int64_t q0 = 0; q0 < n;) {
        int64_t q1 = q0 + 1;
        while (q1 < n && hb[(size_t)q1 + 1] - hb[(size_t)q0] <= budget && q1 - q0 < 65535 * 16) q1++;
        const int64_t total = hb[(size_t)q1] - hb[(size_t)q0], nseg = q1 - q0;
        if (total >= ((int64_t)1 << 31)) GBA_FAIL(GBA_NOT_IMPLEMENTED, "vector of %ld entries is too long for the device sort", (long)total);
        DevBuf keys, keys2, pos, pos2, vals, seg, tmp;
        GBA_OK(keys.alloc((size_t)total * 8)); GBA_OK(keys2.alloc((size_t)total * 8));
        GBA_OK(pos.alloc((size_t)total * 8));  GBA_OK(pos2.alloc((size_t)total * 8));
        GBA_OK(vals.alloc((size_t)total * esize));
        GBA_OK(seg.alloc((size_t)(nseg + 1) * 8));
        hipLaunchKernelGGL(long_gather_kernel, dim3((unsigned)nseg), dim3(256), 0, st, q0, list, C->p, cbeg.as<int64_t>(), C->i,
                           keys.as<int64_t>(), pos.as<int64_t>(), seg.as<int64_t>());
        GBA_HIP(hipGetLastError());
        size_t bytes = 0;
        GBA_HIP(hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, bytes, keys.as<int64_t>(), keys2.as<int64_t>(),
                                                            pos.as<int64_t>(), pos2.as<int64_t>(), (int)total, (int)nseg,
                                                            seg.as<int64_t>(), seg.as<int64_t>() + 1, 0, end_bit, st));
        GBA_OK(tmp.alloc(bytes ? bytes : 1));
        GBA_HIP(hipcub::DeviceSegmentedRadixSort::SortPairs(tmp.p, bytes, keys.as<int64_t>(), keys2.as<int64_t>(),
                                                            pos.as<int64_t>(), pos2.as<int64_t>(), (int)total, (int)nseg,
                                                            seg.as<int64_t>(), seg.as<int64_t>() + 1, 0, end_bit, st));
        int64_t vb = ceil_div(total, 256); if (vb > 16384) vb = 16384;
        hipLaunchKernelGGL(long_values_kernel, dim3((unsigned)vb), dim3(256), 0, st, total, pos2.as<int64_t>(),
                           reinterpret_cast<const uint8_t *>(C->x), esize, vals.as<uint8_t>());
        hipLaunchKernelGGL(long_scatter_kernel, dim3((unsigned)nseg), dim3(256), 0, st, q0, list, C->p, cbeg.as<int64_t>(),
                           keys2.as<int64_t>(), vals.as<uint8_t>(), esize, C->i, reinterpret_cast<uint8_t *>(C->x));
        GBA_HIP(hipGetLastError());
        GBA_HIP(hipStreamSynchronize(st));              // the batch's buffers go back to the pool
        q0 = q1;
    }
    return GBA_SUCCESS;
}

// The distinct values among idx[0..n), ascending: one radix sort of a copy and one pass that keeps the first of
// every run.  `bits` = significant bits of the values.
gba_info distinct_sorted(const int64_t *idx, int64_t n, int bits, DevBuf *out, int64_t *count) {
    hipStream_t st = ctx().stream;
    *count = 0;
    if (n >= ((int64_t)1 << 31)) GBA_FAIL(GBA_NOT_IMPLEMENTED, "%ld indices are too many for the device sort", (long)n);
    GBA_OK(out->alloc((size_t)(n > 0 ? n : 1) * 8));
    if (n == 0) return GBA_SUCCESS;
    DevBuf sorted, tmp, cnt;
    GBA_OK(sorted.alloc((size_t)n * 8));
    GBA_OK(cnt.alloc(8));
    size_t bytes = 0;
    GBA_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, idx, sorted.as<int64_t>(), (int)n, 0, bits, st));
    GBA_OK(tmp.alloc(bytes ? bytes : 1));
    GBA_HIP(hipcub::DeviceRadixSort::SortKeys(tmp.p, bytes, idx, sorted.as<int64_t>(), (int)n, 0, bits, st));
    size_t bytes2 = 0;
    GBA_HIP(hipcub::DeviceSelect::Unique(nullptr, bytes2, sorted.as<int64_t>(), out->as<int64_t>(), cnt.as<int64_t>(), (int)n, st));
    DevBuf tmp2;
    GBA_OK(tmp2.alloc(bytes2 ? bytes2 : 1));
    GBA_HIP(hipcub::DeviceSelect::Unique(tmp2.p, bytes2, sorted.as<int64_t>(), out->as<int64_t>(), cnt.as<int64_t>(), (int)n, st));
    GBA_OK(dev_read_i64(cnt.as<int64_t>(), count));           // (synchronises: the temporaries go back to the pool)
    return GBA_SUCCESS;
}

gba_info matrix_wait(gba_matrix *C) {
    GBA_OK(require_init());
    if (!is_sparse(C)) { C->jumbled = false; return GBA_SUCCESS; }      // bitmap/full: nothing pending
    if (C->mem != GBA_MEM_DEVICE) GBA_FAIL(GBA_NOT_IMPLEMENTED, "wait on a host view");
    hipStream_t st = ctx().stream;
    const int64_t nvec = C->nvec;
    const int esize = (int)esize_of(C->type);
    if (C->iso) GBA_FAIL(GBA_NOT_IMPLEMENTED, "wait on an iso matrix");
    int64_t wblocks = ceil_div(nvec * 64, 256); if (wblocks > 8192) wblocks = 8192; if (wblocks < 1) wblocks = 1;
    if (C->nzombies > 0) {
        DevBuf Np;
        GBA_OK(Np.alloc((size_t)(nvec + 1) * 8));
        GBA_HIP(hipMemsetAsync(Np.p, 0, (size_t)(nvec + 1) * 8, st));
        hipLaunchKernelGGL(count_live_kernel, dim3((unsigned)wblocks), dim3(256), 0, st, nvec, C->p, C->i, Np.as<int64_t>());
        GBA_OK(dev_cumsum_i64(Np.as<int64_t>(), nvec, st));
        int64_t nnz = 0;
        GBA_OK(dev_read_i64(Np.as<int64_t>() + nvec, &nnz));
        DevBuf Ni, Nx;
        GBA_OK(Ni.alloc((size_t)(nnz > 0 ? nnz : 1) * 8));
        GBA_OK(Nx.alloc((size_t)(nnz > 0 ? nnz : 1) * esize));
        hipLaunchKernelGGL(compact_live_kernel, dim3((unsigned)wblocks), dim3(256), 0, st, nvec, C->p, C->i,
                           reinterpret_cast<const uint8_t *>(C->x), esize, Np.as<int64_t>(), Ni.as<int64_t>(), Nx.as<uint8_t>());
        GBA_HIP(hipGetLastError());
        GBA_HIP(hipStreamSynchronize(st));
        gba_free(C->p); gba_free(C->i); gba_free(C->x);
        C->p = reinterpret_cast<int64_t *>(Np.release());
        C->i = reinterpret_cast<int64_t *>(Ni.release());
        C->x = Nx.release();
        C->nvals = nnz; C->nzombies = 0;
    }
    if (C->jumbled && C->nvals > 1) {
        DevBuf counters, lds_list, big_list;
        GBA_OK(counters.alloc(8));
        GBA_OK(lds_list.alloc((size_t)(nvec > 0 ? nvec : 1) * 8));
        GBA_OK(big_list.alloc((size_t)(nvec > 0 ? nvec : 1) * 8));
        GBA_HIP(hipMemsetAsync(counters.p, 0, 8, st));
        hipLaunchKernelGGL(classify_sort_kernel, dim3((unsigned)wblocks), dim3(256), 0, st, nvec, C->p, C->i,
                           esize <= 8 ? SORT_MAX : 0,      // the LDS sort stages values of up to 8 bytes
                           counters.as<unsigned int>(), lds_list.as<int64_t>(), big_list.as<int64_t>());
        unsigned int h[2] = {0, 0};
        GBA_HIP(hipMemcpyAsync(h, counters.p, 8, hipMemcpyDeviceToHost, st));
        GBA_HIP(hipStreamSynchronize(st));
        if (h[0] > 0) {
            const size_t lds = (size_t)SORT_MAX * (8 + 4 + 8);
            GBA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sort_vectors_lds_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            unsigned int grid = h[0] > 16384 ? 16384 : h[0];
            hipLaunchKernelGGL(sort_vectors_lds_kernel, dim3(grid), dim3(SORT_THREADS), lds, st, (int64_t)h[0],
                               lds_list.as<int64_t>(), C->p, C->i, reinterpret_cast<uint8_t *>(C->x), esize);
            GBA_HIP(hipGetLastError());
        }
        if (h[1] > 0) GBA_OK(sort_long_vectors(C, big_list.as<int64_t>(), (int64_t)h[1], esize));
        GBA_HIP(hipStreamSynchronize(st));
    }
    C->jumbled = false;
    return GBA_SUCCESS;
}

// A sorted copy of a sparse matrix, free of zombies: what the reference obtains by finishing the pending work
// of its inputs inside the call (GB_MATRIX_WAIT, Source/GB_mxm.c:105-116; GB_MATRIX_WAIT_IF_JUMBLED,
// Source/GB_AxB_dot.c:78-87).  The caller's matrix is left as it is.
__global__ void __launch_bounds__(256)
rebase_pointers_kernel(int64_t n, const int64_t *__restrict__ in, int64_t base, int64_t *__restrict__ out) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; k < n; k += stride) out[k] = in[k] - base;
}

gba_info sorted_copy(const gba_matrix *X, gba_matrix *out, std::vector<void *> &owned) {
    GBA_OK(require_init());
    if (!is_sparse(X)) GBA_FAIL(GBA_NOT_IMPLEMENTED, "sorted copy of a matrix with sparsity %d", X->sparsity);
    if (X->iso) GBA_FAIL(GBA_NOT_IMPLEMENTED, "jumbled iso-valued operand");
    hipStream_t st = ctx().stream;
    const size_t esize = esize_of(X->type);
    int64_t p0 = 0, pend = 0;
    GBA_OK(dev_read_i64(X->p, &p0));
    GBA_OK(dev_read_i64(X->p + X->nvec, &pend));
    const int64_t n = pend - p0;
    DevBuf Np, Ni, Nx;
    GBA_OK(Np.alloc((size_t)(X->nvec + 1) * 8));
    GBA_OK(Ni.alloc((size_t)(n > 0 ? n : 1) * 8));
    GBA_OK(Nx.alloc((size_t)(n > 0 ? n : 1) * (esize ? esize : 1)));
    int64_t blocks = ceil_div(X->nvec + 1, 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(rebase_pointers_kernel, dim3((unsigned)blocks), dim3(256), 0, st, X->nvec + 1, X->p, p0, Np.as<int64_t>());
    GBA_HIP(hipGetLastError());
    if (n > 0) {
        GBA_HIP(hipMemcpyAsync(Ni.p, X->i + p0, (size_t)n * 8, hipMemcpyDeviceToDevice, st));
        if (X->x) GBA_HIP(hipMemcpyAsync(Nx.p, reinterpret_cast<const uint8_t *>(X->x) + (size_t)p0 * esize, (size_t)n * esize, hipMemcpyDeviceToDevice, st));
    }
    gba_matrix T = *X;
    T.p = reinterpret_cast<int64_t *>(Np.release());
    T.i = reinterpret_cast<int64_t *>(Ni.release());
    T.x = Nx.release();
    T.h = nullptr; T.b = nullptr; T.nvals = n; T.mem = GBA_MEM_DEVICE;
    gba_info info = matrix_wait(&T);                // removes zombies (new arrays) and sorts every vector
    owned.push_back(T.p); owned.push_back(T.i); owned.push_back(T.x);
    if (info != GBA_SUCCESS) return info;
    *out = T;
    return GBA_SUCCESS;
}

// -------------------------------------------------------- bitmap/full -> sparse --
// GB_convert_bitmap_to_sparse / GB_convert_full_to_sparse (Source/GB_convert_bitmap_to_sparse.c,
// GB_convert_full_to_sparse.c): same orientation, entries in index order.  One wavefront per vector.
__global__ void __launch_bounds__(256)
dense_count_kernel(int64_t vlen, int64_t vdim, const int8_t *__restrict__ b, int64_t *__restrict__ cnt) {
    const int lane = threadIdx.x & 63;
    int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (; j < vdim; j += stride) {
        int c = 0;
        for (int64_t i = lane; i < vlen; i += 64) c += (b[j * vlen + i] != 0);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
        if (lane == 0) cnt[j] = c;
    }
}

__global__ void __launch_bounds__(256)
dense_fill_kernel(int64_t vlen, int64_t vdim, const int8_t *__restrict__ b, const uint8_t *__restrict__ x, bool iso,
                  int esize, const int64_t *__restrict__ Sp, int64_t *__restrict__ Si, uint8_t *__restrict__ Sx) {
    const int lane = threadIdx.x & 63;
    int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (; j < vdim; j += stride) {
        int64_t out = Sp[j];
        for (int64_t base = 0; base < vlen; base += 64) {
            const int64_t i = base + lane;
            const bool live = (i < vlen) && (!b || b[j * vlen + i] != 0);
            const unsigned long long m = __ballot(live);
            if (live) {
                const int64_t q = out + __builtin_popcountll(m & ((1ull << lane) - 1));
                Si[q] = i;
                if (!iso) for (int t = 0; t < esize; t++) Sx[q * esize + t] = x[(j * vlen + i) * esize + t];
            }
            out += __builtin_popcountll(m);
        }
    }
}

gba_info to_sparse(const gba_matrix *X, gba_matrix *S, std::vector<void *> &owned) {
    GBA_OK(require_init());
    if (!is_bitmap(X) && !is_full(X)) GBA_FAIL(GBA_INVALID_VALUE, "to_sparse: sparsity %d", X->sparsity);
    hipStream_t st = ctx().stream;
    const int64_t vlen = X->vlen, vdim = X->vdim;
    const int esize = (int)esize_of(X->type);
    DevBuf Sp, Si, Sx;
    GBA_OK(Sp.alloc((size_t)(vdim + 1) * 8));
    GBA_HIP(hipMemsetAsync(Sp.p, 0, (size_t)(vdim + 1) * 8, st));
    int64_t blocks = ceil_div(vdim * 64, 256); if (blocks > 8192) blocks = 8192; if (blocks < 1) blocks = 1;
    int64_t nnz = vlen * vdim;
    if (is_bitmap(X)) {
        if (vdim > 0) hipLaunchKernelGGL(dense_count_kernel, dim3((unsigned)blocks), dim3(256), 0, st, vlen, vdim, X->b, Sp.as<int64_t>());
        GBA_OK(dev_cumsum_i64(Sp.as<int64_t>(), vdim, st));
        GBA_OK(dev_read_i64(Sp.as<int64_t>() + vdim, &nnz));
    } else {
        std::vector<int64_t> hp((size_t)vdim + 1);
        for (int64_t j = 0; j <= vdim; j++) hp[(size_t)j] = j * vlen;
        GBA_HIP(hipMemcpyAsync(Sp.p, hp.data(), (size_t)(vdim + 1) * 8, hipMemcpyHostToDevice, st));
        GBA_HIP(hipStreamSynchronize(st));
    }
    GBA_OK(Si.alloc((size_t)(nnz > 0 ? nnz : 1) * 8));
    GBA_OK(Sx.alloc((size_t)((X->iso || nnz == 0) ? 1 : nnz) * (esize > 0 ? esize : 1)));
    if (X->iso && X->x && esize > 0) GBA_HIP(hipMemcpyAsync(Sx.p, X->x, (size_t)esize, hipMemcpyDeviceToDevice, st));
    if (vdim > 0 && nnz > 0) {
        hipLaunchKernelGGL(dense_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, vlen, vdim, is_bitmap(X) ? X->b : nullptr,
                           reinterpret_cast<const uint8_t *>(X->x), X->iso, esize, Sp.as<int64_t>(), Si.as<int64_t>(), Sx.as<uint8_t>());
        GBA_HIP(hipGetLastError());
    }
    GBA_HIP(hipStreamSynchronize(st));
    *S = *X;
    S->p = reinterpret_cast<int64_t *>(Sp.release()); S->i = reinterpret_cast<int64_t *>(Si.release()); S->x = Sx.release();
    S->b = nullptr; S->h = nullptr;
    S->nvec = vdim; S->nvals = nnz; S->sparsity = GBA_SPARSE; S->jumbled = false; S->nzombies = 0; S->mem = GBA_MEM_DEVICE;
    owned.push_back(S->p); owned.push_back(S->i); owned.push_back(S->x);
    return GBA_SUCCESS;
}

// ------------------------------------------------------ hypersparse -> sparse --
// A hypersparse matrix lists its non-empty vectors in h[0..nvec) with p[nvec+1]
// (Source/Shared/GB_matrix.h:306-360).  The kernels address vectors directly, so the front end gives
// them the sparse form of the same matrix (GB_convert_hyper_to_sparse, Source/GB_convert_hyper_to_sparse.c):
// a pointer array over all vdim vectors, i and x shared with the original.
__global__ void __launch_bounds__(256)
hyper_scatter_kernel(int64_t nvec, int64_t vdim, const int64_t *__restrict__ h, const int64_t *__restrict__ p, int64_t *__restrict__ cnt) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; k < nvec; k += stride) {
        const int64_t j = h[k];
        if ((unsigned long long)j < (unsigned long long)vdim) cnt[j] = p[k + 1] - p[k];      // a malformed list must not write outside
    }
}

gba_info hyper_to_sparse(const gba_matrix *X, gba_matrix *S, std::vector<void *> &owned) {
    GBA_OK(require_init());
    if (!is_hyper(X)) GBA_FAIL(GBA_INVALID_VALUE, "hyper_to_sparse: sparsity %d", X->sparsity);
    if (X->vdim > HYPER_EXPAND_MAX) GBA_FAIL(GBA_NOT_IMPLEMENTED, "hypersparse matrix with %ld vectors (limit %ld)", (long)X->vdim, (long)HYPER_EXPAND_MAX);
    hipStream_t st = ctx().stream;
    const int64_t vdim = X->vdim, nvec = X->nvec;
    DevBuf Sp;
    GBA_OK(Sp.alloc((size_t)(vdim + 1) * 8));
    GBA_HIP(hipMemsetAsync(Sp.p, 0, (size_t)(vdim + 1) * 8, st));
    if (nvec > 0) {
        if (!X->h || !X->p) GBA_FAIL(GBA_INVALID_OBJECT, "hypersparse matrix without h or p");
        int64_t blocks = ceil_div(nvec, 256); if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(hyper_scatter_kernel, dim3((unsigned)blocks), dim3(256), 0, st, nvec, vdim, X->h, X->p, Sp.as<int64_t>());
        GBA_HIP(hipGetLastError());
    }
    GBA_OK(dev_cumsum_i64(Sp.as<int64_t>(), vdim, st));
    GBA_HIP(hipStreamSynchronize(st));
    *S = *X;
    S->p = reinterpret_cast<int64_t *>(Sp.release());
    S->h = nullptr; S->nvec = vdim; S->sparsity = GBA_SPARSE;
    owned.push_back(S->p);
    return GBA_SUCCESS;
}

// ------------------------------------------------------ sparse -> hypersparse --
// The way back (Source/GB_convert_sparse_to_hyper.c): h lists the non-empty vectors, p keeps one pointer per
// listed vector; i and x stay.  C must own its arrays (its p is replaced).
__global__ void __launch_bounds__(256)
hyper_flag_kernel(int64_t nvec, const int64_t *__restrict__ p, int64_t *__restrict__ flag) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; j < nvec; j += stride) flag[j] = (p[j + 1] > p[j]) ? 1 : 0;
}

__global__ void __launch_bounds__(256)
hyper_gather_kernel(int64_t nvec, const int64_t *__restrict__ p, const int64_t *__restrict__ pos,
                    int64_t *__restrict__ h, int64_t *__restrict__ hp) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (j == 0) hp[pos[nvec]] = p[nvec];
    for (; j < nvec; j += stride)
        if (p[j + 1] > p[j]) { h[pos[j]] = j; hp[pos[j]] = p[j]; }
}

// number of non-empty vectors of a sparse or hypersparse matrix on the device
gba_info nonempty_vectors(const gba_matrix *C, int64_t *k) {
    GBA_OK(require_init());
    hipStream_t st = ctx().stream;
    const int64_t nvec = C->nvec;
    *k = 0;
    if (nvec <= 0) return GBA_SUCCESS;
    DevBuf flag;
    GBA_OK(flag.alloc((size_t)(nvec + 1) * 8));
    int64_t blocks = ceil_div(nvec, 256); if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(hyper_flag_kernel, dim3((unsigned)blocks), dim3(256), 0, st, nvec, C->p, flag.as<int64_t>());
    GBA_OK(dev_cumsum_i64(flag.as<int64_t>(), nvec, st));
    return dev_read_i64(flag.as<int64_t>() + nvec, k);
}

gba_info sparse_to_hyper(gba_matrix *C) {
    GBA_OK(require_init());
    if (!is_sparse(C) || C->mem != GBA_MEM_DEVICE) GBA_FAIL(GBA_INVALID_VALUE, "sparse_to_hyper: sparsity %d", C->sparsity);
    hipStream_t st = ctx().stream;
    const int64_t nvec = C->nvec;
    DevBuf pos;
    GBA_OK(pos.alloc((size_t)(nvec + 1) * 8));
    int64_t blocks = ceil_div(nvec, 256); if (blocks > 8192) blocks = 8192; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(hyper_flag_kernel, dim3((unsigned)blocks), dim3(256), 0, st, nvec, C->p, pos.as<int64_t>());
    GBA_OK(dev_cumsum_i64(pos.as<int64_t>(), nvec, st));
    int64_t nonempty = 0;
    GBA_OK(dev_read_i64(pos.as<int64_t>() + nvec, &nonempty));
    DevBuf h, hp;
    GBA_OK(h.alloc((size_t)(nonempty > 0 ? nonempty : 1) * 8));
    GBA_OK(hp.alloc((size_t)(nonempty + 1) * 8));
    hipLaunchKernelGGL(hyper_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, st, nvec, C->p, pos.as<int64_t>(),
                       h.as<int64_t>(), hp.as<int64_t>());
    GBA_HIP(hipGetLastError());
    GBA_HIP(hipStreamSynchronize(st));
    gba_free(C->p);
    C->p = reinterpret_cast<int64_t *>(hp.release());
    C->h = reinterpret_cast<int64_t *>(h.release());
    C->nvec = nonempty;
    C->sparsity = GBA_HYPERSPARSE;
    return GBA_SUCCESS;
}

// --------------------------------------------------------------- transpose --
__global__ void __launch_bounds__(256)
transpose_count_kernel(int64_t anz, const int64_t *__restrict__ Ai, unsigned long long *__restrict__ cnt) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; p < anz; p += stride) atomicAdd(&cnt[Ai[p]], 1ull);
}

__global__ void __launch_bounds__(256)
transpose_fill_kernel(int64_t nvec, const int64_t *__restrict__ Ap, const int64_t *__restrict__ Ai,
                      const uint8_t *__restrict__ Ax, bool A_iso, int esize, unsigned long long *__restrict__ cursor,
                      int64_t *__restrict__ Ti, uint8_t *__restrict__ Tx) {
    const int lane = threadIdx.x & 63;
    int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (; j < nvec; j += stride) {
        const int64_t p1 = Ap[j + 1];
        for (int64_t p = Ap[j] + lane; p < p1; p += 64) {
            const int64_t q = (int64_t)atomicAdd(&cursor[Ai[p]], 1ull);
            Ti[q] = j;
            const int64_t src = A_iso ? 0 : p;
            for (int b = 0; b < esize; b++) Tx[q * esize + b] = Ax[src * esize + b];
        }
    }
}

gba_info transpose(gba_matrix *AT, const gba_matrix *A) {
    GBA_OK(require_init());
    if (!is_sparse(A)) GBA_FAIL(GBA_NOT_IMPLEMENTED, "transpose: sparse matrices only");
    if (A->nzombies) GBA_FAIL(GBA_INVALID_OBJECT, "transpose: zombies");
    if (A->mem != GBA_MEM_DEVICE) GBA_FAIL(GBA_NOT_IMPLEMENTED, "transpose of a host view");
    hipStream_t st = ctx().stream;
    // a view may start in the middle of its arrays (a row slice keeps the absolute offsets in p): its entries
    // are i[p[0] .. p[nvec])
    int64_t a0 = 0, aend = 0;
    GBA_OK(dev_read_i64(A->p, &a0));
    GBA_OK(dev_read_i64(A->p + A->nvec, &aend));
    const int64_t anz = aend - a0, tvec = A->vlen, avec = A->nvec;
    const int esize = (int)esize_of(A->type);
    DevBuf Tp, Ti, Tx, cur;
    GBA_OK(Tp.alloc((size_t)(tvec + 1) * 8));
    GBA_OK(cur.alloc((size_t)(tvec + 1) * 8));
    GBA_OK(Ti.alloc((size_t)(anz > 0 ? anz : 1) * 8));
    GBA_OK(Tx.alloc((size_t)(anz > 0 ? anz : 1) * (esize > 0 ? esize : 1)));
    GBA_HIP(hipMemsetAsync(Tp.p, 0, (size_t)(tvec + 1) * 8, st));
    if (anz > 0) {
        int64_t blocks = ceil_div(anz, 256); if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(transpose_count_kernel, dim3((unsigned)blocks), dim3(256), 0, st, anz, A->i + a0, Tp.as<unsigned long long>());
    }
    GBA_OK(dev_cumsum_i64(Tp.as<int64_t>(), tvec, st));
    GBA_HIP(hipMemcpyAsync(cur.p, Tp.p, (size_t)(tvec + 1) * 8, hipMemcpyDeviceToDevice, st));
    if (anz > 0) {
        int64_t blocks = ceil_div(avec * 64, 256); if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(transpose_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, avec, A->p, A->i,
                           reinterpret_cast<const uint8_t *>(A->x), A->iso, esize, cur.as<unsigned long long>(),
                           Ti.as<int64_t>(), Tx.as<uint8_t>());
    }
    GBA_HIP(hipGetLastError());
    GBA_HIP(hipStreamSynchronize(st));
    gba_matrix T = *A;
    T.vlen = A->vdim; T.vdim = A->vlen; T.nvec = tvec; T.nvals = anz;
    T.p = reinterpret_cast<int64_t *>(Tp.release());
    T.i = reinterpret_cast<int64_t *>(Ti.release());
    T.x = Tx.release();
    T.h = nullptr; T.b = nullptr; T.iso = false; T.nzombies = 0; T.mem = GBA_MEM_DEVICE;
    T.jumbled = true;
    T.is_csc = A->is_csc;
    gba_info info = matrix_wait(&T);        // bucket order -> sorted vectors
    if (info != GBA_SUCCESS) { gba_matrix_free_content(&T); return info; }
    *AT = T;
    return GBA_SUCCESS;
}

}  // namespace gba

using namespace gba;

extern "C" gba_info gba_partition_by_flops(int64_t *part, int32_t nparts, const gba_matrix *A, const gba_matrix *B) {
    GBA_OK(require_init());
    if (!part || !A || !B) { set_error("NULL argument"); return GBA_NULL_POINTER; }
    if (nparts < 1) { set_error("nparts < 1"); return GBA_INVALID_VALUE; }
    // vectors of the "B" operand are the output vectors (for row-major matrices: the rows of the
    // user's left operand).  Same flop count as saxpy3, then an even split of the prefix sum.
    const int64_t nvec = B->nvec;
    DevBuf fl; GBA_OK(fl.alloc((size_t)(nvec + 1) * 8));
    GBA_OK(saxpy3_flopcount(nullptr, fl.as<int64_t>(), nullptr, false, A, B));
    GBA_OK(dev_cumsum_i64(fl.as<int64_t>(), nvec, ctx().stream));
    std::vector<int64_t> h((size_t)nvec + 1);
    GBA_HIP(hipMemcpy(h.data(), fl.p, (size_t)(nvec + 1) * 8, hipMemcpyDeviceToHost));
    const double total = (double)h[(size_t)nvec];
    part[0] = 0;
    for (int r = 1; r < nparts; r++) {
        const int64_t target = (int64_t)(total * (double)r / (double)nparts);
        int64_t k = std::lower_bound(h.begin(), h.end(), target) - h.begin();
        if (k > nvec) k = nvec;
        if (k < part[r - 1]) k = part[r - 1];
        part[r] = k;
    }
    part[nparts] = nvec;
    return GBA_SUCCESS;
}
