// C<#M> = A*B with C sparse: the device counterpart of GB_AxB_saxpy3
// (Source/GB_AxB_saxpy3.c:120) and of the phases it drives:
//   flop count   GB_AxB_saxpy3_flopcount   (Source/GB_AxB_saxpy3_flopcount.c:81)
//   tasks        GB_AxB_saxpy3_slice_balanced (Source/GB_AxB_saxpy3_slice_balanced.c:192)
//   symbolic     GB_AxB_saxpy3_symbolic    (Source/Factories/GB_AxB_saxpy3_symbolic_template.c:30)
//   cumsum       GB_AxB_saxpy3_cumsum      (Source/GB_AxB_saxpy3_cumsum.c:18)
//   numeric      Source/Template/GB_AxB_saxpy3_template.c
//
// The CPU builds coarse tasks (ranges of output vectors) and fine tasks (teams
// sharing one vector) for a few dozen threads.  Here every output vector is
// its own unit of work and is routed by its flop count to one of:
//   wave bin    one 64-lane wavefront, open-addressing hash table in LDS
//   block bins  one workgroup, larger LDS hash table
//   global bin  persistent workgroups, dense Gustavson accumulator in HBM with
//               a two-level bitmap of touched slots (the CPU's Hf/Hx arrays,
//               Source/Template/GB_AxB_saxpy3_coarseGus_noM_phase5.c:92-117)
// Hash function and probing are those of Source/Shared/GB_hash.h:20-31.
// All notation is the reference's: vectors are "columns" j of B and C, k runs
// over B(:,j), i over A(:,k); for row-major user matrices the caller has
// already swapped the operands (Source/GB_AxB_meta.c:428-439).
#include "gba_internal.hpp"

namespace gba {

// ------------------------------------------------------------------- views --
struct SpView {                 // sparse operand
    const int64_t *p; const int64_t *i; const void *x; bool iso;
    const int32_t *i32;         // the same indices in 32 bits (panel kernels only; vlen < 2^31), or null
};
struct MaskView {
    int kind;                   // 0 none, 1 sparse, 2 bitmap, 3 full
    bool comp, structural, iso;
    int msize;                  // bytes per mask value
    int64_t vlen;
    const int64_t *p; const int64_t *i; const int8_t *b; const uint8_t *x;
};

__device__ inline bool mask_value(const MaskView &m, int64_t p) {
    if (m.structural) return true;
    const uint8_t *v = m.x + (m.iso ? 0 : p * m.msize);
    switch (m.msize) {
        case 1: return v[0] != 0;
        case 2: return *reinterpret_cast<const uint16_t *>(v) != 0;
        case 4: return *reinterpret_cast<const uint32_t *>(v) != 0;
        case 8: return *reinterpret_cast<const uint64_t *>(v) != 0;
        default: { bool nz = false; for (int q = 0; q < m.msize; q++) nz |= (v[q] != 0); return nz; }
    }
}

// may C(i,j) be computed?  (GB_CHECK_MASK_ij, Source/Template/GB_AxB_saxpy3_template.h)
__device__ inline bool mask_allows(const MaskView &m, int64_t i, int64_t j) {
    if (m.kind == 0) return true;
    bool mij;
    if (m.kind == 1) {
        int64_t lo = m.p[j], hi = m.p[j + 1] - 1;
        mij = false;
        while (lo <= hi) {
            int64_t mid = (lo + hi) >> 1;
            int64_t v = m.i[mid];
            if (v < i) lo = mid + 1; else if (v > i) hi = mid - 1; else { mij = mask_value(m, mid); break; }
        }
    } else {
        const int64_t p = j * m.vlen + i;
        mij = (m.kind == 2) ? (m.b[p] != 0) : true;
        if (mij) mij = mask_value(m, p);
    }
    return mij != m.comp;
}

// --------------------------------------------------------------- flopcount --
// one wavefront per vector of B; integer-only path (no mask, or complemented)
__global__ void __launch_bounds__(256)
flopcount_wave_kernel(int64_t nvec, const int64_t *__restrict__ Bp, const int64_t *__restrict__ Bi,
                      const int64_t *__restrict__ Ap, const int64_t *__restrict__ Mp,
                      int64_t *__restrict__ vflops, unsigned long long *__restrict__ Mwork) {
    const int lane = threadIdx.x & 63;
    const int64_t kk = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (kk >= nvec) return;
    const int64_t pB0 = Bp[kk], pB1 = Bp[kk + 1];
    const int64_t bjnz = pB1 - pB0;
    if (bjnz == 0) { if (lane == 0) vflops[kk] = 0; return; }
    int64_t base = bjnz;
    if (Mp) {                                   // sparse complemented mask
        int64_t mjnz = Mp[kk + 1] - Mp[kk];
        if (mjnz > 0) { base = mjnz; if (lane == 0) atomicAdd(Mwork, (unsigned long long)mjnz); }
    }
    int64_t s = 0;
    for (int64_t pB = pB0 + lane; pB < pB1; pB += 64) {
        const int64_t k = Bi[pB];
        s += Ap[k + 1] - Ap[k];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
    if (lane == 0) vflops[kk] = base + s;
}

// few, long vectors (a frontier step has one): several workgroups share a vector; vflops must start at zero
__global__ void __launch_bounds__(256)
flopcount_shared_kernel(int64_t nvec, const int64_t *__restrict__ Bp, const int64_t *__restrict__ Bi,
                        const int64_t *__restrict__ Ap, const int64_t *__restrict__ Mp,
                        int64_t *__restrict__ vflops, unsigned long long *__restrict__ Mwork) {
    const int64_t kk = blockIdx.y;
    const int64_t pB0 = Bp[kk], pB1 = Bp[kk + 1];
    const int64_t bjnz = pB1 - pB0;
    if (bjnz == 0) return;
    int64_t s = 0;
    for (int64_t pB = pB0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pB < pB1; pB += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = Bi[pB];
        s += Ap[k + 1] - Ap[k];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int64_t base = bjnz;
        if (Mp) {
            const int64_t mjnz = Mp[kk + 1] - Mp[kk];
            if (mjnz > 0) { base = mjnz; atomicAdd(Mwork, (unsigned long long)mjnz); }
        }
        s += base;
    }
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(reinterpret_cast<unsigned long long *>(&vflops[kk]), (unsigned long long)s);
}

// one thread per vector; mirrors the sequential arithmetic of
// Source/GB_AxB_saxpy3_flopcount.c:239-383 including the (int64 += double) steps
__global__ void __launch_bounds__(256)
flopcount_masked_kernel(int64_t nvec, const int64_t *__restrict__ Bp, const int64_t *__restrict__ Bi,
                        const int64_t *__restrict__ Ap, int64_t avlen, bool A_jumbled,
                        const int64_t *__restrict__ Mp, bool M_is_dense, bool Mask_comp, int64_t mvlen,
                        int64_t *__restrict__ vflops, unsigned long long *__restrict__ Mwork) {
    const int64_t kk = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (kk >= nvec) return;
    const int64_t pB0 = Bp[kk], pB1 = Bp[kk + 1];
    const int64_t bjnz = pB1 - pB0;
    if (bjnz == 0) { vflops[kk] = 0; return; }
    int64_t bjflops = bjnz;
    int64_t mjnz = 0;
    if (!M_is_dense) {
        mjnz = Mp[kk + 1] - Mp[kk];
        if (mjnz == 0 && !Mask_comp) { vflops[kk] = 0; return; }
        if (mjnz > 0) { bjflops = mjnz; atomicAdd(Mwork, (unsigned long long)mjnz); }
    }
    const int64_t mjnz_much = 64 * mjnz;
    const bool mask_is_M = !Mask_comp;
    for (int64_t pB = pB0; pB < pB1; pB++) {
        const int64_t k = Bi[pB];
        const int64_t aknz = Ap[k + 1] - Ap[k];
        if (aknz == 0) continue;
        double bkjflops;
        if (mask_is_M && aknz > 256 && mjnz_much < aknz && mjnz < mvlen && aknz < avlen && !A_jumbled)
            bkjflops = (double)mjnz * (1 + 4 * log2((double)aknz));
        else
            bkjflops = (double)aknz;
        bjflops = (int64_t)((double)bjflops + bkjflops);
    }
    vflops[kk] = bjflops;
}

gba_info saxpy3_flopcount(int64_t *Mwork_out, int64_t *Bflops, const gba_matrix *M, bool Mask_comp,
                          const gba_matrix *A, const gba_matrix *B) {
    GBA_OK(require_init());
    if (!is_sparse(A) || !is_sparse(B)) GBA_FAIL(GBA_NOT_IMPLEMENTED, "flopcount: A and B must be sparse");
    if (M && is_hyper(M)) GBA_FAIL(GBA_NOT_IMPLEMENTED, "flopcount: hypersparse mask");
    hipStream_t st = ctx().stream;
    const int64_t nvec = B->nvec;
    DevBuf mw; GBA_OK(mw.alloc(8));
    GBA_HIP(hipMemsetAsync(mw.p, 0, 8, st));
    GBA_HIP(hipMemsetAsync(Bflops + nvec, 0, 8, st));
    if (nvec > 0) {
        const bool M_is_dense = M && (is_bitmap(M) || as_if_full(M));
        if (M && !Mask_comp) {
            hipLaunchKernelGGL(flopcount_masked_kernel, dim3((unsigned)ceil_div(nvec, 256)), dim3(256), 0, st,
                               nvec, B->p, B->i, A->p, A->vlen, A->jumbled, M->p, M_is_dense, Mask_comp, M->vlen,
                               Bflops, mw.as<unsigned long long>());
        } else {
            const int64_t *Mp = (M && !M_is_dense) ? M->p : nullptr;
            const int64_t avg = B->nvals / nvec;
            if (nvec <= 1024 && avg > 4096) {
                int64_t per = ceil_div(avg, 2048); if (per > 256) per = 256;
                GBA_HIP(hipMemsetAsync(Bflops, 0, (size_t)nvec * 8, st));
                hipLaunchKernelGGL(flopcount_shared_kernel, dim3((unsigned)per, (unsigned)nvec), dim3(256), 0, st,
                                   nvec, B->p, B->i, A->p, Mp, Bflops, mw.as<unsigned long long>());
            } else {
                hipLaunchKernelGGL(flopcount_wave_kernel, dim3((unsigned)ceil_div(nvec * 64, 256)), dim3(256), 0, st,
                                   nvec, B->p, B->i, A->p, Mp, Bflops, mw.as<unsigned long long>());
            }
        }
        GBA_HIP(hipGetLastError());
    }
    if (Mwork_out) {
        int64_t v = 0;
        GBA_OK(dev_read_i64(mw.as<int64_t>(), &v));
        *Mwork_out = v;
    } else {
        GBA_HIP(hipStreamSynchronize(st));
    }
    return GBA_SUCCESS;
}

// ------------------------------------------------------------------ binning --
// Symbolic bins by flop count f (an upper bound on nnz C(:,j)); numeric bins by
// the exact count c.  Table sizes keep the load factor <= 1/2 like the CPU's
// hash_size = 2 * 2^ceil(log2 flmax) (Source/GB_AxB_saxpy3_slice_balanced.c:83).
enum { BIN_EMPTY = 0, BIN_WAVE = 1, BIN_BLK1 = 2, BIN_BLK2 = 3, BIN_GLOBAL = 4, NBINS = 5 };

static constexpr int SYM_WAVE_SLOTS = 1024,  SYM_WAVE_MAX = 512;
static constexpr int SYM_BLK1_SLOTS = 8192,  SYM_BLK1_MAX = 4096,  SYM_BLK1_THREADS = 256;
static constexpr int SYM_BLK2_SLOTS = 32768, SYM_BLK2_MAX = 16384, SYM_BLK2_THREADS = 1024;
static constexpr int NUM_WAVE_SLOTS = 512,   NUM_WAVE_MAX = 256;
static constexpr int NUM_BLK1_SLOTS = 4096,  NUM_BLK1_MAX = 2048,  NUM_BLK1_THREADS = 256;
static constexpr int NUM_BLK2_SLOTS = 8192,  NUM_BLK2_MAX = 4096,  NUM_BLK2_THREADS = 512;
static constexpr int GLOBAL_THREADS = 512;
#ifndef GBA_NUM_PANEL_MIN
#define GBA_NUM_PANEL_MIN 1024           // with the panel kernels at hand: entries of a vector above which the numeric panel kernel takes it
#endif
#ifndef GBA_SYM_PANEL_MIN
#define GBA_SYM_PANEL_MIN 1024
#endif
static constexpr int SYM_PANEL_MIN = GBA_SYM_PANEL_MIN;         // with sorted A the symbolic panel kernel takes every vector of more multiply-adds
// one-pass road for a few light vectors (frontier steps): at most SMALL_NVEC vectors of at most SMALL_ROOM
// multiply-adds each, accumulated in the second block bin's table (three quarters full at worst)
static constexpr int SMALL_NVEC = 256, SMALL_ROOM = 6144;
static constexpr int64_t SMALL_BNZ = 16384;
static_assert(SMALL_ROOM * 4 <= NUM_BLK2_SLOTS * 3, "the one-pass table never fills up");

struct BinLimits { int64_t wave_max, blk1_max, blk2_max; };

__device__ inline int bin_of(int64_t w, const BinLimits &L) {
    if (w <= 0) return BIN_EMPTY;
    if (w <= L.wave_max) return BIN_WAVE;
    if (w <= L.blk1_max) return BIN_BLK1;
    if (w <= L.blk2_max) return BIN_BLK2;
    return BIN_GLOBAL;
}

// bin by work; a second measure (the flop count, when binning by exact entry count) can only move a
// vector to a bin with more threads: a masked vector may have few entries and still many products
__device__ inline int bin_of2(int64_t w, const BinLimits &L, const int64_t *__restrict__ work2, int64_t idx, const BinLimits &L2) {
    int b = bin_of(w, L);
    if (b != BIN_EMPTY && work2) { const int b2 = bin_of(work2[idx], L2); if (b2 > b) b = b2; }
    return b;
}

__global__ void __launch_bounds__(256)
bin_count_kernel(int64_t nvec, const int64_t *__restrict__ work, bool work_is_pointer_diff, BinLimits L,
                 const int64_t *__restrict__ work2, BinLimits L2, unsigned long long *__restrict__ counts) {
    __shared__ unsigned int lc[NBINS];
    if (threadIdx.x < NBINS) lc[threadIdx.x] = 0;
    __syncthreads();
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; idx < nvec; idx += stride) {
        const int64_t w = work_is_pointer_diff ? (work[idx + 1] - work[idx]) : work[idx];
        atomicAdd(&lc[bin_of2(w, L, work2, idx, L2)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < NBINS && lc[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)lc[threadIdx.x]);
}

__global__ void __launch_bounds__(256)
bin_fill_kernel(int64_t nvec, const int64_t *__restrict__ work, bool work_is_pointer_diff, BinLimits L,
                const int64_t *__restrict__ work2, BinLimits L2,
                unsigned long long *__restrict__ cursors, int64_t *__restrict__ lists) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; idx < nvec; idx += stride) {
        const int64_t w = work_is_pointer_diff ? (work[idx + 1] - work[idx]) : work[idx];
        const int b = bin_of2(w, L, work2, idx, L2);
        // one atomic per wave and bin (atomics on one address are carried out one after the other, about 12 ns
        // each: a cursor bumped once per vector costs milliseconds at 10^6 vectors)
        for (int bb = 0; bb < NBINS; bb++) {
            if (bb == BIN_EMPTY) continue;
            const unsigned long long mine = __ballot(b == bb);
            if (!mine) continue;                               // uniform
            const int lane = threadIdx.x & 63;
            unsigned long long base = 0;
            if (lane == __builtin_ctzll(mine)) base = atomicAdd(&cursors[bb], (unsigned long long)__builtin_popcountll(mine));
            base = __shfl(base, __builtin_ctzll(mine), 64);
            if (b == bb) lists[base + __builtin_popcountll(mine & ((1ull << lane) - 1))] = idx;
        }
    }
}

struct Bins {
    DevBuf lists;                 // vector ids grouped by bin
    DevBuf cnt;                   // per-bin counts, then the fill cursors (device)
    int64_t count[NBINS] = {0, 0, 0, 0, 0};
    int64_t start[NBINS] = {0, 0, 0, 0, 0};
    const int64_t *list(int b) const { return lists.as<int64_t>() + start[b]; }
};

// cursors[b] = where bin b starts in the grouped list (the empty bin takes no room)
__global__ void bin_offsets_kernel(const unsigned long long *__restrict__ counts, unsigned long long *__restrict__ cursors) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        unsigned long long off = 0;
        for (int b = 0; b < NBINS; b++) { cursors[b] = off; if (b != BIN_EMPTY) off += counts[b]; }
    }
}

static gba_info build_bins(int64_t nvec, const int64_t *work, bool ptr_diff, BinLimits L, Bins *bins,
                           const int64_t *work2 = nullptr, BinLimits L2 = BinLimits{0, 0, 0}) {
    hipStream_t st = ctx().stream;
    GBA_OK(bins->cnt.alloc(2 * NBINS * 8));
    unsigned long long *counts = bins->cnt.as<unsigned long long>(), *cursors = counts + NBINS;
    GBA_HIP(hipMemsetAsync(counts, 0, NBINS * 8, st));
    GBA_OK(bins->lists.alloc((size_t)(nvec > 0 ? nvec : 1) * 8));
    if (nvec == 0) return GBA_SUCCESS;
    int64_t blocks = ceil_div(nvec, 256); if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(bin_count_kernel, dim3((unsigned)blocks), dim3(256), 0, st, nvec, work, ptr_diff, L, work2, L2, counts);
    hipLaunchKernelGGL(bin_offsets_kernel, dim3(1), dim3(64), 0, st, counts, cursors);
    hipLaunchKernelGGL(bin_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, nvec, work, ptr_diff, L, work2, L2,
                       cursors, bins->lists.as<int64_t>());
    GBA_HIP(hipGetLastError());
    unsigned long long h[NBINS];
    GBA_HIP(hipMemcpyAsync(h, counts, NBINS * 8, hipMemcpyDeviceToHost, st));
    GBA_HIP(hipStreamSynchronize(st));                     // the one round trip: the host sizes its launches by the counts
    int64_t off = 0;
    for (int b = 0; b < NBINS; b++) {
        bins->count[b] = (int64_t)h[b];
        bins->start[b] = off;
        if (b != BIN_EMPTY) off += (int64_t)h[b];
    }
    return GBA_SUCCESS;
}

// totals of nnz C(:,j), nnz B(:,j) and multiply-adds over one bin (reporting only)
__global__ void __launch_bounds__(256)
bin_stats_kernel(int64_t nlist, const int64_t *__restrict__ list, const int64_t *__restrict__ Cp,
                 const int64_t *__restrict__ Bp, const int64_t *__restrict__ vflops,
                 unsigned long long *__restrict__ out /* [3] */) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long c = 0, b = 0, f = 0;
    for (; idx < nlist; idx += stride) {
        const int64_t j = list[idx];
        const unsigned long long bj = (unsigned long long)(Bp[j + 1] - Bp[j]);
        c += (unsigned long long)(Cp[j + 1] - Cp[j]);
        b += bj;
        f += (unsigned long long)vflops[j] - bj;        // the flop count includes nnz B(:,j) itself
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { c += __shfl_down(c, d, 64); b += __shfl_down(b, d, 64); f += __shfl_down(f, d, 64); }
    if ((threadIdx.x & 63) == 0) { if (c) atomicAdd(&out[0], c); if (b) atomicAdd(&out[1], b); if (f) atomicAdd(&out[2], f); }
}

// Persistent workgroups pull vectors from a list; with the heaviest first (longest processing time first)
// the last vectors to finish are light ones.  Counting sort of the list by floor(log2(work)), descending.
__global__ void __launch_bounds__(256)
work_hist_kernel(int64_t n, const int64_t *__restrict__ list, const int64_t *__restrict__ work,
                 unsigned long long *__restrict__ hist /* [64] */) {
    __shared__ unsigned int lh[64];
    if (threadIdx.x < 64) lh[threadIdx.x] = 0;
    __syncthreads();
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; idx < n; idx += stride) {
        const int64_t w = work[list[idx]];
        atomicAdd(&lh[w > 0 ? 63 - __builtin_clzll((unsigned long long)w) : 0], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64 && lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)lh[threadIdx.x]);
}

__global__ void __launch_bounds__(256)
work_scatter_kernel(int64_t n, const int64_t *__restrict__ list, const int64_t *__restrict__ work,
                    unsigned long long *__restrict__ cursor /* [64] */, int64_t *__restrict__ out) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; idx < n; idx += stride) {
        const int64_t j = list[idx], w = work[j];
        const int b = w > 0 ? 63 - __builtin_clzll((unsigned long long)w) : 0;
        // one atomic per wave and class present in it
        const int lane = threadIdx.x & 63;
        bool placed = false;
        while (true) {
            const unsigned long long todo = __ballot(!placed);
            if (!todo) break;                                  // uniform
            const int leader = __builtin_ctzll(todo);
            const int lb = __shfl(b, leader, 64);
            const unsigned long long mine = __ballot(!placed && b == lb);
            unsigned long long base = 0;
            if (lane == leader) base = atomicAdd(&cursor[lb], (unsigned long long)__builtin_popcountll(mine));
            base = __shfl(base, leader, 64);
            if (!placed && b == lb) { out[base + __builtin_popcountll(mine & ((1ull << lane) - 1))] = j; placed = true; }
        }
    }
}

// cursor[b] = number of list entries in heavier classes (classes are filled from 63 down)
__global__ void work_offsets_kernel(const unsigned long long *__restrict__ hist, unsigned long long *__restrict__ cursor) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        unsigned long long off = 0;
        for (int b = 63; b >= 0; b--) { cursor[b] = off; off += hist[b]; }
    }
}

// out: the ordered list (n ids), followed by the histogram and the cursors; nothing comes back to the host
static gba_info order_by_work_desc(int64_t n, const int64_t *list, const int64_t *work, DevBuf *out) {
    hipStream_t st = ctx().stream;
    const int64_t n1 = n > 0 ? n : 1;
    GBA_OK(out->alloc((size_t)(n1 + 128) * 8));
    if (n == 0) return GBA_SUCCESS;
    unsigned long long *hist = out->as<unsigned long long>() + n1, *cursor = hist + 64;
    GBA_HIP(hipMemsetAsync(hist, 0, 64 * 8, st));
    int64_t blocks = ceil_div(n, 256); if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(work_hist_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n, list, work, hist);
    hipLaunchKernelGGL(work_offsets_kernel, dim3(1), dim3(64), 0, st, hist, cursor);
    hipLaunchKernelGGL(work_scatter_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n, list, work, cursor, out->as<int64_t>());
    GBA_HIP(hipGetLastError());
    return GBA_SUCCESS;
}

// ------------------------------------------------------------- LDS hashing --
static constexpr int32_t HASH_EMPTY = -1;

// GB_HASHF (Source/Shared/GB_hash.h:20): (i*257) & mask ; GB_REHASH: linear probe
__device__ inline uint32_t hashf(int32_t i, uint32_t mask) { return ((uint32_t)i * 257u) & mask; }

// returns the slot of key; *is_new tells whether this call created it
__device__ inline uint32_t hash_insert(int32_t *keys, uint32_t mask, int32_t key, bool *is_new) {
    uint32_t h = hashf(key, mask);
    *is_new = false;
    while (true) {
        int32_t cur = keys[h];
        if (cur == key) return h;
        if (cur == HASH_EMPTY) {
            int32_t old = atomicCAS(&keys[h], HASH_EMPTY, key);
            if (old == HASH_EMPTY) { *is_new = true; return h; }
            if (old == key) return h;
        }
        h = (h + 1) & mask;
    }
}

// Walk every product A(i,k)*B(k,j) of vector j with `nthreads` cooperating
// threads, `lpa` consecutive threads sharing one A(:,k).  f(i, k, pA, pB).
template <class F>
__device__ inline void for_each_product(const SpView &A, const SpView &B, int64_t j, int tid, int nthreads, int lpa, F f) {
    const int sub = tid / lpa, lane = tid - sub * lpa, nsub = nthreads / lpa;
    const int64_t pB1 = B.p[j + 1];
    for (int64_t pB = B.p[j] + sub; pB < pB1; pB += nsub) {
        const int64_t k = B.i[pB];
        const int64_t pA1 = A.p[k + 1];
        for (int64_t pA = A.p[k] + lane; pA < pA1; pA += lpa) f(A.i[pA], k, pA, pB);
    }
}

// Workgroup version.  A(:,k) longer than 8*lpa entries (a hub) would leave one
// lane group crawling while the rest of the workgroup idles, so such vectors are
// queued in LDS and then walked by all THREADS together, coalesced.
static constexpr int LONGQ_CAP = 2048;
template <int THREADS, class F>
__device__ inline void for_each_product_block(const SpView &A, const SpView &B, int64_t j, int lpa,
                                              int *queue /* LDS [LONGQ_CAP] */, int *qn /* LDS */, F f) {
    const int tid = threadIdx.x;
    const int sub = tid / lpa, lane = tid - sub * lpa, nsub = THREADS / lpa;
    const int64_t long_len = (int64_t)lpa * 8;
    const int64_t pB0 = B.p[j], pB1 = B.p[j + 1];
    if (tid == 0) *qn = 0;
    __syncthreads();
    for (int64_t pB = pB0 + sub; pB < pB1; pB += nsub) {
        const int64_t k = B.i[pB];
        const int64_t pA0 = A.p[k], pA1 = A.p[k + 1];
        bool defer = false;
        if (pA1 - pA0 > long_len) {
            int q = 0;
            if (lane == 0) q = atomicAdd(qn, 1);
            q = __shfl(q, 0, lpa);
            if (q < LONGQ_CAP) { if (lane == 0) queue[q] = (int)(pB - pB0); defer = true; }
        }
        if (!defer) for (int64_t pA = pA0 + lane; pA < pA1; pA += lpa) f(A.i[pA], k, pA, pB);
    }
    __syncthreads();
    const int nq = (*qn < LONGQ_CAP) ? *qn : LONGQ_CAP;
    for (int q = 0; q < nq; q++) {
        const int64_t pB = pB0 + queue[q];
        const int64_t k = B.i[pB];
        const int64_t pA1 = A.p[k + 1];
        for (int64_t pA = A.p[k] + tid; pA < pA1; pA += THREADS) f(A.i[pA], k, pA, pB);
    }
    __syncthreads();
}

// ---------------------------------------------------------------- symbolic --
// wave bin: 4 waves per workgroup, one vector per wave
__global__ void __launch_bounds__(256)
sym_wave_kernel(int64_t nlist, const int64_t *__restrict__ list, SpView A, SpView B, MaskView M, int lpa,
                int64_t *__restrict__ cnt) {
    __shared__ int32_t keys_all[4 * SYM_WAVE_SLOTS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t li = (int64_t)blockIdx.x * 4 + wave;
    int32_t *keys = keys_all + wave * SYM_WAVE_SLOTS;
    for (int s = lane; s < SYM_WAVE_SLOTS; s += 64) keys[s] = HASH_EMPTY;
    __builtin_amdgcn_wave_barrier();
    if (li >= nlist) return;
    const int64_t j = list[li];
    int c = 0;
    for_each_product(A, B, j, lane, 64, lpa, [&](int64_t i, int64_t, int64_t, int64_t) {
        if (!mask_allows(M, i, j)) return;
        bool is_new;
        hash_insert(keys, SYM_WAVE_SLOTS - 1, (int32_t)i, &is_new);
        c += is_new;
    });
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
    if (lane == 0) cnt[j] = c;
}

template <int SLOTS, int THREADS>
__global__ void __launch_bounds__(THREADS)
sym_block_kernel(int64_t nlist, const int64_t *__restrict__ list, SpView A, SpView B, MaskView M, int lpa,
                 int64_t *__restrict__ cnt) {
    extern __shared__ __align__(16) unsigned char smem[];
    int32_t *keys = reinterpret_cast<int32_t *>(smem);
    __shared__ int total;
    __shared__ int longq[LONGQ_CAP];
    __shared__ int longn;
    const int64_t j = list[blockIdx.x];
    for (int s = threadIdx.x; s < SLOTS; s += THREADS) keys[s] = HASH_EMPTY;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    int c = 0;
    for_each_product_block<THREADS>(A, B, j, lpa, longq, &longn, [&](int64_t i, int64_t, int64_t, int64_t) {
        if (!mask_allows(M, i, j)) return;
        bool is_new;
        hash_insert(keys, SLOTS - 1, (int32_t)i, &is_new);
        c += is_new;
    });
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&total, c);
    __syncthreads();
    if (threadIdx.x == 0) cnt[j] = total;
}

// Global bin: persistent workgroups pull vectors from a counter.  Touched
// slots are marked in a bitmap Hf (1 bit per index) summarised by Hs (1 bit
// per Hf word).  Both are private to the workgroup and self-cleaning.
__device__ inline uint64_t ld_agent(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ inline void mark_slot(uint64_t *Hf, uint64_t *Hs, int64_t i) {
    const int64_t w = i >> 6;
    const uint64_t bit = 1ull << (i & 63);
    const uint64_t old = atomicOr(reinterpret_cast<unsigned long long *>(&Hf[w]), (unsigned long long)bit);
    if (old == 0) atomicOr(reinterpret_cast<unsigned long long *>(&Hs[w >> 6]), 1ull << (w & 63));
}

__device__ inline int block_sum_int(int v, int *scratch /* >= 16 ints */, int nthreads) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    int t = 0;
    for (int w = 0; w < (nthreads >> 6); w++) t += scratch[w];
    __syncthreads();
    return t;
}

__global__ void __launch_bounds__(GLOBAL_THREADS)
sym_global_kernel(int64_t nlist, const int64_t *__restrict__ list, SpView A, SpView B, MaskView M, int lpa,
                  int64_t cvlen, uint64_t *__restrict__ Hf_all, uint64_t *__restrict__ Hs_all,
                  unsigned long long *__restrict__ next, int64_t *__restrict__ cnt) {
    __shared__ int64_t cur;
    __shared__ int scratch[16];
    __shared__ int longq[LONGQ_CAP];
    __shared__ int longn;
    const int64_t nf = (cvlen + 63) >> 6, ns = (nf + 63) >> 6;
    uint64_t *Hf = Hf_all + (int64_t)blockIdx.x * nf;
    uint64_t *Hs = Hs_all + (int64_t)blockIdx.x * ns;
    while (true) {
        if (threadIdx.x == 0) cur = (int64_t)atomicAdd(next, 1ull);
        __syncthreads();
        const int64_t li = cur;
        if (li >= nlist) break;                 // uniform: every wave leaves here
        const int64_t j = list[li];
        for_each_product_block<GLOBAL_THREADS>(A, B, j, lpa, longq, &longn, [&](int64_t i, int64_t, int64_t, int64_t) {
            if (!mask_allows(M, i, j)) return;
            mark_slot(Hf, Hs, i);
        });
        __syncthreads();
        int c = 0;
        for (int64_t s = threadIdx.x; s < ns; s += GLOBAL_THREADS) {
            uint64_t sw = ld_agent(&Hs[s]);
            if (!sw) continue;
            while (sw) {
                const int b = __builtin_ctzll(sw);
                sw &= sw - 1;
                const int64_t w = (s << 6) + b;
                c += __builtin_popcountll(ld_agent(&Hf[w]));
                Hf[w] = 0;
            }
            Hs[s] = 0;
        }
        const int total = block_sum_int(c, scratch, GLOBAL_THREADS);
        if (threadIdx.x == 0) cnt[j] = total;
        __syncthreads();
    }
}

// ----------------------------------------------------------------- numeric --
template <class S>
__device__ inline typename S::z_t product(const S &sr, const SpView &A, const SpView &B, bool apat, bool bpat,
                                          int64_t i, int64_t k, int64_t j, int64_t pA, int64_t pB) {
    using X = typename S::x_t;
    const X a = load_x<X>(reinterpret_cast<const X *>(A.x), pA, A.iso, apat);
    const X b = load_x<X>(reinterpret_cast<const X *>(B.x), pB, B.iso, bpat);
    return sr.mult(a, b, i, k, j);
}

// what one product reads, fetched apart from the multiply so that the loads can be issued early
template <class S>
struct PanelOperands { typename S::x_t a, b; int64_t k; };

template <class S>
__device__ inline PanelOperands<S> fetch_operands(const S &sr, const SpView &A, const SpView &B, bool apat, bool bpat,
                                                  int64_t pA, int64_t pB) {
    using X = typename S::x_t;
    PanelOperands<S> r;
    r.a = load_x<X>(reinterpret_cast<const X *>(A.x), pA, A.iso, apat);
    r.b = load_x<X>(reinterpret_cast<const X *>(B.x), pB, B.iso, bpat);
    r.k = S::positional ? B.i[pB] : 0;
    return r;
}

template <class S>
__global__ void __launch_bounds__(256)
num_wave_kernel(S sr, int64_t nlist, const int64_t *__restrict__ list, SpView A, SpView B, MaskView M, int lpa,
                const int64_t *__restrict__ Cp, int64_t *__restrict__ Ci, typename S::z_t *__restrict__ Cx) {
    using Z = typename S::z_t; using ACC = typename S::acc_t;
    __shared__ __align__(16) ACC vals_all[4 * NUM_WAVE_SLOTS];
    __shared__ int32_t keys_all[4 * NUM_WAVE_SLOTS];
    __shared__ int cursor[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t li = (int64_t)blockIdx.x * 4 + wave;
    int32_t *keys = keys_all + wave * NUM_WAVE_SLOTS;
    ACC *vals = vals_all + wave * NUM_WAVE_SLOTS;
    const ACC ident = (ACC)sr.identity();
    for (int s = lane; s < NUM_WAVE_SLOTS; s += 64) { keys[s] = HASH_EMPTY; vals[s] = ident; }
    if (lane == 0) cursor[wave] = 0;
    __builtin_amdgcn_wave_barrier();
    if (li >= nlist) return;
    const int64_t j = list[li];
    const bool apat = sr.a_is_pattern(), bpat = sr.b_is_pattern();
    for_each_product(A, B, j, lane, 64, lpa, [&](int64_t i, int64_t k, int64_t pA, int64_t pB) {
        if (!mask_allows(M, i, j)) return;
        const Z t = product(sr, A, B, apat, bpat, i, k, j, pA, pB);
        bool is_new;
        const uint32_t h = hash_insert(keys, NUM_WAVE_SLOTS - 1, (int32_t)i, &is_new);
        sr.atomic(&vals[h], t);
    });
    __builtin_amdgcn_wave_barrier();
    const int64_t pC = Cp[j];
    for (int s = lane; s < NUM_WAVE_SLOTS; s += 64) {
        const int32_t key = keys[s];
        if (key != HASH_EMPTY) {
            const int pos = atomicAdd(&cursor[wave], 1);
            Ci[pC + pos] = key;
            Cx[pC + pos] = (Z)vals[s];
        }
    }
}

template <class S, int SLOTS, int THREADS, bool ONE_PASS = false>
__global__ void __launch_bounds__(THREADS)
num_block_kernel(S sr, int64_t nlist, const int64_t *__restrict__ list, SpView A, SpView B, MaskView M, int lpa,
                 const int64_t *__restrict__ Cp, int64_t *__restrict__ Ci, typename S::z_t *__restrict__ Cx,
                 const int64_t *__restrict__ small_flops, int64_t *__restrict__ small_cnt) {
    using Z = typename S::z_t; using ACC = typename S::acc_t;
    extern __shared__ __align__(16) unsigned char smem[];
    ACC *vals = reinterpret_cast<ACC *>(smem);
    int32_t *keys = reinterpret_cast<int32_t *>(smem + (size_t)SLOTS * sizeof(ACC));
    __shared__ int cursor;
    __shared__ int longq[LONGQ_CAP];
    __shared__ int longn;
    // One-pass mode for a product of a few light vectors (ONE_PASS): no list, no symbolic count.  Vector
    // blockIdx.x is emitted into a slab of SMALL_ROOM entries of its own and its entry count is reported, or -1
    // when its multiply-adds could overfill the table (the caller then takes the two-pass road).
    const int64_t j = ONE_PASS ? (int64_t)blockIdx.x : list[blockIdx.x];
    if constexpr (ONE_PASS) {
        if (small_flops[j] > (int64_t)SMALL_ROOM) {
            if (threadIdx.x == 0) small_cnt[j] = -1;
            return;
        }
    }
    const ACC ident = (ACC)sr.identity();
    for (int s = threadIdx.x; s < SLOTS; s += THREADS) { keys[s] = HASH_EMPTY; vals[s] = ident; }
    if (threadIdx.x == 0) cursor = 0;
    __syncthreads();
    const bool apat = sr.a_is_pattern(), bpat = sr.b_is_pattern();
    for_each_product_block<THREADS>(A, B, j, lpa, longq, &longn, [&](int64_t i, int64_t k, int64_t pA, int64_t pB) {
        if (!mask_allows(M, i, j)) return;
        const Z t = product(sr, A, B, apat, bpat, i, k, j, pA, pB);
        bool is_new;
        const uint32_t h = hash_insert(keys, SLOTS - 1, (int32_t)i, &is_new);
        sr.atomic(&vals[h], t);
    });
    __syncthreads();
    const int64_t pC = ONE_PASS ? j * (int64_t)SMALL_ROOM : Cp[j];
    for (int s = threadIdx.x; s < SLOTS; s += THREADS) {
        const int32_t key = keys[s];
        if (key != HASH_EMPTY) {
            const int pos = atomicAdd(&cursor, 1);
            Ci[pC + pos] = key;
            Cx[pC + pos] = (Z)vals[s];
        }
    }
    if constexpr (ONE_PASS) {
        __syncthreads();
        if (threadIdx.x == 0) small_cnt[j] = cursor;
    }
}

// one-pass mode, second step: the slabs are packed into C (Cp holds the prefix sum of the counts)
template <class Z>
__global__ void __launch_bounds__(256)
small_pack_kernel(const int64_t *__restrict__ Cp, const int64_t *__restrict__ slab_i, const Z *__restrict__ slab_x,
                  int64_t *__restrict__ Ci, Z *__restrict__ Cx) {
    const int64_t j = blockIdx.x, pC = Cp[j], n = Cp[j + 1] - pC;
    for (int64_t e = threadIdx.x; e < n; e += 256) {
        Ci[pC + e] = slab_i[j * (int64_t)SMALL_ROOM + e];
        Cx[pC + e] = slab_x[j * (int64_t)SMALL_ROOM + e];
    }
}

template <class ACC> __device__ inline ACC ld_agent_acc(const ACC *p) {
    if constexpr (sizeof(ACC) == 4) {
        uint32_t r = __hip_atomic_load(reinterpret_cast<const uint32_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return *reinterpret_cast<ACC *>(&r);
    } else {
        uint64_t r = __hip_atomic_load(reinterpret_cast<const uint64_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return *reinterpret_cast<ACC *>(&r);
    }
}

// Dense Gustavson accumulator Hx[cvlen] per persistent workgroup, kept at the
// monoid identity between vectors; the gather walks the bitmaps in index order
// so these vectors come out sorted.
template <class S>
__global__ void __launch_bounds__(GLOBAL_THREADS)
num_global_kernel(S sr, int64_t nlist, const int64_t *__restrict__ list, SpView A, SpView B, MaskView M, int lpa,
                  int64_t cvlen, typename S::acc_t *__restrict__ Hx_all, uint64_t *__restrict__ Hf_all,
                  uint64_t *__restrict__ Hs_all, unsigned long long *__restrict__ next,
                  const int64_t *__restrict__ Cp, int64_t *__restrict__ Ci, typename S::z_t *__restrict__ Cx) {
    using Z = typename S::z_t; using ACC = typename S::acc_t;
    __shared__ int64_t cur;
    __shared__ int wsum[GLOBAL_THREADS / 64];
    __shared__ int longq[LONGQ_CAP];
    __shared__ int longn;
    const int64_t nf = (cvlen + 63) >> 6, ns = (nf + 63) >> 6;
    ACC *Hx = Hx_all + (int64_t)blockIdx.x * cvlen;
    uint64_t *Hf = Hf_all + (int64_t)blockIdx.x * nf;
    uint64_t *Hs = Hs_all + (int64_t)blockIdx.x * ns;
    const ACC ident = (ACC)sr.identity();
    const bool apat = sr.a_is_pattern(), bpat = sr.b_is_pattern();
    const int64_t chunk = (ns + GLOBAL_THREADS - 1) / GLOBAL_THREADS;   // Hs words per thread, contiguous
    while (true) {
        if (threadIdx.x == 0) cur = (int64_t)atomicAdd(next, 1ull);
        __syncthreads();
        const int64_t li = cur;
        if (li >= nlist) break;
        const int64_t j = list[li];
        for_each_product_block<GLOBAL_THREADS>(A, B, j, lpa, longq, &longn, [&](int64_t i, int64_t k, int64_t pA, int64_t pB) {
            if (!mask_allows(M, i, j)) return;
            const Z t = product(sr, A, B, apat, bpat, i, k, j, pA, pB);
            sr.atomic(&Hx[i], t);
            mark_slot(Hf, Hs, i);
        });
        __syncthreads();
        // pass 1: entries owned by this thread's contiguous range of Hs words
        const int64_t s0 = (int64_t)threadIdx.x * chunk;
        const int64_t s1 = (s0 + chunk < ns) ? s0 + chunk : ns;
        int c = 0;
        for (int64_t s = s0; s < s1; s++) {
            uint64_t sw = ld_agent(&Hs[s]);
            while (sw) {
                const int b = __builtin_ctzll(sw);
                sw &= sw - 1;
                c += __builtin_popcountll(ld_agent(&Hf[(s << 6) + b]));
            }
        }
        // exclusive scan of c over the workgroup
        int incl = c;
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { int u = __shfl_up(incl, d, 64); if (lane >= d) incl += u; }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int off = 0;
        for (int q = 0; q < wv; q++) off += wsum[q];
        int64_t pC = Cp[j] + off + incl - c;
        // pass 2: emit in index order and restore the workspace
        for (int64_t s = s0; s < s1; s++) {
            uint64_t sw = ld_agent(&Hs[s]);
            if (!sw) continue;
            while (sw) {
                const int b = __builtin_ctzll(sw);
                sw &= sw - 1;
                const int64_t w = (s << 6) + b;
                uint64_t fw = ld_agent(&Hf[w]);
                while (fw) {
                    const int q = __builtin_ctzll(fw);
                    fw &= fw - 1;
                    const int64_t i = (w << 6) + q;
                    Ci[pC] = i;
                    Cx[pC] = (Z)ld_agent_acc<ACC>(&Hx[i]);
                    Hx[i] = ident;
                    pC++;
                }
                Hf[w] = 0;
            }
            Hs[s] = 0;
        }
        __syncthreads();
    }
}

template <class ACC>
__global__ void fill_acc_kernel(ACC *p, int64_t n, ACC v) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; idx < n; idx += stride) p[idx] = v;
}

// ------------------------------------------------------------- team kernels --
// A handful of very heavy vectors -- a frontier step w = A*q has ONE -- would each occupy a single
// workgroup while the rest of the chip idles.  The CPU forms a team of fine tasks per such vector that
// share one Hf/Hx workspace through atomics (Source/GB_AxB_saxpy3_slice_balanced.c:560-771,
// Source/Template/GB_AxB_saxpy3_fineGus_phase2.c).  Here the whole grid is the team: every wavefront takes
// neighbours k of B(:,j) and scatters A(:,k) into a dense accumulator Hx[cvlen] with the monoid's atomic,
// marking touched indices in the bitmap Hf; columns longer than TEAM_LONG are shared by groups of workgroups.
// The products are formed once, during the symbolic phase (the count is the number of marks); the
// numeric phase only gathers.  Entries come out in no particular order (C is returned jumbled).
static constexpr int TEAM_MAX_VECTORS = 32;
static constexpr int64_t TEAM_MIN_FLOPS = 200000;
static constexpr int TEAM_THREADS = 256;
static constexpr int64_t TEAM_LONG = 8192;
static constexpr int TEAM_LONG_WGS = 8;           // workgroups that share one long column (the grid is a multiple of it)

__global__ void __launch_bounds__(256)
team_select_kernel(int64_t nvec, const int64_t *__restrict__ vflops, int64_t min_flops,
                   unsigned long long *__restrict__ count, int64_t *__restrict__ list, int cap) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; idx < nvec; idx += stride) {
        if (vflops[idx] > min_flops) {
            const unsigned long long q = atomicAdd(count, 1ull);
            if (q < (unsigned long long)cap) list[q] = idx;
        }
    }
}

// out[0] = multiply-adds of the product, out[1] = entries of M in the vectors that B fills
__global__ void __launch_bounds__(256)
mask_work_kernel(int64_t nvec, const int64_t *__restrict__ Bp, const int64_t *__restrict__ Mp,
                 const int64_t *__restrict__ vflops, unsigned long long *__restrict__ out) {
    __shared__ unsigned long long part[2][4];
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long f = 0, m = 0;
    for (; j < nvec; j += stride) {
        f += (unsigned long long)vflops[j];
        if (Bp[j + 1] > Bp[j]) m += (unsigned long long)(Mp[j + 1] - Mp[j]);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { f += __shfl_down(f, d, 64); m += __shfl_down(m, d, 64); }
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = f; part[1][threadIdx.x >> 6] = m; }
    __syncthreads();
    if (threadIdx.x == 0) {
        f = part[0][0] + part[0][1] + part[0][2] + part[0][3];
        m = part[1][0] + part[1][1] + part[1][2] + part[1][3];
        if (f) atomicAdd(&out[0], f);
        if (m) atomicAdd(&out[1], m);
    }
}

__global__ void __launch_bounds__(256)
zero_listed_kernel(int n, const int64_t *__restrict__ list, int64_t *__restrict__ work) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) work[list[q]] = 0;
}

template <class S>
__device__ inline void team_update(const S &sr, const SpView &A, const SpView &B, const MaskView &M, bool apat, bool bpat,
                                   int64_t j, int64_t k, int64_t pA, int64_t pB,
                                   typename S::acc_t *__restrict__ Hx, uint64_t *__restrict__ Hf) {
    const int64_t i = A.i[pA];
    if (!mask_allows(M, i, j)) return;
    const uint64_t bit = 1ull << (i & 63);
    const bool seen = (ld_agent(&Hf[i >> 6]) & bit) != 0;
    if (seen && sr.add_is_any()) return;                      // ANY keeps the value some earlier product left there
    sr.atomic(&Hx[i], product(sr, A, B, apat, bpat, i, k, j, pA, pB));
    if (!seen) atomicOr(reinterpret_cast<unsigned long long *>(&Hf[i >> 6]), (unsigned long long)bit);
}

template <class S>
__global__ void __launch_bounds__(TEAM_THREADS)
team_product_kernel(S sr, int64_t j, SpView A, SpView B, MaskView M, typename S::acc_t *__restrict__ Hx,
                    uint64_t *__restrict__ Hf, unsigned long long *__restrict__ nlong, int64_t *__restrict__ longq) {
    const int lane = threadIdx.x & 63;
    const int64_t gw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t pB0 = B.p[j], deg = B.p[j + 1] - pB0;
    const bool apat = sr.a_is_pattern(), bpat = sr.b_is_pattern();
    for (int64_t t = gw; t < deg; t += nw) {
        const int64_t k = B.i[pB0 + t];
        const int64_t pA0 = A.p[k], pA1 = A.p[k + 1];
        if (pA1 - pA0 > TEAM_LONG) {
            if (lane == 0) longq[atomicAdd(nlong, 1ull)] = t;
            continue;
        }
        for (int64_t pA = pA0 + lane; pA < pA1; pA += 64) team_update(sr, A, B, M, apat, bpat, j, k, pA, pB0 + t, Hx, Hf);
    }
}

template <class S>
__global__ void __launch_bounds__(TEAM_THREADS)
team_long_kernel(S sr, int64_t j, SpView A, SpView B, MaskView M, typename S::acc_t *__restrict__ Hx,
                 uint64_t *__restrict__ Hf, const unsigned long long *__restrict__ nlong, const int64_t *__restrict__ longq) {
    // groups of TEAM_LONG_WGS workgroups take the long columns in turn: a frontier of hub vertices queues
    // thousands of them, and a pass of the whole grid over each costs a grid-wide round trip per column
    const int64_t grp = blockIdx.x / TEAM_LONG_WGS, ngrp = gridDim.x / TEAM_LONG_WGS;
    const int64_t gt = (int64_t)(blockIdx.x % TEAM_LONG_WGS) * blockDim.x + threadIdx.x, nt = (int64_t)TEAM_LONG_WGS * blockDim.x;
    const int64_t pB0 = B.p[j];
    const bool apat = sr.a_is_pattern(), bpat = sr.b_is_pattern();
    const int64_t n = (int64_t)*nlong;
    for (int64_t q = grp; q < n; q += ngrp) {
        const int64_t t = longq[q], k = B.i[pB0 + t];
        const int64_t pA1 = A.p[k + 1];
        for (int64_t pA = A.p[k] + gt; pA < pA1; pA += nt) team_update(sr, A, B, M, apat, bpat, j, k, pA, pB0 + t, Hx, Hf);
    }
}

__global__ void __launch_bounds__(256)
team_count_kernel(int64_t nf, const uint64_t *__restrict__ Hf, unsigned long long *__restrict__ out) {
    __shared__ unsigned long long part[4];
    int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long c = 0;
    for (; w < nf; w += stride) c += (unsigned long long)__builtin_popcountll(Hf[w]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) { c = part[0] + part[1] + part[2] + part[3]; if (c) atomicAdd(out, c); }
}

template <class S>
__global__ void __launch_bounds__(256)
team_gather_kernel(int64_t nf, const uint64_t *__restrict__ Hf, const typename S::acc_t *__restrict__ Hx,
                   unsigned long long *__restrict__ cursor, const int64_t *__restrict__ Cp, int64_t j,
                   int64_t *__restrict__ Ci, typename S::z_t *__restrict__ Cx) {
    using Z = typename S::z_t;
    const int64_t pC0 = Cp[j];
    const int lane = threadIdx.x & 63;
    int64_t w0 = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) << 6;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; w0 < nf; w0 += stride) {
        const int64_t w = w0 + lane;
        uint64_t fw = (w < nf) ? Hf[w] : 0ull;
        const int c = __builtin_popcountll(fw);
        int incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(incl, d, 64); if (lane >= d) incl += u; }
        const int total = __shfl(incl, 63, 64);
        if (total == 0) continue;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(cursor, (unsigned long long)total);
        base = __shfl(base, 0, 64);
        int64_t pC = pC0 + (int64_t)base + incl - c;
        while (fw) {
            const int64_t i = (w << 6) + __builtin_ctzll(fw);
            fw &= fw - 1;
            Ci[pC] = i;
            Cx[pC] = (Z)Hx[i];
            pC++;
        }
    }
}

// ------------------------------------------------------------ panel kernels --
// Vectors too heavy for one LDS table (the CPU's Gustavson tasks and fine-task
// teams).  HBM atomics into a dense accumulator run at the memory-side atomic
// rate (~2e10 updates/s chip-wide), far below what the products need, so the
// accumulator stays in LDS and the index range 0..cvlen-1 is swept in PANELS:
// because every A(:,k) is sorted, the entries of A(:,k) that fall into the
// current panel are one contiguous run starting at a per-neighbour cursor.
//   symbolic: fixed panels of SYM_PANEL_WIDTH indices, one LDS bit per index
//   numeric : LDS hash table; the panel width adapts to the density of the
//             vector (a panel that overflows the table is redone at half width)
// One persistent 1024-thread workgroup per CU pulls vectors from a counter.
// Requires sorted A (not jumbled) and fewer than 2^24 entries in every B(:,j);
// otherwise the HBM-accumulator kernels above run.
//
// Per neighbour t of B(:,j) the workgroup keeps, in HBM scratch, the cursor
// into A(:,k), the index stored at the cursor, and the end of A(:,k).  A
// neighbour with nothing inside the panel is dismissed from those coalesced
// arrays alone, without touching A.
#ifndef GBA_TABLE_MAX_DEG
#define GBA_TABLE_MAX_DEG 512
#endif
#ifdef GBA_PANEL_HALF
static constexpr int PANEL_THREADS = 512;
static constexpr int PANEL_WG_PER_CU = 2;
#else
static constexpr int PANEL_THREADS = 1024;
static constexpr int PANEL_WG_PER_CU = 1;
#endif
static constexpr int PANEL_NPT = 1;                                   // neighbours searched per thread per step
static constexpr int PANEL_STEP = PANEL_THREADS * PANEL_NPT;
static constexpr int PANEL_SEGCAP = PANEL_THREADS;                    // runs queued in LDS before they are expanded: one search step's worth (LDS spent on
                                                                      // the queue is LDS taken from the tables, and larger tables mean fewer panels)
static constexpr int PANEL_UNROLL_MAX = 8;
// two windows per wave in flight (the loads of one issued before the other is consumed)?
static constexpr bool SYM_PANEL_PIPELINE = false, NUM_PANEL_PIPELINE = false;
#ifndef GBA_NUM_UNROLL
#define GBA_NUM_UNROLL 4
#endif
#ifndef GBA_SYM_UNROLL
#define GBA_SYM_UNROLL 8
#endif
static constexpr int SYM_PANEL_UNROLL = GBA_SYM_UNROLL, NUM_PANEL_UNROLL = GBA_NUM_UNROLL;            // products in flight per thread
static constexpr int PANEL_T_BITS = 24;
static constexpr int64_t PANEL_MAX_DEG = (int64_t)1 << PANEL_T_BITS;
static constexpr int64_t PANEL_MAX_WIDTH = (int64_t)1 << 20;          // (SEGCAP-1) runs of this length stay below 2^32
static constexpr int SYM_PANEL_WORDS = 32 * PANEL_THREADS;             // 128 KiB of LDS
static constexpr int64_t SYM_PANEL_WIDTH = (int64_t)SYM_PANEL_WORDS * 32;
static constexpr int NUM_PANEL_SLOTS = 11 * PANEL_THREADS;                        // (any multiple of PANEL_THREADS: slots are picked by multiply-shift)
static constexpr int NUM_PANEL_TARGET = 3000;                         // distinct indices aimed at per panel (no boundaries known)
static constexpr int NUM_PANEL_FIT = NUM_PANEL_SLOTS / 4 * 3;                            // entries between two recorded boundaries: three quarters of the slots
static constexpr int SYM_WORDS_PER_THREAD = SYM_PANEL_WORDS / PANEL_THREADS;
static constexpr int SYM_HASH_BITS = (SYM_PANEL_WORDS == 32768) ? 15 : 14;       // the bitmap's words as a hash set of a light vector
static_assert((1 << SYM_HASH_BITS) == SYM_PANEL_WORDS && NUM_PANEL_FIT * 3 <= SYM_PANEL_WORDS, "the hash set stays under a third full");
// The bitmap is stored with one word of padding after every thread's slice: slice t starts at word t * (SWPT + 1).
// In the passes that count, list and clear the bitmap every thread walks its own slice; without the padding the
// lanes of a wave are 32 words apart and fall on the same LDS banks (an 8-way conflict on every 16-byte access),
// with it lane t reads bank t.
static constexpr int SYM_SLICE_STRIDE = SYM_WORDS_PER_THREAD + 1;
static constexpr int SYM_PANEL_LDS_WORDS = PANEL_THREADS * SYM_SLICE_STRIDE;
static_assert(SYM_WORDS_PER_THREAD == 32, "the slice of word w is w >> 5");
static constexpr int PANEL_PROBE_LIMIT = 48;
static constexpr int64_t PANEL_INF = (int64_t)1 << 62;
static constexpr int PANEL_SCRATCH_ARRAYS = 5;
static_assert(SYM_PANEL_WIDTH <= PANEL_MAX_WIDTH, "symbolic panel too wide for 32-bit product offsets");
static_assert((uint64_t)(PANEL_SEGCAP - 1) * PANEL_MAX_WIDTH + 64 * PANEL_UNROLL_MAX < (1ull << 32), "window ends fit 32 bits");
static_assert(SYM_WORDS_PER_THREAD * PANEL_THREADS == SYM_PANEL_WORDS && SYM_WORDS_PER_THREAD * 32 < NUM_PANEL_FIT,
              "each thread owns a bitmap slice that can hold at most one boundary");

// Optional in-kernel stopwatch (thread 0, 100 MHz ticks) for tuning: build with -DGBA_PANEL_TICKS and run with
// GBA_PANEL_PROF=1; slots: 0 search, 1 wait for the other waves' search, 2 prefix sum, 3 products,
// 4 wait for the other waves' products, 5 emit, 6 whole panels.
#ifdef GBA_PANEL_TICKS
#define GBA_TICK(var) const unsigned long long var = wall_clock64()
#define GBA_TOCK(S, slot, a, b) do { if (threadIdx.x == 0) (S)->tk[slot] += (b) - (a); } while (0)
#else
#define GBA_TICK(var) do {} while (0)
#define GBA_TOCK(S, slot, a, b) do {} while (0)
#endif

// runs of the current panel waiting to be expanded (LDS)
struct PanelSegs {
    unsigned long long where[PANEL_SEGCAP];   // position of the run in A (high 40 bits) | neighbour ordinal (low 24)
    uint32_t off[PANEL_SEGCAP];               // run length; exclusive prefix sum once the runs are expanded
    uint32_t wsum[PANEL_THREADS / 64];
    uint64_t wsum2[PANEL_THREADS / 64];        // the symbolic sweep's packed sums (entries | non-empty words)
    unsigned long long wmask[PANEL_THREADS / 64][PANEL_UNROLL_MAX];   // per wave: products of the window at which a run starts
    int nseg;
    uint32_t wnext;                           // next window of products to hand to a wave
    int more;                                 // some neighbour still has entries below the end of the sweep
    int overflow;                             // the numeric table ran out of room
#ifdef GBA_PANEL_TICKS
    unsigned long long tk[8];
#endif
};

struct PanelCursors {
    int64_t *cur_in, *nv_in, *cur_out, *nv_out, *end;
    __device__ PanelCursors(int64_t *base, int64_t maxdeg)
        : cur_in(base), nv_in(base + maxdeg), cur_out(base + 2 * maxdeg), nv_out(base + 3 * maxdeg), end(base + 4 * maxdeg) {}
    __device__ void swap() {
        int64_t *a = cur_in; cur_in = cur_out; cur_out = a;
        int64_t *b = nv_in; nv_in = nv_out; nv_out = b;
    }
};

// a value every lane of the wave holds alike, moved to scalar registers (lane 0's copy)
__device__ inline int64_t uniform_i64(int64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | (uint64_t)lo);
}

// first p in [lo,hi) with X[p] >= target (hi if none): gallop from lo, then bisect
__device__ inline int64_t first_ge(const int32_t *__restrict__ X, int64_t lo, int64_t hi, int64_t target) {
    if (lo >= hi || X[lo] >= target) return lo;
    int64_t step = 1;
    while (lo + step < hi && X[lo + step] < target) { lo += step; step <<= 1; }
    int64_t a = lo + 1, b = (lo + step < hi) ? lo + step : hi;
    while (a < b) { const int64_t m = (a + b) >> 1; if (X[m] < target) a = m + 1; else b = m; }
    return a;
}

// Expand the n queued runs: they are laid end to end (prefix sum of their
// lengths) and cut into equal slabs of products, one per wave, PANEL_UNROLL
// products per lane in flight together, so the work is even however skewed
// the runs are.  load(i, pA, pB) fetches what one product
// needs; apply(i[], payload[], ok[]) consumes a batch of PANEL_UNROLL.  Needs a barrier between the last
// append and the call; ends on a barrier with the queue empty.
template <int PANEL_UNROLL, bool PANEL_PIPELINE, class Load, class Apply>
__device__ inline void panel_expand(const SpView &A, int64_t pB0, int n, PanelSegs *S, Load load, Apply apply, bool take_turns) {
    constexpr int QPT = PANEL_SEGCAP / PANEL_THREADS;      // queue entries per thread in the scan
    static_assert(QPT * PANEL_THREADS == PANEL_SEGCAP && PANEL_SEGCAP <= 64 * 64, "two 64-way steps find a run");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    GBA_TICK(tke0);
    const int b4 = tid * QPT;
    uint32_t v[QPT], s = 0;
#pragma unroll
    for (int q = 0; q < QPT; q++) { v[q] = (b4 + q < n) ? S->off[b4 + q] : 0u; s += v[q]; }
    uint32_t incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t u = __shfl_up(incl, d, 64); if (lane >= d) incl += u; }
    if (lane == 63) S->wsum[wv] = incl;
    __syncthreads();
    uint32_t woff = 0, T = 0;
#pragma unroll
    for (int q = 0; q < PANEL_THREADS / 64; q++) { const uint32_t w = S->wsum[q]; if (q < wv) woff += w; T += w; }
    uint32_t run = woff + incl - s;
#pragma unroll
    for (int q = 0; q < QPT; q++) if (b4 + q < n) { S->off[b4 + q] = run; run += v[q]; }
    if (tid == 0) { S->nseg = 0; S->wnext = 0; }
    __syncthreads();
    GBA_TICK(tke1);
    GBA_TOCK(S, 2, tke0, tke1);
    const int32_t *__restrict__ Ai = A.i32;
    // The products are cut into windows of WIN consecutive ones.  With take_turns and few windows (a numeric
    // panel has a few dozen) the waves take them from an LDS counter as they get free -- the time per window
    // varies with the hash probes and the cache misses it meets; otherwise (many windows, or the symbolic
    // sweep, whose windows cost the same) every wave owns a contiguous slab.  A wave bisects the prefix sums for the run holding the first product
    // of a window (uniform address: broadcast reads; inside a slab the next window's run follows from
    // the marks).  The runs that START inside the window are read 64 at a time (one conflict-free LDS
    // read), marked in WIN bits, and a product's run is the window's first run plus the number of
    // marks at or before it.
    constexpr uint32_t WIN = 64 * PANEL_UNROLL;
    constexpr int NWAVES = PANEL_THREADS / 64;
    const uint32_t nwin = (uint32_t)(((uint64_t)T + WIN - 1) / WIN);
    {
        const bool dynamic = take_turns && nwin < (uint32_t)(NWAVES * 16);
        const uint32_t per = (nwin + NWAVES - 1) / NWAVES;
        uint32_t win = (uint32_t)wv * per;
        const uint32_t win_end = (win + per < nwin) ? win + per : nwin;
        int sb = 0;
        bool placed = false;
        unsigned long long *wm = S->wmask[wv];
        using Raw = decltype(load(0, (int64_t)0, (int64_t)0));
        // One window: take it, find the run of each of its products, and ISSUE the loads of what the products
        // need (index of A, and whatever load() fetches).  Nothing loaded is touched here, so the caller can
        // work on the window before while these loads are in flight.  When no window is left (returns
        // false) the first window is staged again with every product switched off: the same loads are
        // issued on every path, which lets the compiler count them instead of draining the queue.
        auto stage = [&](int32_t (&ii)[PANEL_UNROLL], Raw (&raw)[PANEL_UNROLL], bool (&ok)[PANEL_UNROLL]) -> bool {
            bool real = true;
            if (dynamic) {
                if (lane == 0) win = atomicAdd(&S->wnext, 1u);
                win = __shfl(win, 0, 64);
                if (win >= nwin) { if (!PANEL_PIPELINE) return false; real = false; win = 0; }
            } else if (win >= win_end) {
                if (!PANEL_PIPELINE) return false;
                real = false; win = 0; placed = false; sb = 0;
            }
            const uint32_t w0 = win * WIN;          // T + WIN < 2^32: 32-bit offsets throughout
            if (dynamic || !placed) {
                // largest s with off[s] <= w0 (off[0] = 0), in two 64-way steps: every stride-th prefix sum,
                // then the stride of them that brackets w0 (n <= 64 * 64)
                const int stride = (n + 63) >> 6;
                const int i1 = lane * stride;
                const unsigned long long m1 = __ballot(i1 < n && S->off[i1 < n ? i1 : 0] <= w0);
                const int blk = (int)__builtin_popcountll(m1) - 1;
                const int i2 = blk * stride + lane;
                const unsigned long long m2 = __ballot(lane < stride && i2 < n && S->off[i2 < n ? i2 : 0] <= w0);
                sb = blk * stride + (int)__builtin_popcountll(m2) - 1;
                placed = true;
            }
            const uint32_t wend = w0 + WIN;
            // No run starts inside the window (the common case when runs are long: symbolic panels): every
            // product belongs to run sb, its position is the run's start plus the lane -- one uniform base,
            // no marks, no per-product lookup.
            const uint32_t nxt = (sb + 1 < n) ? S->off[sb + 1] : ~0u;
            if (nxt >= wend) {
                const unsigned long long w = S->where[sb];
                const uint32_t rel = w0 - S->off[sb];
                const int64_t a0 = uniform_i64((int64_t)(w >> PANEL_T_BITS) + (int64_t)rel);
                const int64_t pBu = uniform_i64(pB0 + (int64_t)(w & (unsigned long long)(PANEL_MAX_DEG - 1)));
                const int32_t *__restrict__ Arun = Ai + a0;
                uint32_t xo[PANEL_UNROLL];
#pragma unroll
                for (int u = 0; u < PANEL_UNROLL; u++) {
                    const uint32_t x = (uint32_t)u * 64 + (uint32_t)lane;
                    const bool inside = w0 + x < T;
                    ok[u] = inside && real;
                    xo[u] = inside ? x : 0u;
                }
#pragma unroll
                for (int u = 0; u < PANEL_UNROLL; u++) ii[u] = Arun[xo[u]];
#pragma unroll
                for (int u = 0; u < PANEL_UNROLL; u++) raw[u] = load(ii[u], a0 + (int64_t)xo[u], pBu);
                if (!dynamic) { sb += (nxt == wend) ? 1 : 0; win = real ? win + 1 : win_end; }
                return real;
            }
            if (lane < PANEL_UNROLL) wm[lane] = 0ull;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            bool edge = false;                          // a run starts exactly at the next window
            for (int cursor = sb + 1;; cursor += 64) {
                const int idx = cursor + lane;
                const uint32_t o = (idx < n) ? S->off[idx] : ~0u;
                const bool in = o < wend;               // off[] is strictly increasing and off[sb + 1] > w0
                if (in) { const uint32_t q = o - w0; atomicOr(&wm[q >> 6], 1ull << (q & 63)); }
                if (__ballot(o == wend)) edge = true;
                if (__ballot(in) != ~0ull) break;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            unsigned long long mk[PANEL_UNROLL];
            int starts = 0;
#pragma unroll
            for (int u = 0; u < PANEL_UNROLL; u++) {           // (a volatile read here compiles to a flat load that drains every load in flight)
                mk[u] = __hip_atomic_load(&wm[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                starts += __builtin_popcountll(mk[u]);
            }
            // a product's run = the window's first run + the starts marked at or before the product: the marks of
            // the words before its own are counted once for all lanes, its own word takes one popcount.  Lanes
            // past the last product (last window only) re-read the window's first product.
            int pre[PANEL_UNROLL];
            pre[0] = 0;
#pragma unroll
            for (int u = 1; u < PANEL_UNROLL; u++) pre[u] = pre[u - 1] + __builtin_popcountll(mk[u - 1]);
            const unsigned long long upto = ~0ull >> (63 - lane);
            int64_t pA[PANEL_UNROLL], pB[PANEL_UNROLL];
#pragma unroll
            for (int u = 0; u < PANEL_UNROLL; u++) {
                const uint32_t x = (uint32_t)u * 64 + (uint32_t)lane;
                const bool inside = w0 + x < T;
                ok[u] = inside && real;
                const int seg = sb + (inside ? pre[u] + (int)__builtin_popcountll(mk[u] & upto) : 0);
                const uint32_t pp = w0 + (inside ? x : 0u);
                const unsigned long long w = S->where[seg];
                pA[u] = (int64_t)(w >> PANEL_T_BITS) + (int64_t)(pp - S->off[seg]);
                pB[u] = pB0 + (int64_t)(w & (unsigned long long)(PANEL_MAX_DEG - 1));
            }
#pragma unroll
            for (int u = 0; u < PANEL_UNROLL; u++) ii[u] = Ai[pA[u]];
#pragma unroll
            for (int u = 0; u < PANEL_UNROLL; u++) raw[u] = load(ii[u], pA[u], pB[u]);
            if (!dynamic) { sb += starts + (edge ? 1 : 0); win = real ? win + 1 : win_end; }
            return real;
        };
        // two windows per wave in flight: the loads of one are issued before the other is consumed
        int32_t iiA[PANEL_UNROLL], iiB[PANEL_UNROLL];
        Raw rawA[PANEL_UNROLL], rawB[PANEL_UNROLL];
        bool okA[PANEL_UNROLL], okB[PANEL_UNROLL];
        if constexpr (!PANEL_PIPELINE) {
            while (stage(iiA, rawA, okA)) apply(iiA, rawA, okA);
        } else {
        bool liveA = stage(iiA, rawA, okA);
        while (liveA) {
            const bool liveB = stage(iiB, rawB, okB);
            apply(iiA, rawA, okA);
            if (!liveB) break;
            liveA = stage(iiA, rawA, okA);
            apply(iiB, rawB, okB);
        }
        }
    }
    GBA_TICK(tke2);
    __syncthreads();
    GBA_TICK(tke3);
    GBA_TOCK(S, 3, tke1, tke2); GBA_TOCK(S, 4, tke2, tke3);
}

// First p in [u,e) with X[p] >= target (e if none), found by the G lanes of a group together: the lanes
// probe G positions a stride apart, the stride grows G-fold while every probe is below the target and
// shrinks G-fold once one is not, so a run of r entries costs about 2 log_G r dependent loads instead of
// 2 log_2 r.  G is a power of two <= 64 and a group sits inside one wave; gl = lane within the group,
// gb = first lane of the group.  Every lane of the wave must call (the loop is wave-uniform); groups
// with act == false only keep the others company.  Returns the index found at p through nv
// (PANEL_INF when p == e).
__device__ inline void group_first_ge(const int32_t *__restrict__ X, bool act, int64_t u, int64_t e, int64_t target,
                                      int lgG, int gl, int gb, int64_t &p, int64_t &nv) {
    const int G = 1 << lgG;
    const unsigned long long gmask = (G == 64) ? ~0ull : ((1ull << G) - 1ull);
    int64_t hi = e, s = 1;
    int32_t hv = 0;
    if (u > hi) u = hi;
    bool run = act && u < hi;
    while (__ballot(run)) {
        const int64_t pos = u + (int64_t)(gl + 1) * s - 1;
        const bool inb = run && pos < hi;
        const int32_t xv = inb ? X[pos] : 0;
        const unsigned long long bal = __ballot(inb && (int64_t)xv < target);
        const int cnt = (int)__builtin_popcountll((bal >> gb) & gmask);     // probes below the target: a prefix of the group
        const int src = gb + ((cnt < G) ? cnt : G - 1);
        const int32_t fv = __shfl(xv, src, 64);                             // the first probe that was not
        const int fin = __shfl((int)inb, src, 64);
        if (run) {
            if (cnt == G) { u += s << lgG; s <<= lgG; }
            else {
                u += (int64_t)cnt * s;
                if (fin) { hi = u + s - 1; hv = fv; }
                s = (s + G - 1) >> lgG;
            }
            if (u >= hi) { u = hi; run = false; }
        }
    }
    p = u;
    nv = (u < e) ? (int64_t)hv : PANEL_INF;
}

// panel_walk for a vector with few neighbours (deg * 2 <= PANEL_THREADS): their runs inside a panel are
// long, so each neighbour gets a group of lanes that searches its list together (group_first_ge).
template <int PANEL_UNROLL, bool PANEL_PIPELINE, class Load, class Apply>
__device__ inline void panel_walk_groups(const SpView &A, const SpView &B, int64_t pB0, int64_t deg, int64_t ibeg,
                                         int64_t pend, int64_t iend, bool first, const PanelCursors &pc, PanelSegs *S,
                                         Load load, Apply apply, bool take_turns) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int32_t *__restrict__ Ai = A.i32;
    int lgG = 6;
    while (lgG > 0 && ((int64_t)deg << lgG) > PANEL_THREADS) lgG--;
    const int G = 1 << lgG, gl = tid & (G - 1), gb = lane & ~(G - 1);
    const int64_t t = tid >> lgG;
    const bool live = t < deg;
    GBA_TICK(tk0);
    int64_t c = 0, e = 0, nv = PANEL_INF;
    if (live) {
        if (first) {
            const int64_t k = B.i[pB0 + t];
            c = A.p[k]; e = A.p[k + 1];
            if (gl == 0) pc.end[t] = e;
        } else {
            c = pc.cur_in[t]; nv = pc.nv_in[t]; e = pc.end[t];
        }
    }
    if (first) {
        if (ibeg > 0) group_first_ge(Ai, live, c, e, ibeg, lgG, gl, gb, c, nv);
        else if (live && c < e) nv = Ai[c];
    }
    const bool need = live && nv < pend;
    int64_t stop = c, nv2 = nv;
    if (pend >= PANEL_INF) { if (need) { stop = e; nv2 = PANEL_INF; } }
    else {
        int64_t ps = c, pn = nv;
        group_first_ge(Ai, need, c + 1, e, pend, lgG, gl, gb, ps, pn);      // Ai[c] = nv < pend where need
        if (need) { stop = ps; nv2 = pn; }
    }
    const bool writer = live && gl == 0;
    if (writer) {
        pc.cur_out[t] = stop;
        pc.nv_out[t] = nv2;
        if (nv2 < iend) S->more = 1;
    }
    const bool has = writer && stop > c;
    const unsigned long long m = __ballot(has);
    if (m) {
        int slot = 0;
        if (lane == 0) slot = atomicAdd(&S->nseg, (int)__builtin_popcountll(m));
        slot = __shfl(slot, 0, 64);
        if (has) {
            slot += (int)__builtin_popcountll(m & ((1ull << lane) - 1ull));
            S->where[slot] = ((unsigned long long)c << PANEL_T_BITS) | (unsigned long long)t;
            S->off[slot] = (uint32_t)(stop - c);
        }
    }
    GBA_TICK(tk1);
    __syncthreads();
    GBA_TICK(tk2);
    GBA_TOCK(S, 0, tk0, tk1); GBA_TOCK(S, 1, tk1, tk2);
    const int ns = S->nseg;
    if (ns > 0) panel_expand<PANEL_UNROLL, PANEL_PIPELINE>(A, pB0, ns, S, load, apply, take_turns);
    else __syncthreads();
}

// panel_walk for the numeric sweep of a vector whose panel boundaries are known in advance
// (recorded by the symbolic sweep): where every neighbour's list crosses every boundary of the task was looked up
// once, when the task started (tab[m * deg + t] = first position of neighbour t at or after boundary m; row nrows-1
// = end of the task's range), so a panel that runs from boundary row_lo to boundary row_hi reads its runs from two
// rows of that table instead of searching.  A panel that does not start or end on a boundary (after an overflow, or
// a directly addressed panel) passes -1 for that row and the position is found by bisection inside the task's
// range.  No cursors are kept.  Same barriers, the same queue handling and the same S->more contract as panel_walk.
template <int PANEL_UNROLL, bool PANEL_PIPELINE, class Load, class Apply>
__device__ inline void panel_walk_table(const SpView &A, int64_t pB0, int64_t deg, int64_t base, int64_t pend,
                                        const int64_t *__restrict__ tab, int nrows, int row_lo, int row_hi,
                                        PanelSegs *S, Load load, Apply apply, bool take_turns) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int32_t *__restrict__ Ai = A.i32;
    for (int64_t t0 = 0; t0 < deg; t0 += PANEL_THREADS) {
        GBA_TICK(tk0);
        const int64_t t = t0 + tid;
        const bool live = t < deg;
        int64_t lo = 0, hi = 0;
        if (live) {
            const int64_t first = tab[t], last = tab[(int64_t)(nrows - 1) * deg + t];
            if (row_lo >= 0) lo = tab[(int64_t)row_lo * deg + t];
            else { int64_t a = first, b = last; while (a < b) { const int64_t m = (a + b) >> 1; if ((int64_t)Ai[m] < base) a = m + 1; else b = m; } lo = a; }
            if (row_hi >= 0) hi = tab[(int64_t)row_hi * deg + t];
            else if (pend >= PANEL_INF) hi = last;
            else { int64_t a = lo, b = last; while (a < b) { const int64_t m = (a + b) >> 1; if ((int64_t)Ai[m] < pend) a = m + 1; else b = m; } hi = a; }
            if (hi < last) S->more = 1;
        }
        const bool has = live && hi > lo;
        const unsigned long long m = __ballot(has);
        if (m) {
            int slot = 0;
            if (lane == 0) slot = atomicAdd(&S->nseg, (int)__builtin_popcountll(m));
            slot = __shfl(slot, 0, 64);
            if (has) {
                slot += (int)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                S->where[slot] = ((unsigned long long)lo << PANEL_T_BITS) | (unsigned long long)t;
                S->off[slot] = (uint32_t)(hi - lo);
            }
        }
        GBA_TICK(tk1);
        __syncthreads();
        GBA_TICK(tk2);
        GBA_TOCK(S, 0, tk0, tk1); GBA_TOCK(S, 1, tk1, tk2);
        const int ns = S->nseg;
        const bool last_step = t0 + PANEL_THREADS >= deg;
        if (ns > 0 && (ns > PANEL_SEGCAP - PANEL_THREADS || last_step)) {
            panel_expand<PANEL_UNROLL, PANEL_PIPELINE>(A, pB0, ns, S, load, apply, take_turns);
            if (S->overflow) return;
        } else {
            __syncthreads();                                    // everyone has read nseg before the next append
        }
    }
}

// Visit the products A(i,k)*B(k,j) with ibeg <= i < pend that have not been
// visited yet (iend = end of the whole sweep; pend = PANEL_INF takes all that
// is left).  PANEL_STEP neighbours at a time: each thread advances the cursors
// of PANEL_NPT neighbours (galloping search, the loads of its neighbours issued
// together), non-empty runs are appended to the LDS queue, and the queue is
// expanded whenever it might not hold another step.  The caller provides a
// barrier between resetting S->more / S->overflow and the call; the call ends
// on a barrier.  Returns early (on a barrier) once S->overflow is raised.
template <int PANEL_UNROLL, bool PANEL_PIPELINE, class Load, class Apply>
__device__ inline void panel_walk(const SpView &A, const SpView &B, int64_t pB0, int64_t deg, int64_t ibeg,
                                  int64_t pend, int64_t iend, bool first, const PanelCursors &pc, PanelSegs *S,
                                  Load load, Apply apply, bool take_turns) {
    if (deg * 2 <= PANEL_THREADS) {
        panel_walk_groups<PANEL_UNROLL, PANEL_PIPELINE>(A, B, pB0, deg, ibeg, pend, iend, first, pc, S, load, apply, take_turns);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int32_t *__restrict__ Ai = A.i32;
    for (int64_t t0 = 0; t0 < deg; t0 += PANEL_STEP) {
        GBA_TICK(tk0);
        int64_t c[PANEL_NPT], e[PANEL_NPT], nv[PANEL_NPT], stop[PANEL_NPT];
        bool live[PANEL_NPT], need[PANEL_NPT];
#pragma unroll
        for (int q = 0; q < PANEL_NPT; q++) {
            const int64_t t = t0 + (int64_t)q * PANEL_THREADS + tid;
            live[q] = t < deg; c[q] = 0; e[q] = 0; nv[q] = PANEL_INF;
            if (live[q]) {
                if (first) {
                    const int64_t k = B.i[pB0 + t];
                    c[q] = A.p[k]; e[q] = A.p[k + 1];
                    pc.end[t] = e[q];
                } else {
                    // the end of the list is fetched with the cursor, not after the test that needs it
                    c[q] = pc.cur_in[t]; nv[q] = pc.nv_in[t]; e[q] = pc.end[t];
                }
            }
        }
        if (first) {
            if (ibeg > 0) {
#pragma unroll
                for (int q = 0; q < PANEL_NPT; q++) if (live[q]) c[q] = first_ge(Ai, c[q], e[q], ibeg);
            }
#pragma unroll
            for (int q = 0; q < PANEL_NPT; q++) if (live[q] && c[q] < e[q]) nv[q] = Ai[c[q]];
        }
#pragma unroll
        for (int q = 0; q < PANEL_NPT; q++) {
            need[q] = live[q] && nv[q] < pend;
            stop[q] = c[q];
        }
        int64_t hp[PANEL_NPT]; int32_t hv[PANEL_NPT];       // last probe that met an index >= pend: where, and the index
#pragma unroll
        for (int q = 0; q < PANEL_NPT; q++) { hp[q] = -1; hv[q] = 0; }
        if (pend >= PANEL_INF) {
#pragma unroll
            for (int q = 0; q < PANEL_NPT; q++) if (need[q]) stop[q] = e[q];
        } else {
            // Ai[c] < pend is known where need: gallop to bracket the first entry >= pend, then bisect
            int64_t lo[PANEL_NPT], st[PANEL_NPT];
            bool g[PANEL_NPT];
            bool any = false;
#pragma unroll
            for (int q = 0; q < PANEL_NPT; q++) { lo[q] = c[q]; st[q] = 1; g[q] = need[q]; any |= g[q]; }
            while (any) {
                int64_t pos[PANEL_NPT]; int32_t xv[PANEL_NPT]; bool inb[PANEL_NPT];
#pragma unroll
                for (int q = 0; q < PANEL_NPT; q++) {
                    pos[q] = lo[q] + st[q];
                    inb[q] = g[q] && pos[q] < e[q];
                    xv[q] = Ai[inb[q] ? pos[q] : lo[q]];
                }
                any = false;
#pragma unroll
                for (int q = 0; q < PANEL_NPT; q++) {
                    if (inb[q] && xv[q] < pend) { lo[q] = pos[q]; st[q] <<= 1; }
                    else { if (inb[q]) { hp[q] = pos[q]; hv[q] = xv[q]; } g[q] = false; }
                    any |= g[q];
                }
            }
            int64_t a[PANEL_NPT], b[PANEL_NPT];
            any = false;
#pragma unroll
            for (int q = 0; q < PANEL_NPT; q++) {
                a[q] = lo[q] + 1;
                b[q] = (lo[q] + st[q] < e[q]) ? lo[q] + st[q] : e[q];
                if (!need[q]) { a[q] = 0; b[q] = 0; }
                any |= a[q] < b[q];
            }
            while (any) {
                int64_t m[PANEL_NPT]; int32_t xv[PANEL_NPT]; bool act[PANEL_NPT];
#pragma unroll
                for (int q = 0; q < PANEL_NPT; q++) {
                    act[q] = a[q] < b[q];
                    m[q] = (a[q] + b[q]) >> 1;
                    xv[q] = Ai[act[q] ? m[q] : lo[q]];
                }
                any = false;
#pragma unroll
                for (int q = 0; q < PANEL_NPT; q++) {
                    if (act[q]) { if (xv[q] < pend) a[q] = m[q] + 1; else { b[q] = m[q]; hp[q] = m[q]; hv[q] = xv[q]; } }
                    any |= a[q] < b[q];
                }
            }
#pragma unroll
            for (int q = 0; q < PANEL_NPT; q++) if (need[q]) stop[q] = a[q];
        }
#pragma unroll
        for (int q = 0; q < PANEL_NPT; q++) {
            const int64_t t = t0 + (int64_t)q * PANEL_THREADS + tid;
            if (live[q]) {
                int64_t nv2 = nv[q];
                if (need[q]) nv2 = (stop[q] < e[q]) ? ((stop[q] == hp[q]) ? (int64_t)hv[q] : (int64_t)Ai[stop[q]]) : PANEL_INF;
                pc.cur_out[t] = stop[q];
                pc.nv_out[t] = nv2;
                if (nv2 < iend) S->more = 1;
            }
            const bool has = stop[q] > c[q];
            const unsigned long long m = __ballot(has);
            if (m) {                                            // wave-uniform
                int slot = 0;
                if (lane == 0) slot = atomicAdd(&S->nseg, (int)__builtin_popcountll(m));
                slot = __shfl(slot, 0, 64);
                if (has) {
                    slot += (int)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                    S->where[slot] = ((unsigned long long)c[q] << PANEL_T_BITS) | (unsigned long long)t;
                    S->off[slot] = (uint32_t)(stop[q] - c[q]);
                }
            }
        }
        GBA_TICK(tk1);
        __syncthreads();
        GBA_TICK(tk2);
        GBA_TOCK(S, 0, tk0, tk1); GBA_TOCK(S, 1, tk1, tk2);
        const int ns = S->nseg;
        const bool last = t0 + PANEL_STEP >= deg;
        if (ns > 0 && (ns > PANEL_SEGCAP - PANEL_STEP || last)) {
            panel_expand<PANEL_UNROLL, PANEL_PIPELINE>(A, pB0, ns, S, load, apply, take_turns);
            if (S->overflow) return;
        } else {
            __syncthreads();                                    // everyone has read nseg before the next append
        }
    }
}

__global__ void __launch_bounds__(256)
narrow_index_kernel(const int64_t *__restrict__ in, int32_t *__restrict__ out, int64_t n) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; idx < n; idx += stride) out[idx] = (int32_t)in[idx];
}

// room[j] = boundaries vector j can have at most (0 for vectors the panel kernel does not count)
__global__ void __launch_bounds__(256)
bounds_room_kernel(int64_t nvec, const int64_t *__restrict__ vflops, int64_t cvlen, int64_t panel_min_flops,
                   int64_t *__restrict__ room) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; idx <= nvec; idx += stride) {
        int64_t r = 0;
        if (idx < nvec && vflops[idx] > panel_min_flops) {
            const int64_t most = (vflops[idx] < cvlen) ? vflops[idx] : cvlen;
            r = most / NUM_PANEL_FIT + 1;
        }
        room[idx] = r;
    }
}

// proom[j] = room for the pattern of C(:,j) in units bounded by its multiply-adds and by `most` (0 for vectors
// the panel kernel does not count)
__global__ void __launch_bounds__(256)
pattern_room_kernel(int64_t nvec, const int64_t *__restrict__ vflops, int64_t most, int64_t panel_min_flops,
                    int64_t *__restrict__ proom) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; idx <= nvec; idx += stride) {
        int64_t r = 0;
        if (idx < nvec && vflops[idx] > panel_min_flops) r = (vflops[idx] < most) ? vflops[idx] : most;
        proom[idx] = r;
    }
}

#ifdef GBA_PANEL_TICKS
__device__ unsigned long long g_sym_ticks[8];      // stopwatch slots of the symbolic panel kernel, summed over workgroups
#endif

// The symbolic sweep also records, for the numeric sweep, where the vector's
// pattern crosses every multiple of NUM_PANEL_FIT entries: bounds[bstart[j] + m - 1]
// is one past the index of the (m * NUM_PANEL_FIT)-th entry of C(:,j).  Panels cut
// at those indices fill the numeric table to exactly one half.
// MASKED = false compiles the mask test away (the MaskView then costs no scalar registers either).
template <bool MASKED>
__global__ void __launch_bounds__(PANEL_THREADS)
sym_panel_kernel(int64_t nlist, const int64_t *__restrict__ list, SpView A, SpView B, MaskView M, int64_t cvlen,
                 int64_t maxdeg, int64_t *__restrict__ scratch, unsigned long long *__restrict__ next,
                 int64_t *__restrict__ cnt, const int64_t *__restrict__ bstart, int64_t *__restrict__ bounds,
                 const int64_t *__restrict__ pstart, uint2 *__restrict__ pwords, int64_t *__restrict__ wbounds,
                 const int64_t *__restrict__ vflops) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *bm = reinterpret_cast<uint32_t *>(smem);
    __shared__ PanelSegs Q;
    __shared__ int64_t row;
    __shared__ uint32_t fresh_total;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int w = tid; w < SYM_PANEL_LDS_WORDS; w += PANEL_THREADS) bm[w] = 0;
    if (tid == 0) { Q.nseg = 0; Q.overflow = 0; }
#ifdef GBA_PANEL_TICKS
    if (tid == 0) for (int q = 0; q < 8; q++) Q.tk[q] = 0;
    const unsigned long long tkk = wall_clock64();
#endif
    PanelCursors pc(scratch + (int64_t)blockIdx.x * PANEL_SCRATCH_ARRAYS * maxdeg, maxdeg);
    const int64_t npanels = (cvlen + SYM_PANEL_WIDTH - 1) / SYM_PANEL_WIDTH;
    while (true) {
        if (tid == 0) row = (int64_t)atomicAdd(next, 1ull);
        __syncthreads();
        const int64_t li = row;
        if (li >= nlist) break;
        const int64_t j = list[li];
        const int64_t pB0 = B.p[j], deg = B.p[j + 1] - pB0;
        int64_t *jb = bounds + bstart[j];
        uint2 *pw = pwords ? pwords + pstart[j] : nullptr;  // where the non-empty bitmap words of C(:,j) are written, in index order
        int64_t *jw = pwords ? wbounds + bstart[j] : nullptr;   // and where the word of every boundary stands in that list
        int64_t seen = 0, wseen = 0;                        // entries / non-empty words of C(:,j) below the current panel
        int64_t *__restrict__ my_scratch = scratch + (int64_t)blockIdx.x * PANEL_SCRATCH_ARRAYS * maxdeg;
#ifndef GBA_SYM_NO_SMALL
        // A vector that cannot reach the first boundary (fewer multiply-adds than a stretch has entries) needs its
        // count only, and the sweeps over the whole index range would cost it far more than its products: one walk
        // over all its products instead, the indices hashed into the words of the bitmap (open addressing, i + 1 as
        // the key, at most a quarter full).  No pattern words are recorded: its one stretch is hashed by the
        // numeric sweep as well.
        if (vflops && vflops[j] < (int64_t)NUM_PANEL_FIT && 2 * deg <= (int64_t)PANEL_SCRATCH_ARRAYS * maxdeg) {
            for (int64_t t = tid; t < deg; t += PANEL_THREADS) {
                const int64_t k = B.i[pB0 + t];
                my_scratch[t] = A.p[k]; my_scratch[deg + t] = A.p[k + 1];
            }
            if (tid == 0) { Q.more = 0; fresh_total = 0u; }
            __syncthreads();
            uint32_t fresh = 0;
            auto no_load = [](int32_t, int64_t, int64_t) { return 0; };
            auto insert_products =
                [&](const int32_t (&i)[SYM_PANEL_UNROLL], const int (&)[SYM_PANEL_UNROLL], const bool (&ok)[SYM_PANEL_UNROLL]) {
#pragma unroll
                    for (int u = 0; u < SYM_PANEL_UNROLL; u++) {
                        if (!ok[u]) continue;
                        if constexpr (MASKED) { if (!mask_allows(M, i[u], j)) continue; }
                        const uint32_t key = (uint32_t)i[u] + 1u;
                        uint32_t h = (key * 2654435761u) >> (32 - SYM_HASH_BITS);
                        while (true) {
                            const uint32_t old = atomicCAS(&bm[h], 0u, key);
                            if (old == 0u) { fresh++; break; }
                            if (old == key) break;
                            h = (h + 1u) & (uint32_t)(SYM_PANEL_WORDS - 1);
                        }
                    }
                };
            panel_walk_table<SYM_PANEL_UNROLL, SYM_PANEL_PIPELINE>(A, pB0, deg, 0, PANEL_INF, my_scratch, 2, 0, 1, &Q, no_load, insert_products, false);
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) fresh += __shfl_down(fresh, d, 64);
            if (lane == 0 && fresh) atomicAdd(&fresh_total, fresh);
            for (int w = tid; w < SYM_PANEL_WORDS; w += PANEL_THREADS) bm[w] = 0u;
            __syncthreads();
            if (tid == 0) {
                cnt[j] = (int64_t)fresh_total;
                if (jw) jw[0] = 0;                          // no word list: the numeric sweep hashes the vector
            }
            continue;
        }
#endif
        // Few neighbours: where every neighbour's list crosses the panel edges (multiples of the panel width) is
        // looked up once, by independent bisections, and the sweeps read their runs from that table
        // (panel_walk_table) instead of advancing cursors with the group search.
        const bool by_table = deg * 2 <= PANEL_THREADS && (npanels + 1) * deg <= (int64_t)PANEL_SCRATCH_ARRAYS * maxdeg
#ifdef GBA_PANEL_NO_SYM_TABLE
                              && false
#endif
                              ;
        GBA_TICK(tkt0);
        if (by_table) {
            const int32_t *__restrict__ Ai = A.i32;
            const int64_t cells = (npanels + 1) * deg;
            for (int64_t idx = tid; idx < cells; idx += PANEL_THREADS) {
                const int64_t m = (int64_t)((uint32_t)idx / (uint32_t)deg), t = idx - m * deg;
                const int64_t k = B.i[pB0 + t];
                int64_t a = A.p[k], b = A.p[k + 1];
                if (m == npanels) a = b;
                else if (m > 0) {
                    const int64_t v = m * SYM_PANEL_WIDTH;
                    while (a < b) { const int64_t mid = (a + b) >> 1; if ((int64_t)Ai[mid] < v) a = mid + 1; else b = mid; }
                }
                my_scratch[idx] = a;
            }
            __syncthreads();
        }
        GBA_TICK(tkt1);
        GBA_TOCK(&Q, 6, tkt0, tkt1);
        for (int64_t p = 0; p < npanels; p++) {
            const int64_t base = p * SYM_PANEL_WIDTH;
            const int64_t pend = (p == npanels - 1) ? PANEL_INF : base + SYM_PANEL_WIDTH;
            if (tid == 0) Q.more = 0;
            __syncthreads();
            auto no_load = [](int32_t, int64_t, int64_t) { return 0; };
            auto mark_products =
                [&](const int32_t (&i)[SYM_PANEL_UNROLL], const int (&)[SYM_PANEL_UNROLL], const bool (&ok)[SYM_PANEL_UNROLL]) {
#pragma unroll
                    for (int u = 0; u < SYM_PANEL_UNROLL; u++) {
                        if (!ok[u]) continue;
                        if constexpr (MASKED) { if (!mask_allows(M, i[u], j)) continue; }
                        const uint32_t o = (uint32_t)(i[u] - base);
                        atomicOr(&bm[(o >> 5) + (o >> 10)], 1u << (o & 31));     // (word o >> 5, in slice o >> 10)
                    }
                };
            if (by_table)
                panel_walk_table<SYM_PANEL_UNROLL, SYM_PANEL_PIPELINE>(A, pB0, deg, base, pend, my_scratch, (int)npanels + 1, (int)p, (int)p + 1,
                                                                        &Q, no_load, mark_products, false);
            else
                panel_walk<SYM_PANEL_UNROLL, SYM_PANEL_PIPELINE>(A, B, pB0, deg, 0, pend, PANEL_INF, p == 0, pc, &Q, no_load, mark_products, false);
            // count in index order: each thread owns SYM_WORDS_PER_THREAD consecutive words (its slice); the packed
            // sum carries the entries (low 21 bits: a panel has at most 2^20) and the non-empty words (above)
            GBA_TICK(tks0);
            const int w0 = tid * SYM_WORDS_PER_THREAD;
            uint32_t *mine_bm = bm + tid * SYM_SLICE_STRIDE;
            // first walk over the slice, last word first: the entries, and which words hold any (bit q of held = word q;
            // held = 2 * held + (x != 0) is one add-with-carry)
            uint32_t entries = 0, held = 0;
#pragma unroll 8
            for (int q = SYM_WORDS_PER_THREAD - 1; q >= 0; q--) {
                const uint32_t x = mine_bm[q];
                entries += (uint32_t)__builtin_popcount(x);
                held = held + held + (x != 0u ? 1u : 0u);
            }
            const uint64_t both = (uint64_t)entries + ((uint64_t)__builtin_popcount(held) << 21);
            const int mine = (int)(both & 0x1fffffu);
            uint64_t incl2 = both;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint64_t u = __shfl_up((unsigned long long)incl2, d, 64); if (lane >= d) incl2 += u; }
            if (lane == 63) Q.wsum2[wv] = incl2;
            __syncthreads();
            uint64_t woff2 = 0, total2 = 0;
#pragma unroll
            for (int q = 0; q < PANEL_THREADS / 64; q++) { const uint64_t w = Q.wsum2[q]; if (q < wv) woff2 += w; total2 += w; }
            const uint64_t excl2 = woff2 + incl2 - both;
            const int total = (int)(total2 & 0x1fffffu), nztotal = (int)(total2 >> 21);
            const int nzlo = (int)(excl2 >> 21);                // non-empty words of the panel before this thread's slice
            const int64_t lo = seen + (int64_t)(excl2 & 0x1fffffu);   // entries before this thread's slice
            const int64_t m = (int64_t)((uint32_t)lo / (uint32_t)NUM_PANEL_FIT) + 1;       // the next multiple to be crossed (cvlen < 2^31: the indices are read from a 32-bit copy)
            // second walk, over the non-empty words of the slice only: the boundary that falls into it, if any; the
            // pattern itself, for the numeric sweep that ranks instead of hashing -- the non-empty words as (word number, bits) pairs in index order,
            // a thread's words behind those of the thread before, so the lanes of a wave write neighbouring places;
            // and the slice is left clean (own slice only: no barrier needed)
            if (mine) {
                int r = (m * NUM_PANEL_FIT <= lo + mine) ? (int)(m * NUM_PANEL_FIT - lo) : 0;   // the r-th set bit of the slice, 1-based
                int nzb = 0;                                // non-empty words of the slice so far
                uint2 *out = pw ? pw + wseen + nzlo : nullptr;
                const uint32_t wid0 = (uint32_t)(base >> 5) + (uint32_t)w0;
                auto visit = [&](int q, uint32_t xq) {
                    mine_bm[q] = 0u;
                    if (r > 0) {
                        const int pcq = __builtin_popcount(xq);
                        if (r <= pcq) {
                            uint32_t y = xq;
                            for (int z = 1; z < r; z++) y &= y - 1;
                            jb[m - 1] = base + (int64_t)(w0 + q) * 32 + __builtin_ctz(y) + 1;
                            if (jw) jw[m - 1] = wseen + nzlo + nzb;   // where that word stands in the vector's list of words
                            r = 0;
                        } else {
                            r -= pcq;
                        }
                    }
                    if (out) out[nzb] = make_uint2(wid0 + (uint32_t)q, xq);
                    nzb++;
                };
                for (uint32_t t = held; t; t &= t - 1u) {    // the words that hold something, in index order
                    const int q = __builtin_ctz(t);
                    visit(q, mine_bm[q]);
                }
            }
            if (pw) wseen += nztotal;
            seen += total;
            const int more = Q.more;
            pc.swap();
            __syncthreads();
            GBA_TICK(tks1);
            GBA_TOCK(&Q, 5, tks0, tks1);
            if (!more) break;
        }
        if (tid == 0) {
            cnt[j] = seen;
            if (jw) jw[seen / NUM_PANEL_FIT] = wseen;       // (the slot after the last boundary) the length of the list
        }
    }
#ifdef GBA_PANEL_TICKS
    if (tid == 0) { Q.tk[7] = wall_clock64() - tkk; for (int q = 0; q < 8; q++) atomicAdd(&g_sym_ticks[q], Q.tk[q]); }
#endif
}

// A unit of work of the numeric panel kernel: the stretches [q0, q1) of C(:,j), where stretch q lies
// between the recorded boundaries q and q+1 (stretch 0 starts at index 0, the last one is open-ended) and
// every stretch but the last holds exactly NUM_PANEL_FIT entries -- so the output position of a task is
// known without any exchange: Cp[j] + q0 * NUM_PANEL_FIT.  Vectors heavier than PANEL_SPLIT_FLOPS are cut
// into several tasks so that one hub vector cannot keep a single workgroup busy long after the rest finished.
struct PanelTask { int64_t j; int32_t q0, q1; };
static constexpr int64_t PANEL_SPLIT_FLOPS = 4000000;          // default; GBA_PANEL_SPLIT_FLOPS overrides (tuning)
static int64_t panel_split_flops() {
    static const int64_t v = [] {
        const char *e = getenv("GBA_PANEL_SPLIT_FLOPS");
        const long long x = e ? atoll(e) : 0;
        return (int64_t)(x > 0 ? x : PANEL_SPLIT_FLOPS);
    }();
    return v;
}

__device__ inline int panel_task_count(int64_t j, const int64_t *__restrict__ Cp, const int64_t *__restrict__ vflops,
                                       const int64_t *__restrict__ bstart, int64_t split, int64_t *nstretch) {
    const bool has = bstart && bstart[j + 1] > bstart[j];
    const int64_t ns = has ? (Cp[j + 1] - Cp[j]) / NUM_PANEL_FIT + 1 : 1;
    int64_t want = (vflops[j] + split - 1) / split;
    if (want < 1) want = 1;
    *nstretch = ns;
    return (int)((want < ns) ? want : ns);
}

// count[li] = tasks of the li-th listed vector (count has nlist + 1 entries, the last one zeroed)
__global__ void __launch_bounds__(256)
panel_task_count_kernel(int64_t nlist, const int64_t *__restrict__ list, const int64_t *__restrict__ Cp,
                        const int64_t *__restrict__ vflops, const int64_t *__restrict__ bstart, int64_t split,
                        int64_t *__restrict__ count) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; idx <= nlist; idx += stride) {
        int64_t ns;
        count[idx] = (idx < nlist) ? panel_task_count(list[idx], Cp, vflops, bstart, split, &ns) : 0;
    }
}

__global__ void __launch_bounds__(256)
panel_task_fill_kernel(int64_t nlist, const int64_t *__restrict__ list, const int64_t *__restrict__ Cp,
                       const int64_t *__restrict__ vflops, const int64_t *__restrict__ bstart, int64_t split,
                       const int64_t *__restrict__ first_task, PanelTask *__restrict__ tasks) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; idx < nlist; idx += stride) {
        const int64_t j = list[idx];
        int64_t ns;
        const int nt = panel_task_count(j, Cp, vflops, bstart, split, &ns);
        PanelTask *t = tasks + first_task[idx];
        for (int q = 0; q < nt; q++) {
            t[q].j = j;
            t[q].q0 = (int32_t)(ns * q / nt);
            t[q].q1 = (int32_t)(ns * (q + 1) / nt);
        }
    }
}

// Multiplicative (Fibonacci) hash, reduced to [0, NUM_PANEL_SLOTS) by a second multiply.  The order in which a panel's
// entries leave the table is free (C is returned jumbled), so the function need not be GB_HASHF.
__device__ inline uint32_t panel_hash(int32_t key) {
    return (uint32_t)(((uint64_t)((uint32_t)key * 0x9E3779B1u) * (uint64_t)NUM_PANEL_SLOTS) >> 32);
}
static_assert(NUM_PANEL_SLOTS % PANEL_THREADS == 0 && NUM_PANEL_SLOTS % 4 == 0, "whole buckets; the emit loop is wave-uniform");
static_assert(NUM_PANEL_FIT * 4 <= NUM_PANEL_SLOTS * 3, "a stretch fills the table to three quarters at most");

// Ranked panels.  When the symbolic sweep wrote the pattern of C(:,j) down -- the non-empty words of its bitmap
// as (word number, bits) pairs in index order, and where the word of every boundary stands in that list -- a
// stretch between two boundaries that spans at most RANK_WORDS words needs no hashing.  Its pairs are read,
// coalesced, ranked by a prefix sum of their popcounts, and laid into an LDS table with one entry per 32
// indices of the panel's range: the bits present and the rank of the first of them.  Entry e of the panel owns
// vals[e].  A product then costs one 8-byte LDS read, a popcount, the monoid's LDS atomic and a store of its
// index: no probing, no compare-and-swap, nothing divergent.  Products whose index is not in the pattern
// (masked out) find their bit clear and are dropped, so the mask is not consulted.  The panel leaves vals in
// index order: Cx and Ci are straight copies.  Wider (sparser) stretches are hashed as before; the two table
// layouts share the same LDS and are re-initialised when the kind of panel changes.
static constexpr int RANK_WORDS = 23 * PANEL_THREADS / 4;                               // 46 KiB of LDS
static constexpr int RANK_PPT = 6;                                    // pairs per thread: a ranked stretch has at most RANK_WORDS + 1
static_assert(RANK_PPT * PANEL_THREADS > RANK_WORDS, "all pairs of a ranked stretch are held at once");
static_assert((int64_t)RANK_WORDS * 32 <= PANEL_MAX_WIDTH, "ranked panel too wide for 32-bit product offsets");

template <class ACC> constexpr size_t num_panel_lds() {
    constexpr size_t hashed = (size_t)NUM_PANEL_SLOTS * (sizeof(ACC) + 4);
    constexpr size_t ranked = (size_t)NUM_PANEL_FIT * (sizeof(ACC) + 4) + (size_t)RANK_WORDS * 8;
    return hashed > ranked ? hashed : ranked;
}

template <class S, bool MASKED>
__global__ void __launch_bounds__(PANEL_THREADS)
num_panel_kernel(S sr, int64_t nlist, const PanelTask *__restrict__ tasks, SpView A, SpView B, MaskView M, int64_t cvlen,
                 int64_t maxdeg, int64_t scratch_stride, int64_t *__restrict__ scratch, unsigned long long *__restrict__ next,
                 const int64_t *__restrict__ Cp, int64_t *__restrict__ Ci, typename S::z_t *__restrict__ Cx,
                 const int64_t *__restrict__ bstart, const int64_t *__restrict__ bounds,
                 const int64_t *__restrict__ pstart, const uint2 *__restrict__ pwords, const int64_t *__restrict__ wbounds,
                 const int64_t *__restrict__ vflops) {
    using Z = typename S::z_t; using ACC = typename S::acc_t;
    extern __shared__ __align__(16) unsigned char smem[];
    ACC *vals = reinterpret_cast<ACC *>(smem);
    int32_t *keys = reinterpret_cast<int32_t *>(smem + (size_t)NUM_PANEL_SLOTS * sizeof(ACC));    // hashed layout
    int32_t *ridx = reinterpret_cast<int32_t *>(smem + (size_t)NUM_PANEL_FIT * sizeof(ACC));      // ranked layout: index of entry e
    uint2 *rk = reinterpret_cast<uint2 *>(smem + (size_t)NUM_PANEL_FIT * (sizeof(ACC) + 4));      //   and (bits, first rank) per 32 indices
    unsigned long long st_panels = 0, st_overflows = 0;     // next[1], next[2]: panels swept / redone (reporting)
    unsigned long long st_direct = 0, st_dent = 0, st_hent = 0;   // next[11..13]: direct or ranked panels, entries by kind of panel
    __shared__ PanelSegs Q;
    __shared__ int64_t row;
    __shared__ int out_cursor;
    const ACC ident = (ACC)sr.identity();
    const bool apat = sr.a_is_pattern(), bpat = sr.b_is_pattern();
    const int tid = threadIdx.x, lane64 = tid & 63;
    int64_t *__restrict__ my_scratch = scratch + (int64_t)blockIdx.x * scratch_stride;
    PanelCursors pc(my_scratch, maxdeg);
    for (int s = tid; s < NUM_PANEL_SLOTS; s += PANEL_THREADS) { keys[s] = HASH_EMPTY; vals[s] = ident; }
    bool rank_layout = false;                               // which of the two layouts the LDS holds, clean
    if (tid == 0) Q.nseg = 0;
#ifdef GBA_PANEL_TICKS
    if (tid == 0) for (int q = 0; q < 8; q++) Q.tk[q] = 0;
    const unsigned long long tkk = wall_clock64();
#endif
    while (true) {
        if (tid == 0) row = (int64_t)atomicAdd(next, 1ull);
        __syncthreads();
        const int64_t li = uniform_i64(row);
        if (li >= nlist) break;
        const PanelTask task = tasks[li];
        const int64_t j = task.j;
        const int64_t pB0 = B.p[j], deg = B.p[j + 1] - pB0;
        const int64_t pC0 = Cp[j];
        int64_t pC = pC0;
        const int64_t cj = Cp[j + 1] - pC;
        // Offsets into the products of a panel are 32 bits wide.  All products of a vector whose multiply-adds stay
        // below 2^31 fit whatever the panel's width; only a heavier vector keeps its hashed panels to PANEL_MAX_WIDTH
        // indices (a run inside a panel is no longer than the panel is wide).
#ifdef GBA_PANEL_CAPPED
        const int64_t max_width = PANEL_MAX_WIDTH;
#else
        const int64_t max_width = (vflops[j] < ((int64_t)1 << 31)) ? cvlen : PANEL_MAX_WIDTH;
#endif
        int64_t base = 0, iend = PANEL_INF;
        int64_t width = (cj > 0) ? (cvlen * NUM_PANEL_TARGET) / cj : cvlen;
        if (width > 4096) width = 4096;                 // the low indices are the dense end of a power-law vector
        if (width < 1) width = 1;
        bool first = true;
        // boundaries (and the pattern) recorded by the symbolic sweep, if it was the panel kernel that counted this vector
        const int64_t *jb = nullptr;
        const uint2 *__restrict__ pw = nullptr;
        const int64_t *jw = nullptr;
        int64_t nb = 0, mb = 0;
        if (bstart && bstart[j + 1] > bstart[j]) {
            jb = bounds + bstart[j]; nb = cj / NUM_PANEL_FIT;
            if (pwords) { pw = pwords + pstart[j]; jw = wbounds + bstart[j]; }
            if (task.q0 > 0) { base = jb[task.q0 - 1]; pC += (int64_t)task.q0 * NUM_PANEL_FIT; mb = task.q0; }
            if (task.q1 <= nb) iend = jb[task.q1 - 1];
        }
        const int64_t ibeg = base;
        int64_t shrink = 0;                             // halvings after an overflow (never needed with boundaries)
        // Boundaries on record and room for the table: where every neighbour's list crosses every boundary of the task is
        // looked up once, here, and the panels read their runs from that table (panel_walk_table) instead of
        // searching panel after panel.  Rows: boundary q0 (the task's first index) .. boundary q0 + nrows - 1 (its end).
        const int64_t q0 = task.q0;
        const int nrows = jb ? (int)(((task.q1 <= nb) ? task.q1 : nb + 1) - q0) + 1 : 0;
        const bool by_table = jb && deg <= GBA_TABLE_MAX_DEG && nrows >= 2 && (int64_t)nrows * deg <= scratch_stride
#ifdef GBA_PANEL_NO_TABLE
                              && false
#endif
                              ;
        if (by_table) {
            // first the two ends of every neighbour's range inside the task, then every boundary in between by a
            // bisection of its own inside that range: independent searches, all threads busy
            const int32_t *__restrict__ Ai = A.i32;
            int64_t *__restrict__ row_first = my_scratch, *__restrict__ row_last = my_scratch + (int64_t)(nrows - 1) * deg;
            const int64_t q_last = q0 + nrows - 1;
            for (int64_t t = tid; t < deg; t += PANEL_THREADS) {
                const int64_t k = B.i[pB0 + t];
                int64_t a = A.p[k], b = A.p[k + 1];
                if (q0 > 0) {
                    const int64_t v = jb[q0 - 1];
                    int64_t hi = b;
                    while (a < hi) { const int64_t mid = (a + hi) >> 1; if ((int64_t)Ai[mid] < v) a = mid + 1; else hi = mid; }
                }
                if (q_last <= nb) {
                    const int64_t v = jb[q_last - 1];
                    int64_t lo = a;
                    while (lo < b) { const int64_t mid = (lo + b) >> 1; if ((int64_t)Ai[mid] < v) lo = mid + 1; else b = mid; }
                }
                row_first[t] = a; row_last[t] = b;
            }
            __syncthreads();
            const int64_t cells = (int64_t)(nrows - 2) * deg;               // rows 1 .. nrows - 2
            for (int64_t idx = tid; idx < cells; idx += PANEL_THREADS) {
                const int64_t m = (int64_t)((uint32_t)idx / (uint32_t)deg), t = idx - m * deg;     // (cells < 2^27)
                int64_t a = row_first[t], b = row_last[t];
                const int64_t v = jb[q0 + m];                              // boundary q0 + m + 1
                while (a < b) { const int64_t mid = (a + b) >> 1; if ((int64_t)Ai[mid] < v) a = mid + 1; else b = mid; }
                my_scratch[(m + 1) * deg + t] = a;
            }
            __syncthreads();
        }
        while (true) {
            if (jb) {
                while (mb < nb && jb[mb] <= base) mb++;
                width = ((mb < nb) ? jb[mb] : cvlen) - base;
                width >>= shrink;
                if (width < 1) width = 1;
            }
            // with the pattern on record a whole stretch that spans no more words than the rank table has is
            // ranked; without it a panel no wider than the hash table is addressed directly (slot = i - base);
            // everything else is hashed
            bool ranked = false;
            int64_t p0 = 0, np = 0;                     // the stretch's pairs: pw[p0 .. p0 + np)
            if (pw && shrink == 0 && base == ((mb > 0) ? jb[mb - 1] : 0)
                && ((base + width - 1) >> 5) - (base >> 5) < (int64_t)RANK_WORDS) {
                p0 = (mb > 0) ? jw[mb - 1] : 0;
                np = ((mb < nb) ? jw[mb] : jw[nb] - 1) - p0 + 1;
                ranked = np >= 1 && np <= (int64_t)RANK_PPT * PANEL_THREADS;
            }
            const bool direct = !ranked && width <= NUM_PANEL_SLOTS;
            int64_t lim = base + (ranked ? width : direct ? (int64_t)NUM_PANEL_SLOTS : ((width < max_width) ? width : max_width));
            if (lim > iend) lim = iend;
            lim = uniform_i64(lim);
            const int64_t pend = (lim >= cvlen) ? PANEL_INF : lim;
            const int32_t base32 = (int32_t)base;
            const uint32_t wbase32 = (uint32_t)(base >> 5);
            if (tid == 0) { Q.overflow = 0; Q.more = 0; out_cursor = 0; }
            GBA_TICK(tkp0);
            if (ranked != rank_layout) {                // uniform: the other layout takes over the shared part of the LDS
                if (ranked) {
                    for (int s = tid; s < RANK_WORDS; s += PANEL_THREADS) rk[s] = make_uint2(0u, 0u);
                } else {
                    for (int s = tid; s < NUM_PANEL_SLOTS; s += PANEL_THREADS) keys[s] = HASH_EMPTY;
                    for (int s = NUM_PANEL_FIT + tid; s < NUM_PANEL_SLOTS; s += PANEL_THREADS) vals[s] = ident;
                }
                rank_layout = ranked;
            }
            // a ranked panel's pairs: read, trimmed to [base, lim) at both ends, their popcounts summed per wave
            uint32_t pwid[RANK_PPT], pbit[RANK_PPT];
            uint32_t incl = 0, mysum = 0;
            if (ranked) {
                const uint2 *__restrict__ pp = pw + p0;
                const uint32_t widb = (uint32_t)(base >> 5), widl = (uint32_t)(lim >> 5);
                const uint32_t lowcut = ~((1u << (uint32_t)(base & 31)) - 1u), highcut = (1u << (uint32_t)(lim & 31)) - 1u;
#pragma unroll
                for (int q = 0; q < RANK_PPT; q++) {
                    const int el = tid * RANK_PPT + q;              // consecutive pairs per thread: one prefix sum ranks them
                    uint2 w = make_uint2(0u, 0u);
                    if (el < (int)np) w = pp[el];
                    if (w.x == widb) w.y &= lowcut;
                    if (w.x == widl) w.y &= highcut;                // (lim a multiple of 32: nothing of that word belongs here)
                    if (w.x < widb || w.x > widl) w.y = 0u;         // (the list starts at the word of the last entry before base)
                    pwid[q] = w.x - widb; pbit[q] = w.y;
                    mysum += (uint32_t)__builtin_popcount(w.y);
                }
                incl = mysum;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const uint32_t u = __shfl_up(incl, d, 64); if (lane64 >= d) incl += u; }
                if (lane64 == 63) Q.wsum[tid >> 6] = incl;
            }
            __syncthreads();
            int rtotal = 0;
            if (ranked) {
                // the table: (bits, rank of the word's first index) per 32 indices of the panel's range
                uint32_t woff = 0;
#pragma unroll
                for (int q = 0; q < PANEL_THREADS / 64; q++) { const uint32_t w = Q.wsum[q]; if (q < (tid >> 6)) woff += w; rtotal += (int)w; }
                uint32_t rank = woff + incl - mysum;
#pragma unroll
                for (int q = 0; q < RANK_PPT; q++) {
                    if (pbit[q]) rk[pwid[q]] = make_uint2(pbit[q], rank);
                    rank += (uint32_t)__builtin_popcount(pbit[q]);
                }
            }                                           // (the barrier after the search stands between this and the products)
            // (by_table) the rows of the table this panel starts and ends on, -1 where it does not stand on a boundary
            const int row_lo = (by_table && base == ((mb > 0) ? jb[mb - 1] : 0)) ? (int)(mb - q0) : -1;
            const int row_hi = (by_table && lim == ((mb < nb) ? jb[mb] : cvlen) && mb + 1 - q0 < nrows) ? (int)(mb + 1 - q0)
                             : (by_table && lim == iend && iend < PANEL_INF) ? nrows - 1 : -1;
            auto load_operands = [&](int32_t, int64_t pA, int64_t pB) -> PanelOperands<S> {
                    return fetch_operands(sr, A, B, apat, bpat, pA, pB);
                };
            auto apply_products = [&](const int32_t (&i)[NUM_PANEL_UNROLL], const PanelOperands<S> (&r)[NUM_PANEL_UNROLL], const bool (&ok)[NUM_PANEL_UNROLL]) {
                    Z t[NUM_PANEL_UNROLL];
#pragma unroll
                    for (int u = 0; u < NUM_PANEL_UNROLL; u++) t[u] = sr.mult(r[u].a, r[u].b, (int64_t)i[u], r[u].k, j);
                    if (ranked) {
                        uint2 w[NUM_PANEL_UNROLL];
#pragma unroll
                        for (int u = 0; u < NUM_PANEL_UNROLL; u++) {                   // all table reads of the window together
                            const uint32_t wd = ((uint32_t)i[u] >> 5) - wbase32;       // the table is aligned to whole words
                            w[u] = rk[ok[u] ? wd : 0u];
                        }
#pragma unroll
                        for (int u = 0; u < NUM_PANEL_UNROLL; u++) {
                            const uint32_t bit = 1u << ((uint32_t)i[u] & 31u);
                            if (ok[u] && (w[u].x & bit)) {
                                const uint32_t slot = w[u].y + (uint32_t)__builtin_popcount(w[u].x & (bit - 1u));
                                ridx[slot] = i[u];
                                sr.atomic(&vals[slot], t[u]);
                            }
                        }
                        return;
                    }
                    bool todo[NUM_PANEL_UNROLL];
#pragma unroll
                    for (int u = 0; u < NUM_PANEL_UNROLL; u++) {
                        todo[u] = ok[u];
                        if constexpr (MASKED) todo[u] = todo[u] && mask_allows(M, i[u], j);
                    }
                    if (direct) {
#pragma unroll
                        for (int u = 0; u < NUM_PANEL_UNROLL; u++) if (todo[u]) {
                            const uint32_t h = (uint32_t)(i[u] - base32);
                            keys[h] = (int32_t)i[u];
                            sr.atomic(&vals[h], t[u]);
                        }
                        return;
                    }
                    // open addressing in buckets of four slots: one 16-byte LDS read shows a whole bucket, so a
                    // product costs one or two dependent LDS operations instead of one per slot probed.  The
                    // slots of a bucket fill in order and are never emptied before the emit, so two lanes with
                    // the same index always meet in the same slot.
#pragma unroll
                    for (int u = 0; u < NUM_PANEL_UNROLL; u++) if (todo[u]) {
                        const int32_t key = (int32_t)i[u];
                        uint32_t hb = panel_hash(key) & ~3u;
                        int slot = -1;
                        for (int probe = 0; probe < PANEL_PROBE_LIMIT; probe++) {
                            const int4 kk = *reinterpret_cast<const int4 *>(&keys[hb]);
                            int pos = (kk.x == key) ? 0 : (kk.y == key) ? 1 : (kk.z == key) ? 2 : (kk.w == key) ? 3 : -1;
                            if (pos < 0) {
                                const int e = (kk.x == HASH_EMPTY) ? 0 : (kk.y == HASH_EMPTY) ? 1 : (kk.z == HASH_EMPTY) ? 2
                                            : (kk.w == HASH_EMPTY) ? 3 : -1;
                                if (e < 0) { hb = (hb + 4 < (uint32_t)NUM_PANEL_SLOTS) ? hb + 4 : 0u; continue; }     // bucket full: next one
                                const int32_t old = atomicCAS(&keys[hb + e], HASH_EMPTY, key);
                                if (old != HASH_EMPTY && old != key) continue;                      // lost the slot: look again
                                pos = e;
                            }
                            slot = (int)hb + pos;
                            break;
                        }
                        if (slot >= 0) sr.atomic(&vals[slot], t[u]); else Q.overflow = 1;
                    }
                };
            if (by_table)
                panel_walk_table<NUM_PANEL_UNROLL, NUM_PANEL_PIPELINE>(A, pB0, deg, base, pend, my_scratch, nrows, row_lo, row_hi, &Q,
                                                                        load_operands, apply_products, true);
            else
                panel_walk<NUM_PANEL_UNROLL, NUM_PANEL_PIPELINE>(A, B, pB0, deg, ibeg, pend, iend, first, pc, &Q,
                                                                  load_operands, apply_products, true);
            if (!ranked && Q.overflow) {                 // uniform: raised before the barrier that ended the walk
                width = (width > 1) ? (width >> 1) : 1;
                shrink++;
                st_overflows++;
                for (int s = tid; s < NUM_PANEL_SLOTS; s += PANEL_THREADS) { keys[s] = HASH_EMPTY; vals[s] = ident; }
                __syncthreads();
                continue;
            }
            GBA_TICK(tkp1);
            if (ranked) {
                // in index order: Cx and Ci are straight copies; leave the table clean
                Z *__restrict__ cx = Cx + pC;
                int64_t *__restrict__ ci = Ci + pC;
                const int nw = (int)(((lim - 1) >> 5) - (base >> 5)) + 1;
                for (int s = tid; s < rtotal; s += PANEL_THREADS) { cx[s] = (Z)vals[s]; ci[s] = (int64_t)ridx[s]; vals[s] = ident; }
                for (int s = tid; s < nw; s += PANEL_THREADS) rk[s] = make_uint2(0u, 0u);
            } else {
                // emit the panel (wave-aggregated positions, coalesced stores) and leave the table clean
                for (int s = tid; s < NUM_PANEL_SLOTS; s += PANEL_THREADS) {
                    const int32_t key = keys[s];
                    const bool occ = (key != HASH_EMPTY);
                    const unsigned long long m = __ballot(occ);
                    int wbase = 0;
                    if (lane64 == 0 && m) wbase = atomicAdd(&out_cursor, __builtin_popcountll(m));
                    wbase = __shfl(wbase, 0, 64);
                    if (occ) {
                        const int64_t q = pC + wbase + __builtin_popcountll(m & ((1ull << lane64) - 1));
                        Ci[q] = key;
                        Cx[q] = (Z)vals[s];
                        keys[s] = HASH_EMPTY;
                        vals[s] = ident;
                    }
                }
            }
            __syncthreads();
            GBA_TICK(tkp2);
            GBA_TOCK(&Q, 5, tkp1, tkp2); GBA_TOCK(&Q, 6, tkp0, tkp2);
            const int d = ranked ? rtotal : out_cursor;
            const int more = Q.more;
            pC = uniform_i64(pC + d);
            pc.swap();
            first = false;
            base = lim;
            shrink = 0;
            st_panels++;
            if (direct || ranked) { st_direct++; st_dent += (unsigned long long)d; } else st_hent += (unsigned long long)d;
            if (d == 0) width = (width < cvlen) ? (width << 2) : width;
            else {
                const int64_t covered = direct ? (int64_t)NUM_PANEL_SLOTS : width;
                int64_t w2 = (int64_t)((double)covered * (double)NUM_PANEL_TARGET / (double)d);
                if (w2 > 4 * covered) w2 = 4 * covered;
                if (w2 < 1) w2 = 1;
                width = w2;
            }
            __syncthreads();
            if (!more) break;
        }
    }
    if (tid == 0) {
        atomicAdd(&next[1], st_panels); atomicAdd(&next[2], st_overflows);
        atomicAdd(&next[11], st_direct); atomicAdd(&next[12], st_dent); atomicAdd(&next[13], st_hent);
    }
#ifdef GBA_PANEL_TICKS
    if (tid == 0) { Q.tk[7] = wall_clock64() - tkk; for (int q = 0; q < 8; q++) atomicAdd(&next[3 + q], Q.tk[q]); }
#endif
}

// ------------------------------------------------------------------ driver --
static int pick_lpa(const gba_matrix *A) {
    // threads that share one A(:,k): about its average length, a power of two in [4,64]
    const double avg = (A->nvec > 0) ? (double)A->nvals / (double)A->nvec : 1.0;
    int lpa = 4;
    while (lpa < 64 && (double)lpa < avg) lpa <<= 1;
    return lpa;
}

static gba_info make_mask_view(const gba_matrix *M, bool comp, bool structural, MaskView *mv) {
    MaskView m{};
    if (!M) { m.kind = 0; *mv = m; return GBA_SUCCESS; }
    if (is_hyper(M)) GBA_FAIL(GBA_NOT_IMPLEMENTED, "hypersparse mask");
    if (is_sparse(M) && M->jumbled) GBA_FAIL(GBA_NOT_IMPLEMENTED, "jumbled sparse mask (wait on it first)");
    if (M->nzombies) GBA_FAIL(GBA_INVALID_OBJECT, "mask has zombies (wait on it first)");
    m.kind = is_sparse(M) ? 1 : (is_bitmap(M) ? 2 : 3);
    m.comp = comp; m.structural = structural; m.iso = M->iso;
    m.msize = (int)type_size(M->type);
    if (!structural && (m.msize == 0 || !M->x)) GBA_FAIL(GBA_INVALID_OBJECT, "valued mask without values");
    m.vlen = M->vlen;
    m.p = M->p; m.i = M->i; m.b = M->b; m.x = reinterpret_cast<const uint8_t *>(M->x);
    *mv = m;
    return GBA_SUCCESS;
}

template <class K> static gba_info set_lds(K kernel, size_t bytes) {
    if (bytes > 48 * 1024) GBA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return GBA_SUCCESS;
}

// HIP events at phase boundaries on the library stream
struct PhaseClock {
    static constexpr int MAXEV = 12;
    hipEvent_t ev[MAXEV];
    int n = 0;
    bool on;
    explicit PhaseClock(bool enable) : on(enable) { if (on) for (int k = 0; k < MAXEV; k++) hipEventCreate(&ev[k]); }
    ~PhaseClock() { if (on) for (int k = 0; k < MAXEV; k++) hipEventDestroy(ev[k]); }
    int mark() { if (!on || n >= MAXEV) return -1; hipEventRecord(ev[n], ctx().stream); return n++; }
    double ms(int a, int b) {
        if (!on || a < 0 || b < 0) return 0.0;
        hipEventSynchronize(ev[b]);
        float f = 0.f; hipEventElapsedTime(&f, ev[a], ev[b]); return (double)f;
    }
};

struct Saxpy3Plan {
    SpView A, B; MaskView M; int lpa; int64_t cvlen, nvec;
    Bins sym, num;
    int64_t *Cp, *Ci; void *Cx;
    PhaseClock *clock = nullptr;
    int ev_num[5] = {-1, -1, -1, -1, -1};   // before wave, blk1, blk2, global, after global
    bool use_panels = false;                // A sorted and the product large enough: panel kernels for the heavy vectors
    int64_t maxdeg = 0;                     // longest vector of B
    const int64_t *vflops = nullptr;        // per-vector flop count (device)
    // team vectors: ids (host), their workspaces one after the other (device)
    std::vector<int64_t> team;
    void *team_Hx = nullptr; uint64_t *team_Hf = nullptr;
    const int64_t *bstart = nullptr;        // panel boundaries of vector j: bounds[bstart[j] .. bstart[j+1])
    int64_t *bounds = nullptr;
    const int64_t *pstart = nullptr;        // pattern of vector j, written by the symbolic sweep: the non-empty words of its
    uint2 *pwords = nullptr;                //   bitmap, (word number, bits), in pwords[pstart[j] ..), in index order
    int64_t *wbounds = nullptr;             //   and where the word of every boundary stands in that list (laid out like bounds)
    unsigned long long panel_stats[14] = {0};        // tasks pulled, panels swept, panels redone, stopwatch slots [3..10], direct panels, entries of direct / hashed panels
};

template <class S>
static gba_info run_team_products(const S &sr, Saxpy3Plan &P, int64_t *cnt) {
    using ACC = typename S::acc_t;
    hipStream_t st = ctx().stream;
    const int64_t nf = (P.cvlen + 63) >> 6;
    const int64_t nteam = (int64_t)P.team.size();
    ACC *Hx = reinterpret_cast<ACC *>(P.team_Hx);
    const int64_t nacc = nteam * P.cvlen;
    int64_t fb = ceil_div(nacc, 256); if (fb > 8192) fb = 8192;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(fill_acc_kernel<ACC>), dim3((unsigned)fb), dim3(256), 0, st, Hx, nacc, (ACC)sr.identity());
    GBA_HIP(hipMemsetAsync(P.team_Hf, 0, (size_t)nteam * nf * 8, st));
    DevBuf nlong, longq;
    GBA_OK(nlong.alloc((size_t)nteam * 8));
    GBA_OK(longq.alloc((size_t)(P.maxdeg > 0 ? P.maxdeg : 1) * 8));
    GBA_HIP(hipMemsetAsync(nlong.p, 0, (size_t)nteam * 8, st));
    const unsigned grid = (unsigned)(ctx().num_cu * 8);
    int64_t cb = ceil_div(nf, 256 * 8); if (cb > 512) cb = 512; if (cb < 1) cb = 1;
    for (int64_t q = 0; q < nteam; q++) {
        const int64_t j = P.team[(size_t)q];
        ACC *hx = Hx + q * P.cvlen; uint64_t *hf = P.team_Hf + q * nf;
        unsigned long long *nl = nlong.as<unsigned long long>() + q;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(team_product_kernel<S>), dim3(grid), dim3(TEAM_THREADS), 0, st, sr, j, P.A, P.B, P.M,
                           hx, hf, nl, longq.as<int64_t>());
        hipLaunchKernelGGL(HIP_KERNEL_NAME(team_long_kernel<S>), dim3(grid), dim3(TEAM_THREADS), 0, st, sr, j, P.A, P.B, P.M,
                           hx, hf, nl, longq.as<int64_t>());
        hipLaunchKernelGGL(team_count_kernel, dim3((unsigned)cb), dim3(256), 0, st, nf, hf,
                           reinterpret_cast<unsigned long long *>(cnt + j));
    }
    GBA_HIP(hipGetLastError());
    GBA_HIP(hipStreamSynchronize(st));          // longq is released on return
    return GBA_SUCCESS;
}

template <class S>
static gba_info run_team_gather(const S &sr, Saxpy3Plan &P) {
    using Z = typename S::z_t; using ACC = typename S::acc_t;
    hipStream_t st = ctx().stream;
    const int64_t nf = (P.cvlen + 63) >> 6;
    const int64_t nteam = (int64_t)P.team.size();
    DevBuf cursor, cp;
    GBA_OK(cursor.alloc((size_t)nteam * 8));
    GBA_HIP(hipMemsetAsync(cursor.p, 0, (size_t)nteam * 8, st));
    int64_t gb = ceil_div(nf, 256); if (gb > 4096) gb = 4096;
    for (int64_t q = 0; q < nteam; q++) {
        const int64_t j = P.team[(size_t)q];
        hipLaunchKernelGGL(HIP_KERNEL_NAME(team_gather_kernel<S>), dim3((unsigned)gb), dim3(256), 0, st, nf,
                           P.team_Hf + q * nf, reinterpret_cast<const ACC *>(P.team_Hx) + q * P.cvlen,
                           cursor.as<unsigned long long>() + q, P.Cp, j, P.Ci, reinterpret_cast<Z *>(P.Cx));
    }
    GBA_HIP(hipGetLastError());
    GBA_HIP(hipStreamSynchronize(st));
    return GBA_SUCCESS;
}

template <class S>
static gba_info run_numeric(const S &sr, Saxpy3Plan &P) {
    using Z = typename S::z_t; using ACC = typename S::acc_t;
    hipStream_t st = ctx().stream;
    Z *Cx = reinterpret_cast<Z *>(P.Cx);
    const Bins &nb = P.num;
    auto mark = [&](int slot) { if (P.clock) P.ev_num[slot] = P.clock->mark(); };
    mark(0);
    if (nb.count[BIN_WAVE] > 0) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(num_wave_kernel<S>), dim3((unsigned)ceil_div(nb.count[BIN_WAVE], 4)), dim3(256), 0, st,
                           sr, nb.count[BIN_WAVE], nb.list(BIN_WAVE), P.A, P.B, P.M, P.lpa, P.Cp, P.Ci, Cx);
    }
    mark(1);
    if (nb.count[BIN_BLK1] > 0) {
        auto k = num_block_kernel<S, NUM_BLK1_SLOTS, NUM_BLK1_THREADS>;
        const size_t lds = (size_t)NUM_BLK1_SLOTS * (sizeof(ACC) + 4);
        GBA_OK(set_lds(k, lds));
        hipLaunchKernelGGL(k, dim3((unsigned)nb.count[BIN_BLK1]), dim3(NUM_BLK1_THREADS), lds, st,
                           sr, nb.count[BIN_BLK1], nb.list(BIN_BLK1), P.A, P.B, P.M, P.lpa, P.Cp, P.Ci, Cx,
                           (const int64_t *)nullptr, (int64_t *)nullptr);
    }
    mark(2);
    if (nb.count[BIN_BLK2] > 0) {
        auto k = num_block_kernel<S, NUM_BLK2_SLOTS, NUM_BLK2_THREADS>;
        const size_t lds = (size_t)NUM_BLK2_SLOTS * (sizeof(ACC) + 4);
        GBA_OK(set_lds(k, lds));
        hipLaunchKernelGGL(k, dim3((unsigned)nb.count[BIN_BLK2]), dim3(NUM_BLK2_THREADS), lds, st,
                           sr, nb.count[BIN_BLK2], nb.list(BIN_BLK2), P.A, P.B, P.M, P.lpa, P.Cp, P.Ci, Cx,
                           (const int64_t *)nullptr, (int64_t *)nullptr);
    }
    GBA_HIP(hipGetLastError());
    mark(3);
    if (nb.count[BIN_GLOBAL] > 0 && P.use_panels && P.maxdeg < PANEL_MAX_DEG) {
        int64_t nblocks = (int64_t)ctx().num_cu * PANEL_WG_PER_CU;
        if (nblocks > nb.count[BIN_GLOBAL]) nblocks = nb.count[BIN_GLOBAL];
        // tasks: heaviest first, hub vectors cut at recorded boundaries
        const int64_t nl = nb.count[BIN_GLOBAL];
        DevBuf order, tcount, tasks;
        GBA_OK(order_by_work_desc(nl, nb.list(BIN_GLOBAL), P.vflops, &order));
        GBA_OK(tcount.alloc((size_t)(nl + 1) * 8));
        int64_t tb = ceil_div(nl + 1, 256); if (tb > 4096) tb = 4096;
        hipLaunchKernelGGL(panel_task_count_kernel, dim3((unsigned)tb), dim3(256), 0, st, nl, order.as<int64_t>(), P.Cp,
                           P.vflops, P.bstart, panel_split_flops(), tcount.as<int64_t>());
        GBA_HIP(hipGetLastError());
        GBA_OK(dev_cumsum_i64(tcount.as<int64_t>(), nl, st));
        int64_t ntasks = 0;
        GBA_OK(dev_read_i64(tcount.as<int64_t>() + nl, &ntasks));
        GBA_OK(tasks.alloc((size_t)(ntasks > 0 ? ntasks : 1) * sizeof(PanelTask)));
        hipLaunchKernelGGL(panel_task_fill_kernel, dim3((unsigned)tb), dim3(256), 0, st, nl, order.as<int64_t>(), P.Cp,
                           P.vflops, P.bstart, panel_split_flops(), tcount.as<int64_t>(), tasks.as<PanelTask>());
        GBA_HIP(hipGetLastError());
        if (nblocks > ntasks) nblocks = ntasks;
        DevBuf scratch, next;
        // per workgroup: the cursor arrays of panel_walk, or the boundary table of panel_walk_table (at least 2^20 cells)
        int64_t scratch_stride = (int64_t)PANEL_SCRATCH_ARRAYS * (P.maxdeg > 0 ? P.maxdeg : 1);
        if (scratch_stride < ((int64_t)1 << 20)) scratch_stride = (int64_t)1 << 20;
        GBA_OK(scratch.alloc((size_t)nblocks * (size_t)scratch_stride * 8));
        GBA_OK(next.alloc(128));
        GBA_HIP(hipMemsetAsync(next.p, 0, 128, st));
        auto k = (P.M.kind == 0) ? num_panel_kernel<S, false> : num_panel_kernel<S, true>;
        const size_t lds = num_panel_lds<ACC>();
        GBA_OK(set_lds(k, lds));
        hipLaunchKernelGGL(k, dim3((unsigned)nblocks), dim3(PANEL_THREADS), lds, st,
                           sr, ntasks, tasks.as<PanelTask>(), P.A, P.B, P.M, P.cvlen, P.maxdeg, scratch_stride,
                           scratch.as<int64_t>(), next.as<unsigned long long>(), P.Cp, P.Ci, Cx, P.bstart, P.bounds,
                           P.pstart, P.pwords, P.wbounds, P.vflops);
        GBA_HIP(hipGetLastError());
        mark(4);
        GBA_HIP(hipMemcpyAsync(P.panel_stats, next.p, 112, hipMemcpyDeviceToHost, st));
        GBA_HIP(hipStreamSynchronize(st));      // scratch is released on return
        if (getenv("GBA_PANEL_PROF"))
            fprintf(stderr, "[gba] numeric panels %llu (ranked or direct %llu), entries from those %llu, from hashed panels %llu\n",
                    P.panel_stats[1], P.panel_stats[11], P.panel_stats[12], P.panel_stats[13]);
    } else if (nb.count[BIN_GLOBAL] > 0) {
        const int64_t nf = (P.cvlen + 63) >> 6, ns = (nf + 63) >> 6;
        const size_t per_block = (size_t)P.cvlen * sizeof(ACC) + (size_t)(nf + ns) * 8;
        size_t free_b = 0, total_b = 0;
        GBA_HIP(hipMemGetInfo(&free_b, &total_b));
        size_t idle_b = 0;
        gba_pool_stats(&idle_b, nullptr);               // idle pool blocks are released on demand
        free_b += idle_b;
        int64_t nblocks = 2 * (int64_t)ctx().num_cu;
        if (nblocks > nb.count[BIN_GLOBAL]) nblocks = nb.count[BIN_GLOBAL];
        const int64_t fit = (int64_t)((free_b / 2) / per_block);
        if (fit < 1) GBA_FAIL(GBA_OUT_OF_MEMORY, "no room for one Gustavson workspace of %zu bytes", per_block);
        if (nblocks > fit) nblocks = fit;
        DevBuf Hx, Hf, Hs, next;
        GBA_OK(Hx.alloc((size_t)nblocks * P.cvlen * sizeof(ACC)));
        GBA_OK(Hf.alloc((size_t)nblocks * nf * 8));
        GBA_OK(Hs.alloc((size_t)nblocks * ns * 8));
        GBA_OK(next.alloc(8));
        GBA_HIP(hipMemsetAsync(Hf.p, 0, Hf.bytes, st));
        GBA_HIP(hipMemsetAsync(Hs.p, 0, Hs.bytes, st));
        GBA_HIP(hipMemsetAsync(next.p, 0, 8, st));
        const int64_t nacc = nblocks * P.cvlen;
        int64_t fb = ceil_div(nacc, 256); if (fb > 8192) fb = 8192;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(fill_acc_kernel<ACC>), dim3((unsigned)fb), dim3(256), 0, st,
                           Hx.as<ACC>(), nacc, (ACC)sr.identity());
        hipLaunchKernelGGL(HIP_KERNEL_NAME(num_global_kernel<S>), dim3((unsigned)nblocks), dim3(GLOBAL_THREADS), 0, st,
                           sr, nb.count[BIN_GLOBAL], nb.list(BIN_GLOBAL), P.A, P.B, P.M, P.lpa, P.cvlen,
                           Hx.as<ACC>(), Hf.as<uint64_t>(), Hs.as<uint64_t>(), next.as<unsigned long long>(),
                           P.Cp, P.Ci, Cx);
        GBA_HIP(hipGetLastError());
        mark(4);
        GBA_HIP(hipStreamSynchronize(st));      // workspaces are released on return
    } else {
        mark(4);
    }
    return GBA_SUCCESS;
}

static gba_info run_symbolic(Saxpy3Plan &P, int64_t *cnt) {
    hipStream_t st = ctx().stream;
    const Bins &sb = P.sym;
    if (sb.count[BIN_WAVE] > 0) {
        hipLaunchKernelGGL(sym_wave_kernel, dim3((unsigned)ceil_div(sb.count[BIN_WAVE], 4)), dim3(256), 0, st,
                           sb.count[BIN_WAVE], sb.list(BIN_WAVE), P.A, P.B, P.M, P.lpa, cnt);
    }
    if (sb.count[BIN_BLK1] > 0) {
        auto k = sym_block_kernel<SYM_BLK1_SLOTS, SYM_BLK1_THREADS>;
        const size_t lds = (size_t)SYM_BLK1_SLOTS * 4;
        hipLaunchKernelGGL(k, dim3((unsigned)sb.count[BIN_BLK1]), dim3(SYM_BLK1_THREADS), lds, st,
                           sb.count[BIN_BLK1], sb.list(BIN_BLK1), P.A, P.B, P.M, P.lpa, cnt);
    }
    if (sb.count[BIN_BLK2] > 0) {
        auto k = sym_block_kernel<SYM_BLK2_SLOTS, SYM_BLK2_THREADS>;
        const size_t lds = (size_t)SYM_BLK2_SLOTS * 4;
        GBA_OK(set_lds(k, lds));
        hipLaunchKernelGGL(k, dim3((unsigned)sb.count[BIN_BLK2]), dim3(SYM_BLK2_THREADS), lds, st,
                           sb.count[BIN_BLK2], sb.list(BIN_BLK2), P.A, P.B, P.M, P.lpa, cnt);
    }
    GBA_HIP(hipGetLastError());
    if (sb.count[BIN_GLOBAL] > 0 && P.use_panels && P.maxdeg < PANEL_MAX_DEG) {
        int64_t nblocks = (int64_t)ctx().num_cu * PANEL_WG_PER_CU;
        if (nblocks > sb.count[BIN_GLOBAL]) nblocks = sb.count[BIN_GLOBAL];
        DevBuf scratch, next;
        GBA_OK(scratch.alloc((size_t)nblocks * PANEL_SCRATCH_ARRAYS * (size_t)(P.maxdeg > 0 ? P.maxdeg : 1) * 8));
        GBA_OK(next.alloc(8));
        GBA_HIP(hipMemsetAsync(next.p, 0, 8, st));
        const size_t lds = (size_t)SYM_PANEL_LDS_WORDS * 4;
        auto k = (P.M.kind == 0) ? sym_panel_kernel<false> : sym_panel_kernel<true>;
        GBA_OK(set_lds(k, lds));
        DevBuf order;
        GBA_OK(order_by_work_desc(sb.count[BIN_GLOBAL], sb.list(BIN_GLOBAL), P.vflops, &order));
        hipLaunchKernelGGL(k, dim3((unsigned)nblocks), dim3(PANEL_THREADS), lds, st,
                           sb.count[BIN_GLOBAL], order.as<int64_t>(), P.A, P.B, P.M, P.cvlen, P.maxdeg,
                           scratch.as<int64_t>(), next.as<unsigned long long>(), cnt, P.bstart, P.bounds, P.pstart, P.pwords, P.wbounds,
                           P.vflops);
        GBA_HIP(hipGetLastError());
        GBA_HIP(hipStreamSynchronize(st));
#ifdef GBA_PANEL_TICKS
        if (getenv("GBA_PANEL_PROF")) {
            unsigned long long tk[8], zero[8] = {0};
            GBA_HIP(hipMemcpyFromSymbol(tk, HIP_SYMBOL(g_sym_ticks), sizeof(tk)));
            GBA_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_sym_ticks), zero, sizeof(zero)));
            fprintf(stderr, "[gba] symbolic panel kernel, vectors %lld; 10 ns ticks summed over workgroups: search %llu (+wait %llu) prefix %llu "
                    "products %llu (+wait %llu) count passes %llu table %llu kernel %llu\n", (long long)sb.count[BIN_GLOBAL], tk[0], tk[1], tk[2], tk[3], tk[4], tk[5], tk[6], tk[7]);
        }
#endif
    } else if (sb.count[BIN_GLOBAL] > 0) {
        const int64_t nf = (P.cvlen + 63) >> 6, ns = (nf + 63) >> 6;
        int64_t nblocks = 2 * (int64_t)ctx().num_cu;
        if (nblocks > sb.count[BIN_GLOBAL]) nblocks = sb.count[BIN_GLOBAL];
        DevBuf Hf, Hs, next;
        GBA_OK(Hf.alloc((size_t)nblocks * nf * 8));
        GBA_OK(Hs.alloc((size_t)nblocks * ns * 8));
        GBA_OK(next.alloc(8));
        GBA_HIP(hipMemsetAsync(Hf.p, 0, Hf.bytes, st));
        GBA_HIP(hipMemsetAsync(Hs.p, 0, Hs.bytes, st));
        GBA_HIP(hipMemsetAsync(next.p, 0, 8, st));
        hipLaunchKernelGGL(sym_global_kernel, dim3((unsigned)nblocks), dim3(GLOBAL_THREADS), 0, st,
                           sb.count[BIN_GLOBAL], sb.list(BIN_GLOBAL), P.A, P.B, P.M, P.lpa, P.cvlen,
                           Hf.as<uint64_t>(), Hs.as<uint64_t>(), next.as<unsigned long long>(), cnt);
        GBA_HIP(hipGetLastError());
        GBA_HIP(hipStreamSynchronize(st));
    }
    return GBA_SUCCESS;
}

static gba_info saxpy3_long_vectors(gba_matrix *C, const gba_matrix *M, bool Mask_comp, bool Mask_struct,
                                    const gba_matrix *A, const gba_matrix *B, const Canon &c,
                                    int axb_method, int do_sort, bool *mask_applied, gba_report *rep, bool may_discard_mask);

gba_info saxpy3(gba_matrix *C, const gba_matrix *M, bool Mask_comp, bool Mask_struct,
                const gba_matrix *A, const gba_matrix *B, const Canon &c,
                int axb_method, int do_sort, bool *mask_applied, gba_report *rep, bool may_discard_mask) {
    (void)axb_method;
    GBA_OK(require_init());
    if (!is_sparse(A) || !is_sparse(B))
        GBA_FAIL(GBA_NOT_IMPLEMENTED, "saxpy3 on device needs sparse A and B (got sparsity %d, %d)", A->sparsity, B->sparsity);
    if (A->vdim != B->vlen) GBA_FAIL(GBA_DIMENSION_MISMATCH, "saxpy3: A->vdim %ld != B->vlen %ld", (long)A->vdim, (long)B->vlen);
    if (A->nzombies || B->nzombies) GBA_FAIL(GBA_INVALID_OBJECT, "saxpy3: operands have zombies");
    if (A->vlen >= (int64_t)0x7fffffff)                    // the kernels keep indices in 32 bits: see below
        return saxpy3_long_vectors(C, M, Mask_comp, Mask_struct, A, B, c, axb_method, do_sort, mask_applied, rep, may_discard_mask);
    if (M && (M->vlen != A->vlen || M->vdim != B->vdim)) GBA_FAIL(GBA_DIMENSION_MISMATCH, "saxpy3: mask dimensions");
    hipStream_t st = ctx().stream;
    DevBuf sorted_i, sorted_x;
    gba_matrix A_sorted;
    Saxpy3Plan P;
    P.A = SpView{A->p, A->i, A->x, A->iso};
    P.B = SpView{B->p, B->i, B->x, B->iso};
    GBA_OK(make_mask_view(M, Mask_comp, Mask_struct, &P.M));
    P.lpa = pick_lpa(A);
    P.cvlen = A->vlen;
    P.nvec = B->nvec;
    const int64_t nvec = P.nvec;

    PhaseClock clock(rep != nullptr);
    P.clock = rep ? &clock : nullptr;
    const int e0 = clock.mark();

    // phase 0: flop count per vector (kept un-summed for binning).  The un-masked count is an upper
    // bound on nnz C(:,j) whatever the mask does, which is what sizes the hash tables.
    DevBuf vflops; GBA_OK(vflops.alloc((size_t)(nvec + 1) * 8));
    GBA_OK(saxpy3_flopcount(nullptr, vflops.as<int64_t>(), nullptr, false, A, B));
    // A sparse mask with far more entries than the product has multiply-adds costs more to scatter than it saves:
    // the caller then multiplies without it and applies it in the write-back.  The reference's rule
    // (Source/GB_AxB_saxpy3_slice_balanced.c:22, 309-403): discard when flops < 0.01 * (entries of M in the vectors
    // that B fills).
    // A mask that is bitmap, full, or sparse with every entry present is always applied (slice_balanced.c:319), and so
    // is one too small for the rule to ever fire; neither costs the count below and its trip to the host.
    const bool m_as_if_full = M && M->nvals == M->vlen * M->vdim;
    if (may_discard_mask && M && is_sparse(M) && !m_as_if_full && M->nvals >= 100 && nvec > 0) {
        DevBuf w; GBA_OK(w.alloc(16));
        GBA_HIP(hipMemsetAsync(w.p, 0, 16, st));
        int64_t blocks = ceil_div(nvec, 256); if (blocks > 512) blocks = 512;
        hipLaunchKernelGGL(mask_work_kernel, dim3((unsigned)blocks), dim3(256), 0, st, nvec, B->p, M->p, vflops.as<int64_t>(),
                           w.as<unsigned long long>());
        unsigned long long hw[2] = {0, 0};
        GBA_HIP(hipMemcpyAsync(hw, w.p, 16, hipMemcpyDeviceToHost, st));
        GBA_HIP(hipStreamSynchronize(st));
        if ((double)hw[0] < 0.01 * (double)hw[1]) {
            if (mask_applied) *mask_applied = false;
            return GBA_NO_VALUE;
        }
    }
    // A few light vectors (the late levels of a traversal): one pass.  Every vector is accumulated in an LDS table
    // and emitted into a slab of its own, the counts come back in one read, the slabs are packed into C -- two
    // round trips to the host instead of the binning, counting and sizing of the two-pass road.
    if (nvec > 0 && nvec <= SMALL_NVEC && B->nvals <= SMALL_BNZ && !getenv("GBA_NO_ONE_PASS")) {
        const size_t zsize = type_size(c.ztype);
        DevBuf scnt, slab_i, slab_x;
        GBA_OK(scnt.alloc((size_t)nvec * 8));
        GBA_OK(slab_i.alloc((size_t)nvec * SMALL_ROOM * 8));
        GBA_OK(slab_x.alloc((size_t)nvec * SMALL_ROOM * zsize));
        GBA_OK(with_semiring(c, [&](auto sr) -> gba_info {
            using S = decltype(sr); using Z = typename S::z_t; using ACC = typename S::acc_t;
            auto k = num_block_kernel<S, NUM_BLK2_SLOTS, NUM_BLK2_THREADS, true>;
            const size_t lds = (size_t)NUM_BLK2_SLOTS * (sizeof(ACC) + 4);
            GBA_OK(set_lds(k, lds));
            hipLaunchKernelGGL(k, dim3((unsigned)nvec), dim3(NUM_BLK2_THREADS), lds, st, sr, nvec, (const int64_t *)nullptr,
                               P.A, P.B, P.M, P.lpa, (const int64_t *)nullptr, slab_i.as<int64_t>(), slab_x.as<Z>(),
                               vflops.as<int64_t>(), scnt.as<int64_t>());
            GBA_HIP(hipGetLastError());
            return GBA_SUCCESS;
        }));
        std::vector<int64_t> hp((size_t)nvec + 1, 0);
        GBA_HIP(hipMemcpyAsync(hp.data() + 1, scnt.p, (size_t)nvec * 8, hipMemcpyDeviceToHost, st));
        GBA_HIP(hipStreamSynchronize(st));
        bool fits = true;
        for (int64_t j = 1; j <= nvec; j++) { if (hp[(size_t)j] < 0) { fits = false; break; } hp[(size_t)j] += hp[(size_t)j - 1]; }
        if (fits) {
            const int64_t cnz = hp[(size_t)nvec];
            DevBuf Cp, Ci, Cx;
            GBA_OK(Cp.alloc((size_t)(nvec + 1) * 8));
            GBA_OK(Ci.alloc((size_t)(cnz > 0 ? cnz : 1) * 8));
            GBA_OK(Cx.alloc((size_t)(cnz > 0 ? cnz : 1) * zsize));
            GBA_HIP(hipMemcpyAsync(Cp.p, hp.data(), (size_t)(nvec + 1) * 8, hipMemcpyHostToDevice, st));
            if (cnz > 0) {
                GBA_OK(with_semiring(c, [&](auto sr) -> gba_info {
                    using Z = typename decltype(sr)::z_t;
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(small_pack_kernel<Z>), dim3((unsigned)nvec), dim3(256), 0, st,
                                       Cp.as<int64_t>(), slab_i.as<int64_t>(), slab_x.as<Z>(), Ci.as<int64_t>(), Cx.as<Z>());
                    GBA_HIP(hipGetLastError());
                    return GBA_SUCCESS;
                }));
            }
            GBA_HIP(hipStreamSynchronize(st));          // hp and the slabs are released on return
            if (rep) {
                const int e9 = clock.mark();
                rep->phase_ms[0] = clock.ms(e0, e9);
                rep->kernel_ms = clock.ms(e0, e9);
                rep->cnz = cnz;
            }
            C->vlen = A->vlen; C->vdim = B->vdim; C->nvec = nvec; C->nvals = cnz;
            C->p = reinterpret_cast<int64_t *>(Cp.release());
            C->h = nullptr;
            C->i = reinterpret_cast<int64_t *>(Ci.release());
            C->x = Cx.release();
            C->b = nullptr;
            C->nzombies = 0;
            C->type = c.ztype;
            C->sparsity = GBA_SPARSE;
            C->mem = GBA_MEM_DEVICE;
            C->iso = false;
            C->is_csc = B->is_csc;
            C->jumbled = true;
            if (mask_applied) *mask_applied = (M != nullptr);
            if (do_sort) GBA_OK(matrix_wait(C));
            return GBA_SUCCESS;
        }
    }
    // a few very heavy vectors: the whole grid works on each of them (team kernels)
    DevBuf team_list, team_Hx, team_Hf, symwork;
    const int64_t *symflops = vflops.as<int64_t>();
    {
        // the count and the list come back in one read
        GBA_OK(team_list.alloc((1 + TEAM_MAX_VECTORS) * 8));
        unsigned long long *tcount = team_list.as<unsigned long long>();
        int64_t *tlist = team_list.as<int64_t>() + 1;
        GBA_HIP(hipMemsetAsync(tcount, 0, 8, st));
        if (nvec > 0) {
            int64_t blocks = ceil_div(nvec, 256); if (blocks > 2048) blocks = 2048;
            // With only a few vectors in all (a frontier step has one) nothing else can fill the chip: every vector
            // too heavy for a workgroup's LDS table is then worked on by the whole grid.
            const int64_t min_flops = (nvec <= TEAM_MAX_VECTORS) ? (int64_t)SYM_BLK1_MAX : TEAM_MIN_FLOPS;
            hipLaunchKernelGGL(team_select_kernel, dim3((unsigned)blocks), dim3(256), 0, st, nvec, vflops.as<int64_t>(),
                               min_flops, tcount, tlist, TEAM_MAX_VECTORS);
        }
        int64_t hteam[1 + TEAM_MAX_VECTORS];
        GBA_HIP(hipMemcpyAsync(hteam, team_list.p, sizeof(hteam), hipMemcpyDeviceToHost, st));
        GBA_HIP(hipStreamSynchronize(st));
        const int64_t nteam = hteam[0];
        const size_t acc_size = type_size(c.ztype) < 4 ? 4 : type_size(c.ztype);
        const int64_t nf = (P.cvlen + 63) >> 6;
        const size_t need = (size_t)nteam * ((size_t)P.cvlen * acc_size + (size_t)nf * 8);
        // the workspaces must leave room for the product; a need of a few per cent of the device is not worth the
        // driver query that tells how much is free
        bool room = need < ctx().total_mem / 32;
        if (nteam > 0 && !room) {
            size_t free_b = 0, total_b = 0, idle_b = 0;
            GBA_HIP(hipMemGetInfo(&free_b, &total_b));
            gba_pool_stats(&idle_b, nullptr);
            room = need < (free_b + idle_b) / 4;
        }
        int64_t team_max = TEAM_MAX_VECTORS;                 // GBA_TEAM_MAX_VECTORS=0 turns the team path off (tests)
        if (const char *e = getenv("GBA_TEAM_MAX_VECTORS")) { team_max = atoll(e); if (team_max > TEAM_MAX_VECTORS) team_max = TEAM_MAX_VECTORS; }
        if (nteam > 0 && nteam <= team_max && room) {
            P.team.assign(hteam + 1, hteam + 1 + nteam);
            GBA_OK(team_Hx.alloc((size_t)nteam * (size_t)P.cvlen * acc_size));
            GBA_OK(team_Hf.alloc((size_t)nteam * (size_t)nf * 8));
            P.team_Hx = team_Hx.p; P.team_Hf = team_Hf.as<uint64_t>();
            // the bins must not see them
            GBA_OK(symwork.alloc((size_t)(nvec + 1) * 8));
            GBA_HIP(hipMemcpyAsync(symwork.p, vflops.p, (size_t)(nvec + 1) * 8, hipMemcpyDeviceToDevice, st));
            hipLaunchKernelGGL(zero_listed_kernel, dim3(1), dim3(256), 0, st, (int)nteam, team_list.as<int64_t>() + 1, symwork.as<int64_t>());
            GBA_HIP(hipGetLastError());
            symflops = symwork.as<int64_t>();
        }
    }
    // With sorted A the bitmap panel kernel also takes the vectors of the second block bin: it has no table to
    // clear per vector and records the numeric panel boundaries.
    // The panel kernels need sorted A and read its indices from a 32-bit copy made per call: a pass over all of
    // A, which a small product (a short frontier against a large graph) must not pay for.  Such products
    // keep the memory-side accumulator kernels.
    // when the team takes every vector (the one vector of a frontier step) there is nothing to bin, count or size
    const bool all_team = nvec > 0 && (int64_t)P.team.size() == nvec;
    // A jumbled A (the result of an earlier product, say) is multiplied as it is by the reference, whose Gustavson and
    // hash tasks do not care about the order inside a vector.  The panel kernels do: when the product has several
    // multiply-adds per entry of A, a sorted copy of A's indices and values is made first (the caller's matrix stays
    // jumbled), which costs one segmented sort and keeps the heavy vectors off the memory-side accumulators.
    if (A->jumbled && !all_team && !A->iso && nvec > 0 && A->nvals >= ((int64_t)1 << 16) && !getenv("GBA_SAXPY3_KEEP_JUMBLED")) {
        DevBuf w; GBA_OK(w.alloc(16));
        GBA_HIP(hipMemsetAsync(w.p, 0, 16, st));
        int64_t blocks = ceil_div(nvec, 256); if (blocks > 512) blocks = 512;
        hipLaunchKernelGGL(mask_work_kernel, dim3((unsigned)blocks), dim3(256), 0, st, nvec, B->p, B->p, vflops.as<int64_t>(),
                           w.as<unsigned long long>());
        unsigned long long hw[2] = {0, 0};
        GBA_HIP(hipMemcpyAsync(hw, w.p, 16, hipMemcpyDeviceToHost, st));
        GBA_HIP(hipStreamSynchronize(st));
        if ((double)hw[0] >= 4.0 * (double)A->nvals) {
            int64_t aend = 0;
            GBA_OK(dev_read_i64(A->p + A->nvec, &aend));
            const size_t esize = esize_of(A->type);
            GBA_OK(sorted_i.alloc((size_t)(aend > 0 ? aend : 1) * 8));
            GBA_OK(sorted_x.alloc((size_t)(aend > 0 ? aend : 1) * esize));
            GBA_HIP(hipMemcpyAsync(sorted_i.p, A->i, (size_t)aend * 8, hipMemcpyDeviceToDevice, st));
            GBA_HIP(hipMemcpyAsync(sorted_x.p, A->x, (size_t)aend * esize, hipMemcpyDeviceToDevice, st));
            A_sorted = *A;
            A_sorted.i = sorted_i.as<int64_t>(); A_sorted.x = sorted_x.p;
            GBA_OK(matrix_wait(&A_sorted));                // no zombies: sorts inside the two arrays above
            A = &A_sorted;
            P.A = SpView{A->p, A->i, A->x, A->iso};
        }
    }
    P.use_panels = !A->jumbled && !all_team;
    if (P.use_panels && B->nvals < A->nvals / 32 && nvec > 0) {
        // few entries in B: the product may still be large (a block of hub rows); the multiply-adds decide
        DevBuf w; GBA_OK(w.alloc(16));
        GBA_HIP(hipMemsetAsync(w.p, 0, 16, st));
        int64_t blocks = ceil_div(nvec, 256); if (blocks > 512) blocks = 512;
        hipLaunchKernelGGL(mask_work_kernel, dim3((unsigned)blocks), dim3(256), 0, st, nvec, B->p, B->p, vflops.as<int64_t>(),
                           w.as<unsigned long long>());
        unsigned long long hw[2] = {0, 0};
        GBA_HIP(hipMemcpyAsync(hw, w.p, 16, hipMemcpyDeviceToHost, st));
        GBA_HIP(hipStreamSynchronize(st));
        P.use_panels = (double)hw[0] >= (double)A->nvals / 8;
    }
    // (measured at RMAT-22: handing the vectors of 2049-4096 multiply-adds to the panel kernel as well saves a quarter of
    // the first block bin's time, and gives the numeric sweep their boundaries and patterns; since the panel kernel counts
    // a vector below one stretch in a single hashed walk, 1024 is better still: 358 -> 352 ms of symbolic time; 512: 353)
    const int64_t sym_blk1_max = P.use_panels ? (int64_t)SYM_PANEL_MIN : (int64_t)SYM_BLK1_MAX;
    const int64_t sym_blk2_max = P.use_panels ? (int64_t)SYM_PANEL_MIN : (int64_t)SYM_BLK2_MAX;
    if (!all_team) GBA_OK(build_bins(nvec, symflops, false, BinLimits{SYM_WAVE_MAX, sym_blk1_max, sym_blk2_max}, &P.sym));
    P.vflops = vflops.as<int64_t>();
    if (all_team && nvec == 1) P.maxdeg = B->nvals;         // (the one vector holds every entry of B)
    else if (P.sym.count[BIN_GLOBAL] > 0 || !P.team.empty()) GBA_OK(dev_max_diff_i64(B->p, nvec, &P.maxdeg));
    // the panel kernels read the indices of A from a 32-bit copy (vlen < 2^31 was checked above)
    DevBuf ai32;
    auto narrow_A = [&]() -> gba_info {
        if (P.A.i32) return GBA_SUCCESS;
        int64_t aend = 0;
        GBA_OK(dev_read_i64(A->p + A->nvec, &aend));
        GBA_OK(ai32.alloc((size_t)(aend > 0 ? aend : 1) * 4));
        if (aend > 0) {
            int64_t blocks = ceil_div(aend, 1024); if (blocks > 16384) blocks = 16384;
            hipLaunchKernelGGL(narrow_index_kernel, dim3((unsigned)blocks), dim3(256), 0, st, A->i, ai32.as<int32_t>(), aend);
            GBA_HIP(hipGetLastError());
        }
        P.A.i32 = ai32.as<int32_t>();
        return GBA_SUCCESS;
    };
    DevBuf bstart, bounds, pstart, patbuf, wbounds;
    if (P.sym.count[BIN_GLOBAL] > 0 && P.use_panels && P.maxdeg < PANEL_MAX_DEG) {
        GBA_OK(narrow_A());
        GBA_OK(bstart.alloc((size_t)(nvec + 1) * 8));
        int64_t blocks = ceil_div(nvec + 1, 256); if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(bounds_room_kernel, dim3((unsigned)blocks), dim3(256), 0, st, nvec, vflops.as<int64_t>(),
                           P.cvlen, sym_blk2_max, bstart.as<int64_t>());     // team vectors get room too; unused
        GBA_HIP(hipGetLastError());
        GBA_OK(dev_cumsum_i64(bstart.as<int64_t>(), nvec, st));
        int64_t room = 0;
        GBA_OK(dev_read_i64(bstart.as<int64_t>() + nvec, &room));
        GBA_OK(bounds.alloc((size_t)(room > 0 ? room : 1) * 8));
        P.bstart = bstart.as<int64_t>();
        P.bounds = bounds.as<int64_t>();
        // Room for the patterns (one 8-byte pair per non-empty bitmap word: at most one per multiply-add and per 32
        // indices): with them the numeric sweep ranks instead of hashing.  Without the room (or with
        // GBA_PANEL_RANKED=0) it hashes as before.
        const char *renv = getenv("GBA_PANEL_RANKED");               // read per call: tests switch it
        const bool ranked = !(renv && atoi(renv) == 0);
        if (ranked) {
            GBA_OK(pstart.alloc((size_t)(nvec + 1) * 8));
            hipLaunchKernelGGL(pattern_room_kernel, dim3((unsigned)blocks), dim3(256), 0, st, nvec, vflops.as<int64_t>(),
                               (P.cvlen + 31) / 32, sym_blk2_max, pstart.as<int64_t>());
            GBA_HIP(hipGetLastError());
            GBA_OK(dev_cumsum_i64(pstart.as<int64_t>(), nvec, st));
            int64_t proom = 0;
            GBA_OK(dev_read_i64(pstart.as<int64_t>() + nvec, &proom));
            size_t free_b = 0, total_b = 0, idle_b = 0;
            GBA_HIP(hipMemGetInfo(&free_b, &total_b));
            gba_pool_stats(&idle_b, nullptr);
            if ((size_t)proom * 8 < (free_b + idle_b) / 3) {
                GBA_OK(patbuf.alloc((size_t)(proom > 0 ? proom : 1) * 8));
                GBA_OK(wbounds.alloc((size_t)(room > 0 ? room : 1) * 8));
                P.pstart = pstart.as<int64_t>();
                P.pwords = patbuf.as<uint2>();
                P.wbounds = wbounds.as<int64_t>();
            }
        }
    }
    const int e1 = clock.mark();

    // phase 1: symbolic, cnt[j] = nnz C(:,j), written into the future C->p
    DevBuf Cp; GBA_OK(Cp.alloc((size_t)(nvec + 1) * 8));
    GBA_HIP(hipMemsetAsync(Cp.p, 0, (size_t)(nvec + 1) * 8, st));
    GBA_OK(run_symbolic(P, Cp.as<int64_t>()));
    if (!P.team.empty()) GBA_OK(with_semiring(c, [&](auto sr) -> gba_info { return run_team_products(sr, P, Cp.as<int64_t>()); }));
    const int e2 = clock.mark();

    // numeric bins use the exact counts, taken before the prefix sum
    if (!all_team) {
        const int64_t *numwork = Cp.as<int64_t>();
        DevBuf nw;
        if (!P.team.empty()) {
            GBA_OK(nw.alloc((size_t)(nvec + 1) * 8));
            GBA_HIP(hipMemcpyAsync(nw.p, Cp.p, (size_t)(nvec + 1) * 8, hipMemcpyDeviceToDevice, st));
            hipLaunchKernelGGL(zero_listed_kernel, dim3(1), dim3(256), 0, st, (int)P.team.size(), team_list.as<int64_t>() + 1, nw.as<int64_t>());
            GBA_HIP(hipGetLastError());
            numwork = nw.as<int64_t>();
        }
        // products per thread stay bounded: by flops a vector may need a larger bin than its entry count asks for
        // (with the panel kernel at hand the second block bin stays empty: the panel kernel emits those vectors in a
        // third of the time; and it takes the vectors of 1025 .. 2048 entries as well -- at RMAT-22 the first block
        // bin falls from 58 to 9 ms per step for 13 ms more in the panel kernel; from 513 on the panel kernel loses more
        // than the block bin gains)
        const bool num_panels = P.use_panels && P.maxdeg < PANEL_MAX_DEG;
        const int64_t blk1_max = num_panels ? (int64_t)GBA_NUM_PANEL_MIN : (int64_t)NUM_BLK1_MAX;
        GBA_OK(build_bins(nvec, numwork, false, BinLimits{NUM_WAVE_MAX, blk1_max, num_panels ? blk1_max : (int64_t)NUM_BLK2_MAX}, &P.num,
                          vflops.as<int64_t>(), BinLimits{2048, 16384, 65536}));
    }
    if (P.num.count[BIN_GLOBAL] > 0 && P.maxdeg == 0) GBA_OK(dev_max_diff_i64(B->p, nvec, &P.maxdeg));
    if (P.num.count[BIN_GLOBAL] > 0 && P.use_panels && P.maxdeg < PANEL_MAX_DEG) GBA_OK(narrow_A());

    // phases 3/4: Cp = cumsum (cnt)
    GBA_OK(dev_cumsum_i64(Cp.as<int64_t>(), nvec, st));
    int64_t cnz = 0;
    GBA_OK(dev_read_i64(Cp.as<int64_t>() + nvec, &cnz));

    DevBuf Ci, Cx;
    const size_t zsize = type_size(c.ztype);
    GBA_OK(Ci.alloc((size_t)(cnz > 0 ? cnz : 1) * 8));
    GBA_OK(Cx.alloc((size_t)(cnz > 0 ? cnz : 1) * zsize));
    P.Cp = Cp.as<int64_t>(); P.Ci = Ci.as<int64_t>(); P.Cx = Cx.p;

    // phase 5: numeric
    if (cnz > 0) {
        GBA_OK(with_semiring(c, [&](auto sr) -> gba_info { return run_numeric(sr, P); }));
        if (!P.team.empty()) GBA_OK(with_semiring(c, [&](auto sr) -> gba_info { return run_team_gather(sr, P); }));
    }
    GBA_HIP(hipStreamSynchronize(st));
    if (rep) {
        const int e4 = clock.mark();
        rep->phase_ms[0] = clock.ms(e0, e1);
        rep->phase_ms[1] = clock.ms(e1, e2);
        if (cnz > 0) {
            rep->phase_ms[2] = clock.ms(e2, P.ev_num[0]);
            for (int b = 0; b < 4; b++) rep->phase_ms[3 + b] = clock.ms(P.ev_num[b], P.ev_num[b + 1]);
        } else {
            rep->phase_ms[2] = clock.ms(e2, e4);
        }
        rep->kernel_ms = clock.ms(e0, e4);
        rep->cnz = cnz;
        rep->bin_vectors[0] = P.num.count[BIN_WAVE]; rep->bin_vectors[1] = P.num.count[BIN_BLK1];
        rep->bin_vectors[2] = P.num.count[BIN_BLK2]; rep->bin_vectors[3] = P.num.count[BIN_GLOBAL];
        rep->panels = (int64_t)P.panel_stats[1]; rep->panels_redone = (int64_t)P.panel_stats[2];
#ifdef GBA_PANEL_TICKS
        if (getenv("GBA_PANEL_PROF"))
            fprintf(stderr, "[gba] numeric panels %llu; 10 ns ticks summed over workgroups: search %llu (+wait %llu) prefix %llu "
                    "products %llu (+wait %llu) emit %llu panels %llu kernel %llu\n", P.panel_stats[1], P.panel_stats[3],
                    P.panel_stats[4], P.panel_stats[5], P.panel_stats[6], P.panel_stats[7], P.panel_stats[8], P.panel_stats[9],
                    P.panel_stats[10]);
#endif
        DevBuf bs; GBA_OK(bs.alloc(12 * 8));
        GBA_HIP(hipMemsetAsync(bs.p, 0, 96, st));
        for (int b = 0; b < 4; b++) {
            const int64_t cntb = P.num.count[BIN_WAVE + b];
            if (cntb == 0) continue;
            int64_t blocks = ceil_div(cntb, 256); if (blocks > 1024) blocks = 1024;
            hipLaunchKernelGGL(bin_stats_kernel, dim3((unsigned)blocks), dim3(256), 0, st, cntb, P.num.list(BIN_WAVE + b),
                               Cp.as<int64_t>(), B->p, vflops.as<int64_t>(), bs.as<unsigned long long>() + 3 * b);
        }
        unsigned long long hs[12];
        GBA_HIP(hipMemcpyAsync(hs, bs.p, 96, hipMemcpyDeviceToHost, st));
        GBA_HIP(hipStreamSynchronize(st));
        for (int b = 0; b < 4; b++) {
            rep->bin_cnz[b] = (int64_t)hs[3 * b]; rep->bin_bnz[b] = (int64_t)hs[3 * b + 1];
            rep->bin_madds[b] = (int64_t)hs[3 * b + 2];
        }
    }

    C->vlen = A->vlen; C->vdim = B->vdim; C->nvec = nvec; C->nvals = cnz;
    C->p = reinterpret_cast<int64_t *>(Cp.release());
    C->h = nullptr;
    C->i = reinterpret_cast<int64_t *>(Ci.release());
    C->x = Cx.release();
    C->b = nullptr;
    C->nzombies = 0;
    C->type = c.ztype;
    C->sparsity = GBA_SPARSE;
    C->mem = GBA_MEM_DEVICE;
    C->iso = false;
    C->is_csc = B->is_csc;
    C->jumbled = true;      // hash bins emit in table order (the reference leaves C jumbled too)
    if (mask_applied) *mask_applied = (M != nullptr);
    if (do_sort) GBA_OK(matrix_wait(C));
    return GBA_SUCCESS;
}

// ---- vectors of 2^31 indices or more -----------------------------------------------------------------
// The kernels above hold indices of C in 32 bits (hash keys, bitmap offsets, the narrowed copy of A's indices); the
// CPU method has no such limit (Source/GB_AxB_saxpy3.c:120 takes any vlen).  A matrix that long holds few of its
// indices, and only indices that occur in A can occur in C: the distinct indices of A are ranked (one sort), A and
// the mask are rewritten in ranks -- entries of the mask at other indices can never meet a product and are dropped --
// the product is formed over the ranks, and the result's ranks are turned back into indices.  Ranking keeps the
// order, so sorted vectors stay sorted.
__global__ void __launch_bounds__(256)
rank_of_index_kernel(int64_t n, const int64_t *__restrict__ idx, const int64_t *__restrict__ U, int64_t nu,
                     int64_t *__restrict__ out) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; e < n; e += stride) {
        const int64_t v = idx[e];
        int64_t a = 0, b = nu;
        while (a < b) { const int64_t m = (a + b) >> 1; if (U[m] < v) a = m + 1; else b = m; }
        out[e] = (a < nu && U[a] == v) ? a : -1;          // -1: not among the ranked indices
    }
}

__global__ void __launch_bounds__(256)
index_of_rank_kernel(int64_t n, int64_t *__restrict__ idx, const int64_t *__restrict__ U) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; e < n; e += stride) {
        const int64_t r = idx[e];
        idx[e] = (r >= 0) ? U[r] : -U[-r - 2] - 2;        // (a zombie keeps its flipped form, GB_FLIP (i) = -i - 2)
    }
}

// the ranked entries of every vector of the mask, kept in order: a wavefront per vector; FILL = false counts
template <bool FILL>
__global__ void __launch_bounds__(256)
ranked_mask_kernel(int64_t nvec, const int64_t *__restrict__ Mp, const int64_t *__restrict__ rank,
                   const unsigned char *__restrict__ Mx, int msize, int64_t *__restrict__ Rp, int64_t *__restrict__ Ri,
                   unsigned char *__restrict__ Rx) {
    const int lane = threadIdx.x & 63;
    int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6, p0 = Mp[0];
    for (; j < nvec; j += stride) {
        const int64_t m0 = Mp[j], m1 = Mp[j + 1];
        int64_t out = FILL ? Rp[j] : 0;
        for (int64_t q = m0; q < m1; q += 64) {
            const int64_t p = q + lane;
            const int64_t r = (p < m1) ? rank[p - p0] : -1;
            const unsigned long long m = __ballot(r >= 0);
            if (FILL && r >= 0) {
                const int64_t w = out + __builtin_popcountll(m & ((1ull << lane) - 1ull));
                Ri[w] = r;
                if (Rx) for (int b = 0; b < msize; b++) Rx[w * msize + b] = Mx[p * msize + b];
            }
            out += __builtin_popcountll(m);
        }
        if (!FILL && lane == 0) Rp[j] = out;
    }
}

// out[e] = the rank of idx[e] in the ascending list U[0..nu), or -1 where the list does not hold it; and the way back
gba_info ranks_in_list(const int64_t *idx, int64_t n, const int64_t *U, int64_t nu, int64_t *out) {
    if (n <= 0) return GBA_SUCCESS;
    int64_t b = ceil_div(n, 256); if (b > 16384) b = 16384;
    hipLaunchKernelGGL(rank_of_index_kernel, dim3((unsigned)b), dim3(256), 0, ctx().stream, n, idx, U, nu, out);
    GBA_HIP(hipGetLastError());
    return GBA_SUCCESS;
}

gba_info indices_of_ranks(int64_t *idx, int64_t n, const int64_t *U) {
    if (n <= 0) return GBA_SUCCESS;
    int64_t b = ceil_div(n, 256); if (b > 16384) b = 16384;
    hipLaunchKernelGGL(index_of_rank_kernel, dim3((unsigned)b), dim3(256), 0, ctx().stream, n, idx, U);
    GBA_HIP(hipGetLastError());
    return GBA_SUCCESS;
}

// X (sparse or hypersparse) with its indices replaced by their ranks in the ascending list U[0..nu) and the entries
// at other indices (and zombies) dropped, the rest kept in order: new p and i, a compacted copy of x when the values
// are wanted and X is not iso-valued (else X's own x), vlen = new_vlen; h, if any, is X's own.  The new arrays are
// added to `owned`.
gba_info ranked_copy(const gba_matrix *X, const int64_t *U, int64_t nu, int64_t new_vlen, bool with_values,
                     gba_matrix *out, std::vector<void *> &owned) {
    hipStream_t st = ctx().stream;
    if (!is_sparse(X) && !is_hyper(X)) GBA_FAIL(GBA_INVALID_VALUE, "ranked_copy: sparsity %d", X->sparsity);
    auto grid = [](int64_t n) { int64_t b = ceil_div(n, 256); if (b > 16384) b = 16384; if (b < 1) b = 1; return dim3((unsigned)b); };
    int64_t m0 = 0, m1 = 0;
    const int64_t nvec = X->nvec;
    if (X->p) { GBA_OK(dev_read_i64(X->p, &m0)); GBA_OK(dev_read_i64(X->p + nvec, &m1)); }
    const int64_t mnz = m1 - m0;
    const int msize = (int)esize_of(X->type);
    const bool valued = with_values && !X->iso && X->x;
    DevBuf Mr, Rp, Ri, Rx;
    GBA_OK(Mr.alloc((size_t)(mnz > 0 ? mnz : 1) * 8));
    if (mnz > 0) hipLaunchKernelGGL(rank_of_index_kernel, grid(mnz), dim3(256), 0, st, mnz, X->i + m0, U, nu, Mr.as<int64_t>());
    GBA_OK(Rp.alloc((size_t)(nvec + 1) * 8));
    GBA_HIP(hipMemsetAsync(Rp.p, 0, (size_t)(nvec + 1) * 8, st));
    const dim3 wgrid = grid(nvec * 64);
    if (nvec > 0 && mnz > 0)
        hipLaunchKernelGGL(ranked_mask_kernel<false>, wgrid, dim3(256), 0, st, nvec, X->p, Mr.as<int64_t>(),
                           (const unsigned char *)nullptr, msize, Rp.as<int64_t>(), (int64_t *)nullptr, (unsigned char *)nullptr);
    GBA_HIP(hipGetLastError());
    GBA_OK(dev_cumsum_i64(Rp.as<int64_t>(), nvec, st));
    int64_t rnz = 0;
    GBA_OK(dev_read_i64(Rp.as<int64_t>() + nvec, &rnz));
    GBA_OK(Ri.alloc((size_t)(rnz > 0 ? rnz : 1) * 8));
    if (valued) GBA_OK(Rx.alloc((size_t)(rnz > 0 ? rnz : 1) * msize));
    if (nvec > 0 && rnz > 0)
        hipLaunchKernelGGL(ranked_mask_kernel<true>, wgrid, dim3(256), 0, st, nvec, X->p, Mr.as<int64_t>(),
                           reinterpret_cast<const unsigned char *>(X->x), msize, Rp.as<int64_t>(), Ri.as<int64_t>(),
                           valued ? Rx.as<unsigned char>() : (unsigned char *)nullptr);
    GBA_HIP(hipGetLastError());
    GBA_HIP(hipStreamSynchronize(st));                      // the ranks go back to the pool
    *out = *X;
    out->vlen = new_vlen; out->nvals = rnz; out->nzombies = 0;
    out->p = reinterpret_cast<int64_t *>(Rp.release()); owned.push_back(out->p);
    out->i = reinterpret_cast<int64_t *>(Ri.release()); owned.push_back(out->i);
    if (valued) { out->x = Rx.release(); owned.push_back(out->x); }      // (an iso or structural X keeps reading its own value)
    return GBA_SUCCESS;
}

static gba_info saxpy3_long_vectors(gba_matrix *C, const gba_matrix *M, bool Mask_comp, bool Mask_struct,
                                    const gba_matrix *A, const gba_matrix *B, const Canon &c,
                                    int axb_method, int do_sort, bool *mask_applied, gba_report *rep, bool may_discard_mask) {
    hipStream_t st = ctx().stream;
    if (M && !is_sparse(M)) GBA_FAIL(GBA_NOT_IMPLEMENTED, "saxpy3: a bitmap or full mask over vectors of 2^31 indices or more");
    if (M && (M->vlen != A->vlen || M->vdim != B->vdim)) GBA_FAIL(GBA_DIMENSION_MISMATCH, "saxpy3: mask dimensions");
    if (M && (M->nzombies || M->jumbled)) GBA_FAIL(GBA_INVALID_OBJECT, "saxpy3: the mask has pending work");
    int64_t a0 = 0, a1 = 0;
    GBA_OK(dev_read_i64(A->p, &a0));
    GBA_OK(dev_read_i64(A->p + A->nvec, &a1));
    const int64_t anz = a1 - a0;
    int bits = 1;
    while (bits < 63 && ((int64_t)1 << bits) < A->vlen) bits++;
    DevBuf U, Ar;
    int64_t nu = 0;
    GBA_OK(distinct_sorted(A->i + a0, anz, bits, &U, &nu));
    if (nu >= (int64_t)0x7fffffff) GBA_FAIL(GBA_NOT_IMPLEMENTED, "saxpy3: %ld distinct indices in A", (long)nu);
    GBA_OK(Ar.alloc((size_t)(anz > 0 ? anz : 1) * 8));
    auto grid = [](int64_t n) { int64_t b = ceil_div(n, 256); if (b > 16384) b = 16384; if (b < 1) b = 1; return dim3((unsigned)b); };
    if (anz > 0) hipLaunchKernelGGL(rank_of_index_kernel, grid(anz), dim3(256), 0, st, anz, A->i + a0, U.as<int64_t>(), nu, Ar.as<int64_t>());
    GBA_HIP(hipGetLastError());
    gba_matrix A2 = *A;
    A2.vlen = nu > 0 ? nu : 1;
    A2.i = Ar.as<int64_t>() - a0;                          // (indexed by the same absolute offsets as A->i)
    gba_matrix M2;
    std::vector<void *> mine;
    struct Free { std::vector<void *> &v; ~Free() { for (void *q : v) gba_free(q); } } free_mine{mine};
    if (M) GBA_OK(ranked_copy(M, U.as<int64_t>(), nu, A2.vlen, !Mask_struct, &M2, mine));
    const gba_info info = saxpy3(C, M ? &M2 : nullptr, Mask_comp, Mask_struct, &A2, B, c, axb_method, do_sort, mask_applied, rep, may_discard_mask);
    if (info != GBA_SUCCESS) return info;                   // (GBA_NO_VALUE: the caller multiplies again without the mask)
    int64_t cnz = 0;
    if (C->p) GBA_OK(dev_read_i64(C->p + C->nvec, &cnz));
    if (cnz > 0) hipLaunchKernelGGL(index_of_rank_kernel, grid(cnz), dim3(256), 0, st, cnz, C->i, U.as<int64_t>());
    GBA_HIP(hipGetLastError());
    GBA_HIP(hipStreamSynchronize(st));                      // U is released on return
    C->vlen = A->vlen;
    return GBA_SUCCESS;
}

}  // namespace gba
