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
            hipLaunchKernelGGL(flopcount_wave_kernel, dim3((unsigned)ceil_div(nvec * 64, 256)), dim3(256), 0, st,
                               nvec, B->p, B->i, A->p, Mp, Bflops, mw.as<unsigned long long>());
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

struct BinLimits { int64_t wave_max, blk1_max, blk2_max; };

__device__ inline int bin_of(int64_t w, const BinLimits &L) {
    if (w <= 0) return BIN_EMPTY;
    if (w <= L.wave_max) return BIN_WAVE;
    if (w <= L.blk1_max) return BIN_BLK1;
    if (w <= L.blk2_max) return BIN_BLK2;
    return BIN_GLOBAL;
}

__global__ void __launch_bounds__(256)
bin_count_kernel(int64_t nvec, const int64_t *__restrict__ work, bool work_is_pointer_diff, BinLimits L,
                 unsigned long long *__restrict__ counts) {
    __shared__ unsigned int lc[NBINS];
    if (threadIdx.x < NBINS) lc[threadIdx.x] = 0;
    __syncthreads();
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; idx < nvec; idx += stride) {
        const int64_t w = work_is_pointer_diff ? (work[idx + 1] - work[idx]) : work[idx];
        atomicAdd(&lc[bin_of(w, L)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < NBINS && lc[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)lc[threadIdx.x]);
}

__global__ void __launch_bounds__(256)
bin_fill_kernel(int64_t nvec, const int64_t *__restrict__ work, bool work_is_pointer_diff, BinLimits L,
                unsigned long long *__restrict__ cursors, int64_t *__restrict__ lists) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; idx < nvec; idx += stride) {
        const int64_t w = work_is_pointer_diff ? (work[idx + 1] - work[idx]) : work[idx];
        const int b = bin_of(w, L);
        if (b != BIN_EMPTY) {
            unsigned long long pos = atomicAdd(&cursors[b], 1ull);
            lists[pos] = idx;
        }
    }
}

struct Bins {
    DevBuf lists;                 // vector ids grouped by bin
    int64_t count[NBINS] = {0, 0, 0, 0, 0};
    int64_t start[NBINS] = {0, 0, 0, 0, 0};
    const int64_t *list(int b) const { return lists.as<int64_t>() + start[b]; }
};

static gba_info build_bins(int64_t nvec, const int64_t *work, bool ptr_diff, BinLimits L, Bins *bins) {
    hipStream_t st = ctx().stream;
    DevBuf cnt; GBA_OK(cnt.alloc(NBINS * 8));
    GBA_HIP(hipMemsetAsync(cnt.p, 0, NBINS * 8, st));
    GBA_OK(bins->lists.alloc((size_t)(nvec > 0 ? nvec : 1) * 8));
    if (nvec == 0) return GBA_SUCCESS;
    int64_t blocks = ceil_div(nvec, 256); if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(bin_count_kernel, dim3((unsigned)blocks), dim3(256), 0, st, nvec, work, ptr_diff, L,
                       cnt.as<unsigned long long>());
    unsigned long long h[NBINS];
    GBA_HIP(hipMemcpyAsync(h, cnt.p, NBINS * 8, hipMemcpyDeviceToHost, st));
    GBA_HIP(hipStreamSynchronize(st));
    unsigned long long cur[NBINS];
    int64_t off = 0;
    for (int b = 0; b < NBINS; b++) {
        bins->count[b] = (int64_t)h[b];
        bins->start[b] = off;
        cur[b] = (unsigned long long)off;
        if (b != BIN_EMPTY) off += (int64_t)h[b];
    }
    GBA_HIP(hipMemcpyAsync(cnt.p, cur, NBINS * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(bin_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, nvec, work, ptr_diff, L,
                       cnt.as<unsigned long long>(), bins->lists.as<int64_t>());
    GBA_HIP(hipGetLastError());
    GBA_HIP(hipStreamSynchronize(st));
    return GBA_SUCCESS;
}

// totals of nnz C(:,j) and nnz B(:,j) over one bin (reporting only)
__global__ void __launch_bounds__(256)
bin_stats_kernel(int64_t nlist, const int64_t *__restrict__ list, const int64_t *__restrict__ Cp,
                 const int64_t *__restrict__ Bp, unsigned long long *__restrict__ out /* [2] */) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long c = 0, b = 0;
    for (; idx < nlist; idx += stride) {
        const int64_t j = list[idx];
        c += (unsigned long long)(Cp[j + 1] - Cp[j]);
        b += (unsigned long long)(Bp[j + 1] - Bp[j]);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { c += __shfl_down(c, d, 64); b += __shfl_down(b, d, 64); }
    if ((threadIdx.x & 63) == 0) { if (c) atomicAdd(&out[0], c); if (b) atomicAdd(&out[1], b); }
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
    const int64_t j = list[blockIdx.x];
    for (int s = threadIdx.x; s < SLOTS; s += THREADS) keys[s] = HASH_EMPTY;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    int c = 0;
    for_each_product(A, B, j, threadIdx.x, THREADS, lpa, [&](int64_t i, int64_t, int64_t, int64_t) {
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
    const int64_t nf = (cvlen + 63) >> 6, ns = (nf + 63) >> 6;
    uint64_t *Hf = Hf_all + (int64_t)blockIdx.x * nf;
    uint64_t *Hs = Hs_all + (int64_t)blockIdx.x * ns;
    while (true) {
        if (threadIdx.x == 0) cur = (int64_t)atomicAdd(next, 1ull);
        __syncthreads();
        const int64_t li = cur;
        if (li >= nlist) break;                 // uniform: every wave leaves here
        const int64_t j = list[li];
        for_each_product(A, B, j, threadIdx.x, GLOBAL_THREADS, lpa, [&](int64_t i, int64_t, int64_t, int64_t) {
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

template <class S, int SLOTS, int THREADS>
__global__ void __launch_bounds__(THREADS)
num_block_kernel(S sr, int64_t nlist, const int64_t *__restrict__ list, SpView A, SpView B, MaskView M, int lpa,
                 const int64_t *__restrict__ Cp, int64_t *__restrict__ Ci, typename S::z_t *__restrict__ Cx) {
    using Z = typename S::z_t; using ACC = typename S::acc_t;
    extern __shared__ __align__(16) unsigned char smem[];
    ACC *vals = reinterpret_cast<ACC *>(smem);
    int32_t *keys = reinterpret_cast<int32_t *>(smem + (size_t)SLOTS * sizeof(ACC));
    __shared__ int cursor;
    const int64_t j = list[blockIdx.x];
    const ACC ident = (ACC)sr.identity();
    for (int s = threadIdx.x; s < SLOTS; s += THREADS) { keys[s] = HASH_EMPTY; vals[s] = ident; }
    if (threadIdx.x == 0) cursor = 0;
    __syncthreads();
    const bool apat = sr.a_is_pattern(), bpat = sr.b_is_pattern();
    for_each_product(A, B, j, threadIdx.x, THREADS, lpa, [&](int64_t i, int64_t k, int64_t pA, int64_t pB) {
        if (!mask_allows(M, i, j)) return;
        const Z t = product(sr, A, B, apat, bpat, i, k, j, pA, pB);
        bool is_new;
        const uint32_t h = hash_insert(keys, SLOTS - 1, (int32_t)i, &is_new);
        sr.atomic(&vals[h], t);
    });
    __syncthreads();
    const int64_t pC = Cp[j];
    for (int s = threadIdx.x; s < SLOTS; s += THREADS) {
        const int32_t key = keys[s];
        if (key != HASH_EMPTY) {
            const int pos = atomicAdd(&cursor, 1);
            Ci[pC + pos] = key;
            Cx[pC + pos] = (Z)vals[s];
        }
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
        for_each_product(A, B, j, threadIdx.x, GLOBAL_THREADS, lpa, [&](int64_t i, int64_t k, int64_t pA, int64_t pB) {
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
};

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
                           sr, nb.count[BIN_BLK1], nb.list(BIN_BLK1), P.A, P.B, P.M, P.lpa, P.Cp, P.Ci, Cx);
    }
    mark(2);
    if (nb.count[BIN_BLK2] > 0) {
        auto k = num_block_kernel<S, NUM_BLK2_SLOTS, NUM_BLK2_THREADS>;
        const size_t lds = (size_t)NUM_BLK2_SLOTS * (sizeof(ACC) + 4);
        GBA_OK(set_lds(k, lds));
        hipLaunchKernelGGL(k, dim3((unsigned)nb.count[BIN_BLK2]), dim3(NUM_BLK2_THREADS), lds, st,
                           sr, nb.count[BIN_BLK2], nb.list(BIN_BLK2), P.A, P.B, P.M, P.lpa, P.Cp, P.Ci, Cx);
    }
    GBA_HIP(hipGetLastError());
    mark(3);
    if (nb.count[BIN_GLOBAL] > 0) {
        const int64_t nf = (P.cvlen + 63) >> 6, ns = (nf + 63) >> 6;
        const size_t per_block = (size_t)P.cvlen * sizeof(ACC) + (size_t)(nf + ns) * 8;
        size_t free_b = 0, total_b = 0;
        GBA_HIP(hipMemGetInfo(&free_b, &total_b));
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
    if (sb.count[BIN_GLOBAL] > 0) {
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

gba_info saxpy3(gba_matrix *C, const gba_matrix *M, bool Mask_comp, bool Mask_struct,
                const gba_matrix *A, const gba_matrix *B, const Canon &c,
                int axb_method, int do_sort, bool *mask_applied, gba_report *rep) {
    (void)axb_method;
    GBA_OK(require_init());
    if (!is_sparse(A) || !is_sparse(B))
        GBA_FAIL(GBA_NOT_IMPLEMENTED, "saxpy3 on device needs sparse A and B (got sparsity %d, %d)", A->sparsity, B->sparsity);
    if (A->vdim != B->vlen) GBA_FAIL(GBA_DIMENSION_MISMATCH, "saxpy3: A->vdim %ld != B->vlen %ld", (long)A->vdim, (long)B->vlen);
    if (A->nzombies || B->nzombies) GBA_FAIL(GBA_INVALID_OBJECT, "saxpy3: operands have zombies");
    if (A->vlen >= (int64_t)0x7fffffff) GBA_FAIL(GBA_NOT_IMPLEMENTED, "saxpy3: vector length >= 2^31");
    if (M && (M->vlen != A->vlen || M->vdim != B->vdim)) GBA_FAIL(GBA_DIMENSION_MISMATCH, "saxpy3: mask dimensions");
    hipStream_t st = ctx().stream;
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
    GBA_OK(build_bins(nvec, vflops.as<int64_t>(), false, BinLimits{SYM_WAVE_MAX, SYM_BLK1_MAX, SYM_BLK2_MAX}, &P.sym));
    const int e1 = clock.mark();

    // phase 1: symbolic, cnt[j] = nnz C(:,j), written into the future C->p
    DevBuf Cp; GBA_OK(Cp.alloc((size_t)(nvec + 1) * 8));
    GBA_HIP(hipMemsetAsync(Cp.p, 0, (size_t)(nvec + 1) * 8, st));
    GBA_OK(run_symbolic(P, Cp.as<int64_t>()));
    const int e2 = clock.mark();

    // numeric bins use the exact counts, taken before the prefix sum
    GBA_OK(build_bins(nvec, Cp.as<int64_t>(), false, BinLimits{NUM_WAVE_MAX, NUM_BLK1_MAX, NUM_BLK2_MAX}, &P.num));

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
        DevBuf bs; GBA_OK(bs.alloc(8 * 8));
        GBA_HIP(hipMemsetAsync(bs.p, 0, 64, st));
        for (int b = 0; b < 4; b++) {
            const int64_t cntb = P.num.count[BIN_WAVE + b];
            if (cntb == 0) continue;
            int64_t blocks = ceil_div(cntb, 256); if (blocks > 1024) blocks = 1024;
            hipLaunchKernelGGL(bin_stats_kernel, dim3((unsigned)blocks), dim3(256), 0, st, cntb, P.num.list(BIN_WAVE + b),
                               Cp.as<int64_t>(), B->p, bs.as<unsigned long long>() + 2 * b);
        }
        unsigned long long hs[8];
        GBA_HIP(hipMemcpyAsync(hs, bs.p, 64, hipMemcpyDeviceToHost, st));
        GBA_HIP(hipStreamSynchronize(st));
        for (int b = 0; b < 4; b++) { rep->bin_cnz[b] = (int64_t)hs[2 * b]; rep->bin_bnz[b] = (int64_t)hs[2 * b + 1]; }
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

}  // namespace gba
