// Methods with a bitmap or full operand or result: the device counterparts of
//   GB_AxB_dot2   C<#M> = A'*B, C bitmap          (Source/GB_AxB_dot2.c:53)
//   GB_AxB_dot4   C += A'*B,   C full in place    (Source/GB_AxB_dot4.c:49,
//                                                  Source/Template/GB_AxB_dot4_template.c:316-711)
//   GB_AxB_saxpy4 C += A*B, A sparse, B dense     (Source/GB_AxB_saxpy4.c:48)
//   GB_AxB_saxpy5 C += A*B, A dense,  B sparse    (Source/GB_AxB_saxpy5.c:55)
//   GB_AxB_saxbit C<#M> = A*B, C bitmap           (Source/GB_AxB_saxbit.c:46)
// "dense" = bitmap or full.  These are the SpMV / SpMM / frontier-step shapes:
// one sparse operand is streamed once, coalesced, and the other side is
// indexed directly.  The CPU code splits the work into coarse / fine-atomic /
// fine-workspace tasks (Source/GB_AxB_saxpy4_tasks.c:84-103); on the GPU every
// vector of the sparse operand gets a group of lanes sized to its length, long
// vectors go to whole workgroups, and cross-lane results are combined with
// wavefront shuffles (dot) or HBM atomics (saxpy).
#include "gba_internal.hpp"

namespace gba {

struct DenseView {          // bitmap or full operand, column-major (vector-major)
    const void *x; const int8_t *b; int64_t vlen; bool iso;
};
struct MaskViewD {          // mask for the bitmap-C methods; same meaning as in saxpy3
    int kind; bool comp, structural, iso; int msize; int64_t vlen;
    const int64_t *p; const int64_t *i; const int8_t *b; const uint8_t *x;
};

__device__ inline bool maskd_value(const MaskViewD &m, int64_t p) {
    if (m.structural) return true;
    const uint8_t *v = m.x + (m.iso ? 0 : p * m.msize);
    bool nz = false;
    for (int q = 0; q < m.msize; q++) nz |= (v[q] != 0);
    return nz;
}
__device__ inline bool maskd_allows(const MaskViewD &m, int64_t i, int64_t j) {
    if (m.kind == 0) return true;
    bool mij = false;
    if (m.kind == 1) {
        int64_t lo = m.p[j], hi = m.p[j + 1] - 1;
        while (lo <= hi) {
            int64_t mid = (lo + hi) >> 1;
            int64_t v = m.i[mid];
            if (v < i) lo = mid + 1; else if (v > i) hi = mid - 1; else { mij = maskd_value(m, mid); break; }
        }
    } else {
        const int64_t p = j * m.vlen + i;
        mij = (m.kind == 2) ? (m.b[p] != 0) : true;
        if (mij) mij = maskd_value(m, p);
    }
    return mij != m.comp;
}

static gba_info make_maskd(const gba_matrix *M, bool comp, bool structural, MaskViewD *mv) {
    MaskViewD m{};
    if (!M) { *mv = m; return GBA_SUCCESS; }
    if (is_hyper(M)) GBA_FAIL(GBA_NOT_IMPLEMENTED, "hypersparse mask");
    if (is_sparse(M) && M->jumbled) GBA_FAIL(GBA_NOT_IMPLEMENTED, "jumbled sparse mask");
    if (M->nzombies) GBA_FAIL(GBA_INVALID_OBJECT, "mask has zombies");
    m.kind = is_sparse(M) ? 1 : (is_bitmap(M) ? 2 : 3);
    m.comp = comp; m.structural = structural; m.iso = M->iso; m.msize = (int)type_size(M->type);
    if (!structural && (m.msize == 0 || !M->x)) GBA_FAIL(GBA_INVALID_OBJECT, "valued mask without values");
    m.vlen = M->vlen; m.p = M->p; m.i = M->i; m.b = M->b; m.x = reinterpret_cast<const uint8_t *>(M->x);
    *mv = m;
    return GBA_SUCCESS;
}

// ---------------------------------------------------------------- dot2/dot4 --
// C(i,j) = reduce_k A(k,i) (x) B(k,j): A sparse (vector i lists k), B dense.
// MODE 0: dot4, Cx full updated in place.  MODE 1: dot2, Cb/Cx written.
// Each vector i is handled by `lpr` lanes; vectors longer than long_len are
// only queued for dot_long_kernel.
template <class S, int MODE>
__global__ void __launch_bounds__(256)
dot_sd_kernel(S sr, int64_t nvecA, const int64_t *__restrict__ Ap, const int64_t *__restrict__ Ai,
              const void *__restrict__ Ax_, bool A_iso, DenseView B, MaskViewD M, int lpr, int64_t long_len,
              typename S::z_t *__restrict__ Cx, int8_t *__restrict__ Cb, int64_t cvlen,
              unsigned int *__restrict__ nlong, int64_t *__restrict__ long_list) {
    using X = typename S::x_t; using Z = typename S::z_t;
    const X *Ax = reinterpret_cast<const X *>(Ax_);
    const X *Bx = reinterpret_cast<const X *>(B.x);
    const int64_t j = blockIdx.y;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = gid / lpr;
    const int lane = (int)(gid - i * lpr);
    if (i >= nvecA) return;                         // whole groups leave together (lpr divides 64)
    const int64_t pA0 = Ap[i], pA1 = Ap[i + 1];
    const int64_t pC = j * cvlen + i;
    if (pA1 - pA0 > long_len) {
        if (lane == 0 && j == 0) { unsigned int q = atomicAdd(nlong, 1u); long_list[q] = i; }
        return;
    }
    if (MODE == 1 && !maskd_allows(M, i, j)) { if (lane == 0) Cb[pC] = 0; return; }
    const bool apat = sr.a_is_pattern(), bpat = sr.b_is_pattern();
    const int64_t boff = j * B.vlen;
    Z acc = sr.identity();
    int found = 0;
    for (int64_t p = pA0 + lane; p < pA1; p += lpr) {
        const int64_t k = Ai[p];
        if (B.b && !B.b[boff + k]) continue;
        const X a = load_x<X>(Ax, p, A_iso, apat);
        const X b = load_x<X>(Bx, boff + k, B.iso, bpat);
        const Z t = sr.mult(a, b, i, k, j);        // C(i,j) += A'(i,k) (x) B(k,j)
        acc = found ? sr.add(acc, t) : t;
        found = 1;
    }
    group_reduce(sr, acc, found, lpr);
    if (lane == 0) {
        if (MODE == 0) { if (found) Cx[pC] = sr.add(Cx[pC], acc); }
        else { Cb[pC] = (int8_t)found; if (found) Cx[pC] = acc; }
    }
}

template <class S, int MODE>
__global__ void __launch_bounds__(256)
dot_long_kernel(S sr, const int64_t *__restrict__ Ap, const int64_t *__restrict__ Ai,
                const void *__restrict__ Ax_, bool A_iso, DenseView B, MaskViewD M,
                typename S::z_t *__restrict__ Cx, int8_t *__restrict__ Cb, int64_t cvlen,
                const unsigned int *__restrict__ nlong, const int64_t *__restrict__ long_list) {
    using X = typename S::x_t; using Z = typename S::z_t;
    __shared__ Z part[4];
    __shared__ int pfound[4];
    const X *Ax = reinterpret_cast<const X *>(Ax_);
    const X *Bx = reinterpret_cast<const X *>(B.x);
    const int64_t j = blockIdx.y;
    const unsigned int n = *nlong;
    const bool apat = sr.a_is_pattern(), bpat = sr.b_is_pattern();
    const int64_t boff = j * B.vlen;
    for (unsigned int q = blockIdx.x; q < n; q += gridDim.x) {
        const int64_t i = long_list[q];
        const int64_t pC = j * cvlen + i;
        if (MODE == 1 && !maskd_allows(M, i, j)) { if (threadIdx.x == 0) Cb[pC] = 0; continue; }
        const int64_t pA1 = Ap[i + 1];
        Z acc = sr.identity();
        int found = 0;
        for (int64_t p = Ap[i] + threadIdx.x; p < pA1; p += 256) {
            const int64_t k = Ai[p];
            if (B.b && !B.b[boff + k]) continue;
            const X a = load_x<X>(Ax, p, A_iso, apat);
            const X b = load_x<X>(Bx, boff + k, B.iso, bpat);
            const Z t = sr.mult(a, b, i, k, j);
            acc = found ? sr.add(acc, t) : t;
            found = 1;
        }
        group_reduce(sr, acc, found, 64);
        if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = acc; pfound[threadIdx.x >> 6] = found; }
        __syncthreads();
        if (threadIdx.x == 0) {
            Z r = sr.identity(); int f = 0;
            for (int w = 0; w < 4; w++) if (pfound[w]) { r = f ? sr.add(r, part[w]) : part[w]; f = 1; }
            if (MODE == 0) { if (f) Cx[pC] = sr.add(Cx[pC], r); }
            else { Cb[pC] = (int8_t)f; if (f) Cx[pC] = r; }
        }
        __syncthreads();
    }
}

// Entry-balanced version for monoids with a hardware-sized accumulator and multipliers that do not
// use the row index: every workgroup takes DOT_TILE consecutive ENTRIES of A (coalesced loads of
// Ai/Ax whatever the row lengths are -- RMAT rows range from 0 to 10^5 entries), puts the products
// in LDS, finds the rows its tile touches by bisection of Ap and reduces each row segment from LDS.
// A row that continues into a neighbouring tile is combined with the monoid's atomic.
// MODE 1 expects Cb zeroed and Cx filled with the monoid identity.
static constexpr int DOT_TILE_THREADS = 256, DOT_TILE_ITEMS = 8, DOT_TILE = DOT_TILE_THREADS * DOT_TILE_ITEMS;
template <class ACC> __global__ void fill_kernel(ACC *p, int64_t n, ACC v);

template <class S, int MODE>
__device__ inline void dot_commit(const S &sr, const MaskViewD &M, int64_t i, int64_t j, int64_t cvlen,
                                  typename S::z_t acc, bool shared, typename S::z_t *__restrict__ Cx,
                                  int8_t *__restrict__ Cb) {
    if (MODE == 1 && !maskd_allows(M, i, j)) return;
    const int64_t pC = j * cvlen + i;
    if (shared) sr.atomic(&Cx[pC], acc);
    else Cx[pC] = (MODE == 0) ? sr.add(Cx[pC], acc) : acc;
    if (MODE == 1) Cb[pC] = 1;
}

// tile_row[t] = the row that holds entry t * tile of A (the largest r with Ap[r] <= Ap[0] + t * tile):
// one pass over the row pointers, each row naming the tiles that start inside it.  A tile then learns its
// first row from one load instead of a search of Ap with several dependent probes.
__global__ void __launch_bounds__(256)
dot_tile_rows_kernel(int64_t nvec, const int64_t *__restrict__ Ap, int64_t tile, int64_t *__restrict__ tile_row) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t p0 = Ap[0];
    for (; r < nvec; r += stride) {
        const int64_t a0 = Ap[r] - p0, a1 = Ap[r + 1] - p0;
        if (a1 <= a0) continue;
        for (int64_t t = (a0 + tile - 1) / tile; t * tile < a1; t++) tile_row[t] = r;
    }
}

// out[e] = (int32) in[e] for the entries Ap[0] .. Ap[nvec) of a sparse matrix (a view may start inside its arrays)
__global__ void __launch_bounds__(256)
narrow_indices_kernel(const int64_t *__restrict__ Ap, int64_t nvec, const int64_t *__restrict__ in, int32_t *__restrict__ out) {
    const int64_t e0 = Ap[0], e1 = Ap[nvec];
    int64_t e = e0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; e < e1; e += stride) out[e - e0] = (int32_t)in[e];          // (entry e of the view is out[e - Ap[0]])
}

static constexpr int DOT_SEG_SERIAL = 32;       // row segments longer than this are reduced by a whole wave

// (IDX: the type the indices are read in -- int64_t as the caller holds them, or int32_t from the remembered copy)
template <class S, int MODE, class IDX>
__global__ void __launch_bounds__(DOT_TILE_THREADS)
dot_tile_kernel(S sr, int64_t nvecA, const int64_t *__restrict__ Ap, const IDX *__restrict__ Ai,
                const void *__restrict__ Ax_, bool A_iso, DenseView B, MaskViewD M,
                typename S::z_t *__restrict__ Cx, int8_t *__restrict__ Cb, int64_t cvlen,
                const int64_t *__restrict__ tile_row) {
    using X = typename S::x_t; using Z = typename S::z_t;
    __shared__ Z prod[DOT_TILE];
    __shared__ uint8_t have[DOT_TILE];
    __shared__ Z part[DOT_TILE_THREADS / 64];
    __shared__ int pfound[DOT_TILE_THREADS / 64];
    const X *Ax = reinterpret_cast<const X *>(Ax_);
    const X *Bx = reinterpret_cast<const X *>(B.x);
    const int tid = threadIdx.x;
    const int64_t j = blockIdx.y;
    const int64_t pN = Ap[nvecA];
    const int64_t start = Ap[0] + (int64_t)blockIdx.x * DOT_TILE;
    if (start >= pN) return;                                  // uniform
    if (sizeof(IDX) == 4) Ai -= Ap[0];                        // the 32-bit copy holds the view's own entries only
    const int64_t end = (start + DOT_TILE < pN) ? start + DOT_TILE : pN;
    const int64_t r0 = tile_row[blockIdx.x];                  // row holding the first entry of the tile
    __shared__ int longq[DOT_TILE / DOT_SEG_SERIAL + 2];
    __shared__ int nlongq;
    if (tid == 0) nlongq = 0;
    const bool apat = sr.a_is_pattern(), bpat = sr.b_is_pattern();
    const int64_t boff = j * B.vlen;
    const bool plain = !B.b;                                  // every entry of B present: have[] is not used
    // the row pointers the reduction needs, asked for before the entries so that they arrive with them
    const int64_t next0 = Ap[r0 + 1];
    const int64_t rmine = r0 + tid;
    int64_t a0_first = 0, a1_first = 0;
    if (rmine < nvecA) { a0_first = Ap[rmine]; a1_first = Ap[rmine + 1]; }
    if (plain && end - start == DOT_TILE) {
        // whole tile: all index loads in flight together, then all gathers (one wait each, not one per entry)
        int64_t k[DOT_TILE_ITEMS]; X av[DOT_TILE_ITEMS], bv[DOT_TILE_ITEMS];
#pragma unroll
        for (int it = 0; it < DOT_TILE_ITEMS; it++)           // streamed once: keep the cache for B
            k[it] = (int64_t)__builtin_nontemporal_load(&Ai[start + it * DOT_TILE_THREADS + tid]);
#pragma unroll
        for (int it = 0; it < DOT_TILE_ITEMS; it++)
            av[it] = apat ? X() : __builtin_nontemporal_load(&Ax[A_iso ? 0 : start + it * DOT_TILE_THREADS + tid]);
#pragma unroll
        for (int it = 0; it < DOT_TILE_ITEMS; it++) bv[it] = load_x<X>(Bx, boff + k[it], B.iso, bpat);
#pragma unroll
        for (int it = 0; it < DOT_TILE_ITEMS; it++) prod[it * DOT_TILE_THREADS + tid] = sr.mult(av[it], bv[it], 0, k[it], j);
    } else {
#pragma unroll
        for (int it = 0; it < DOT_TILE_ITEMS; it++) {
            const int idx = it * DOT_TILE_THREADS + tid;
            const int64_t p = start + idx;
            if (p < end) {
                const int64_t k = (int64_t)__builtin_nontemporal_load(&Ai[p]);
                const bool ok = plain || B.b[boff + k];
                Z t = Z();
                if (ok) t = sr.mult(((apat) ? X() : __builtin_nontemporal_load(&Ax[A_iso ? 0 : p])), load_x<X>(Bx, boff + k, B.iso, bpat), 0, k, j);
                prod[idx] = t;
                have[idx] = ok;
            }
        }
    }
    __syncthreads();
    if (next0 >= end) {                                       // the tile lies inside one long row
        Z acc = sr.identity(); int found = 0;
        for (int idx = tid; idx < (int)(end - start); idx += DOT_TILE_THREADS)
            if (plain || have[idx]) { acc = found ? sr.add(acc, prod[idx]) : prod[idx]; found = 1; }
        group_reduce(sr, acc, found, 64);
        if ((tid & 63) == 0) { part[tid >> 6] = acc; pfound[tid >> 6] = found; }
        __syncthreads();
        if (tid == 0) {
            Z r = sr.identity(); int f = 0;
            for (int w = 0; w < DOT_TILE_THREADS / 64; w++) if (pfound[w]) { r = f ? sr.add(r, part[w]) : part[w]; f = 1; }
            const bool shared = a0_first < start || next0 > end;      // thread 0 holds Ap[r0]
            if (f) dot_commit<S, MODE>(sr, M, r0, j, cvlen, r, shared, Cx, Cb);
        }
        return;
    }
    // one thread per row for short segments; long ones (a hub row inside the tile) are queued for whole waves
    for (int64_t r = rmine; r < nvecA; r += DOT_TILE_THREADS) {
        const int64_t a0 = (r == rmine) ? a0_first : Ap[r], a1 = (r == rmine) ? a1_first : Ap[r + 1];
        if (a0 >= end) break;                                 // past the last row of the tile
        const int lo = (int)((a0 > start ? a0 : start) - start), hi = (int)((a1 < end ? a1 : end) - start);
        if (hi - lo > DOT_SEG_SERIAL) { longq[atomicAdd(&nlongq, 1)] = (int)(r - r0); continue; }
        Z acc = sr.identity(); int found = 0;
        for (int idx = lo; idx < hi; idx++)
            if (plain || have[idx]) { acc = found ? sr.add(acc, prod[idx]) : prod[idx]; found = 1; }
        if (found) dot_commit<S, MODE>(sr, M, r, j, cvlen, acc, a0 < start || a1 > end, Cx, Cb);
    }
    __syncthreads();
    const int nq = nlongq, lane = tid & 63;
    for (int q = tid >> 6; q < nq; q += DOT_TILE_THREADS / 64) {
        const int64_t r = r0 + longq[q];
        const int64_t a0 = Ap[r], a1 = Ap[r + 1];
        const int lo = (int)((a0 > start ? a0 : start) - start), hi = (int)((a1 < end ? a1 : end) - start);
        Z acc = sr.identity(); int found = 0;
        for (int idx = lo + lane; idx < hi; idx += 64)
            if (plain || have[idx]) { acc = found ? sr.add(acc, prod[idx]) : prod[idx]; found = 1; }
        group_reduce(sr, acc, found, 64);
        if (lane == 0 && found) dot_commit<S, MODE>(sr, M, r, j, cvlen, acc, a0 < start || a1 > end, Cx, Cb);
    }
}

// does the semiring fit dot_tile_kernel?
template <class S> struct dot_tile_ok {
    static constexpr bool v = !S::positional && sizeof(typename S::z_t) >= 4 &&
                              std::is_same<typename S::z_t, typename S::acc_t>::value;
};

static gba_info count_ones(const int8_t *b, int64_t n, int64_t *out) { return dev_count_ones(b, n, out); }

static int pick_group(const gba_matrix *A) {
    const double avg = (A->nvec > 0) ? (double)A->nvals / (double)A->nvec : 1.0;
    int g = 2;
    while (g < 64 && (double)g < avg) g <<= 1;
    return g;
}

template <int MODE>
static gba_info dot_sparse_dense(gba_matrix *C, const gba_matrix *M, bool Mask_comp, bool Mask_struct,
                                 const gba_matrix *A, const gba_matrix *B, const Canon &c, gba_report *rep) {
    GBA_OK(require_init());
    if (!is_sparse(A) || !is_dense_fmt(B))
        GBA_FAIL(GBA_NOT_IMPLEMENTED, "dot%d on device: A must be sparse and B bitmap/full (got %d, %d)",
                 MODE == 0 ? 4 : 2, A->sparsity, B->sparsity);
    if (A->jumbled && !(c.add == GBA_ANY)) { /* order only matters for fp rounding; any order is a valid sum */ }
    if (A->vlen != B->vlen) GBA_FAIL(GBA_DIMENSION_MISMATCH, "dot: A->vlen %ld != B->vlen %ld", (long)A->vlen, (long)B->vlen);
    if (A->nzombies) GBA_FAIL(GBA_INVALID_OBJECT, "dot: A has zombies");
    hipStream_t st = ctx().stream;
    const int64_t cvlen = A->vdim, cvdim = B->vdim;
    if (cvdim > 65535) GBA_FAIL(GBA_NO_VALUE, "dot: more than 65535 dense vectors (declined: the caller takes the general path)");
    MaskViewD mv; GBA_OK(make_maskd(M, Mask_comp, Mask_struct, &mv));
    DevBuf Cb, Cx;
    const size_t zsize = type_size(c.ztype);
    if (MODE == 0) {
        if (!is_full(C) || C->type != c.ztype || C->vlen != cvlen || C->vdim != cvdim || C->iso)
            GBA_FAIL(GBA_INVALID_VALUE, "dot4: C must be a full non-iso matrix of the monoid type, %ld-by-%ld", (long)cvlen, (long)cvdim);
    } else {
        GBA_OK(Cb.alloc((size_t)(cvlen * cvdim > 0 ? cvlen * cvdim : 1)));
        GBA_OK(Cx.alloc((size_t)(cvlen * cvdim > 0 ? cvlen * cvdim : 1) * zsize));
    }
    DevBuf nlong, long_list, tile_rows;
    GBA_OK(nlong.alloc(4));
    GBA_OK(long_list.alloc((size_t)(A->nvec > 0 ? A->nvec : 1) * 8));
    GBA_HIP(hipMemsetAsync(nlong.p, 0, 4, st));
    KernelTimer timer(rep != nullptr);
    if (cvlen > 0 && cvdim > 0) {
        const int lpr = pick_group(A);
        const int64_t long_len = 8192;
        DenseView Bv{B->x, is_bitmap(B) ? B->b : nullptr, B->vlen, B->iso};
        void *cx = (MODE == 0) ? C->x : Cx.p;
        int8_t *cb = (MODE == 0) ? nullptr : Cb.as<int8_t>();
        GBA_OK(with_semiring(c, [&](auto sr) -> gba_info {
            using S = decltype(sr); using Z = typename S::z_t;
            if constexpr (dot_tile_ok<S>::v) {
                if (A->nvals > 0) {
                    const int64_t cn = cvlen * cvdim;
                    if (MODE == 1) {
                        GBA_HIP(hipMemsetAsync(cb, 0, (size_t)cn, st));
                        int64_t fb = ceil_div(cn, 256); if (fb > 8192) fb = 8192;
                        hipLaunchKernelGGL(HIP_KERNEL_NAME(fill_kernel<Z>), dim3((unsigned)fb), dim3(256), 0, st,
                                           reinterpret_cast<Z *>(cx), cn, sr.identity());
                    }
                    const int64_t ntiles = ceil_div(A->nvals, DOT_TILE);
                    int64_t rb = ceil_div(A->nvec, 256); if (rb > 8192) rb = 8192; if (rb < 1) rb = 1;
                    dim3 tgrid((unsigned)ntiles, (unsigned)cvdim);
                    // A pattern seen before (an iteration multiplies by the same matrix again and again): the tile table
                    // and a 32-bit copy of the indices are built once and kept; the kernel then streams 4-byte indices.
                    PatternAux *aux = pattern_aux(A);
                    const int32_t *i32 = nullptr;
                    const int64_t *trow = nullptr;
                    if (aux && aux->sightings >= 2) {
                        if (!aux->tile_row[0] || aux->tile_size[0] != DOT_TILE) {
                            if (aux->tile_row[0]) { gba_free(aux->tile_row[0]); aux->tile_row[0] = nullptr; }
                            aux->tile_row[0] = reinterpret_cast<int64_t *>(gba_malloc((size_t)ntiles * 8));
                            if (aux->tile_row[0]) {
                                aux->tile_size[0] = DOT_TILE;
                                hipLaunchKernelGGL(dot_tile_rows_kernel, dim3((unsigned)rb), dim3(256), 0, st, A->nvec, A->p,
                                                   (int64_t)DOT_TILE, aux->tile_row[0]);
                            }
                        }
                        trow = aux->tile_row[0];
                        if (!aux->i32 && A->vlen < (int64_t)0x7fffffff && A->nvals <= PATTERN_I32_MAX_ENTRIES) {
                            aux->i32 = reinterpret_cast<int32_t *>(gba_malloc((size_t)A->nvals * 4));
                            if (aux->i32) {
                                int64_t nb = ceil_div(A->nvals, 1024); if (nb > 16384) nb = 16384;
                                hipLaunchKernelGGL(narrow_indices_kernel, dim3((unsigned)nb), dim3(256), 0, st, A->p, A->nvec, A->i, aux->i32);
                            }
                        }
                        i32 = aux->i32;
                    }
                    if (!trow) {
                        GBA_OK(tile_rows.alloc((size_t)ntiles * 8));
                        hipLaunchKernelGGL(dot_tile_rows_kernel, dim3((unsigned)rb), dim3(256), 0, st, A->nvec, A->p,
                                           (int64_t)DOT_TILE, tile_rows.as<int64_t>());
                        trow = tile_rows.as<int64_t>();
                    }
                    if (i32) {
                        hipLaunchKernelGGL(HIP_KERNEL_NAME(dot_tile_kernel<S, MODE, int32_t>), tgrid, dim3(DOT_TILE_THREADS), 0, st, sr,
                                           A->nvec, A->p, i32, A->x, A->iso, Bv, mv, reinterpret_cast<Z *>(cx), cb, cvlen, trow);
                        if (rep) rep->index_bytes = 4;          // bytes per index the kernel streamed
                    } else {
                        hipLaunchKernelGGL(HIP_KERNEL_NAME(dot_tile_kernel<S, MODE, int64_t>), tgrid, dim3(DOT_TILE_THREADS), 0, st, sr,
                                           A->nvec, A->p, A->i, A->x, A->iso, Bv, mv, reinterpret_cast<Z *>(cx), cb, cvlen, trow);
                        if (rep) rep->index_bytes = 8;
                    }
                } else if (MODE == 1) {
                    GBA_HIP(hipMemsetAsync(cb, 0, (size_t)(cvlen * cvdim), st));
                }
                GBA_HIP(hipGetLastError());
                return GBA_SUCCESS;
            }
            dim3 grid((unsigned)ceil_div(A->nvec * lpr, 256), (unsigned)cvdim);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(dot_sd_kernel<S, MODE>), grid, dim3(256), 0, st, sr, A->nvec, A->p, A->i,
                               A->x, A->iso, Bv, mv, lpr, long_len, reinterpret_cast<Z *>(cx), cb, cvlen,
                               nlong.as<unsigned int>(), long_list.as<int64_t>());
            hipLaunchKernelGGL(HIP_KERNEL_NAME(dot_long_kernel<S, MODE>), dim3(512, (unsigned)cvdim), dim3(256), 0, st, sr,
                               A->p, A->i, A->x, A->iso, Bv, mv, reinterpret_cast<Z *>(cx), cb, cvlen,
                               nlong.as<unsigned int>(), long_list.as<int64_t>());
            GBA_HIP(hipGetLastError());
            return GBA_SUCCESS;
        }));
    }
    if (rep) rep->kernel_ms = timer.stop_ms();
    if (MODE == 1) {
        int64_t nvals = 0;
        GBA_OK(count_ones(Cb.as<int8_t>(), cvlen * cvdim, &nvals));
        C->vlen = cvlen; C->vdim = cvdim; C->nvec = cvdim; C->nvals = nvals;
        C->p = nullptr; C->h = nullptr; C->i = nullptr;
        C->b = Cb.as<int8_t>(); Cb.release();
        C->x = Cx.release();
        C->nzombies = 0; C->type = c.ztype; C->sparsity = GBA_BITMAP; C->mem = GBA_MEM_DEVICE;
        C->iso = false; C->jumbled = false; C->is_csc = B->is_csc;
    } else {
        GBA_HIP(hipStreamSynchronize(st));
    }
    return GBA_SUCCESS;
}

gba_info dot2(gba_matrix *C, const gba_matrix *M, bool Mask_comp, bool Mask_struct,
              const gba_matrix *A, const gba_matrix *B, const Canon &c, gba_report *rep) {
    if (M && (M->vlen != A->vdim || M->vdim != B->vdim)) GBA_FAIL(GBA_DIMENSION_MISMATCH, "dot2: mask dimensions");
    return dot_sparse_dense<1>(C, M, Mask_comp, Mask_struct, A, B, c, rep);
}
gba_info dot4(gba_matrix *C, const gba_matrix *A, const gba_matrix *B, const Canon &c, gba_report *rep) {
    if (c.add == GBA_ANY) return GBA_NO_VALUE;      // like the CPU (Source/GB_AxB_dot4.c): ANY is not an accumulator
    return dot_sparse_dense<0>(C, nullptr, false, false, A, B, c, rep);
}

// ------------------------------------------------------------ saxpy4/saxbit --
// C(i,j) (+)= A(i,k) (x) B(k,j): A sparse (vector k lists i), B dense.
// MODE 0: saxpy4, C full.  MODE 1: saxbit, C bitmap with Cx pre-set to the
// monoid identity and Cb raised by any contributing product.
template <class S, int MODE>
__global__ void __launch_bounds__(256)
saxpy_sd_kernel(S sr, int64_t nvecA, const int64_t *__restrict__ Ap, const int64_t *__restrict__ Ai,
                const void *__restrict__ Ax_, bool A_iso, DenseView B, MaskViewD M, int lpr,
                typename S::acc_t *__restrict__ Cx, int8_t *__restrict__ Cb, int64_t cvlen) {
    using X = typename S::x_t; using Z = typename S::z_t;
    const X *Ax = reinterpret_cast<const X *>(Ax_);
    const X *Bx = reinterpret_cast<const X *>(B.x);
    const int64_t j = blockIdx.y;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t k = gid / lpr;
    const int lane = (int)(gid - k * lpr);
    if (k >= nvecA) return;
    const int64_t pB = j * B.vlen + k;
    if (B.b && !B.b[pB]) return;
    const bool apat = sr.a_is_pattern(), bpat = sr.b_is_pattern();
    const X b = load_x<X>(Bx, pB, B.iso, bpat);
    const int64_t pA1 = Ap[k + 1];
    const int64_t coff = j * cvlen;
    for (int64_t p = Ap[k] + lane; p < pA1; p += lpr) {
        const int64_t i = Ai[p];
        if (MODE == 1 && !maskd_allows(M, i, j)) continue;
        const X a = load_x<X>(Ax, p, A_iso, apat);
        const Z t = sr.mult(a, b, i, k, j);
        sr.atomic(&Cx[coff + i], t);
        if (MODE == 1) Cb[coff + i] = 1;
    }
}

static unsigned grid_for(int64_t n) { int64_t b = ceil_div(n, 256); if (b > 8192) b = 8192; if (b < 1) b = 1; return (unsigned)b; }

// ---- saxpy4 by destination bins ------------------------------------------------------------------
// The scatter above sends one atomic per product to memory; atomics are carried out at the memory side
// (about 6e9 a second, far fewer on a popular destination), which leaves y' += x'*A of a large graph two
// orders of magnitude short of the memory rate.  The reference keeps a private workspace per task and
// reduces the workspaces afterwards (Source/Template/GB_AxB_saxpy4_template.c:429-445); the same idea
// here, shaped for the memory system:
//   1  bin_count     every tile of BIN_TILE entries counts its destinations per bin of 2^shift indices
//   2  (prefix sum of the bin-major count table = where each tile writes inside each bin)
//   3  bin_scatter   the tile forms its products, sorts them by bin in LDS and writes (index, value) pairs
//                    in runs, so every bin ends up contiguous
//   4  bin_reduce    one workgroup per chunk of a bin sums the pairs into an LDS panel (LDS atomics) and
//                    adds the panel to C, or parks it when the bin has several chunks
//   5  bin_combine   sums the parked panels of such bins into C
// Every byte moves in streams; no atomic leaves the CU.
static constexpr int BIN_THREADS = 512;         // a tile is BIN_THREADS * ITEMS entries
template <class S> struct bin_items { static constexpr int v = 8; };
static constexpr int BIN_MAX = 2048;            // bins
static constexpr int BIN_CHUNK = 32768;         // pairs per bin_reduce workgroup
static constexpr int BIN_WALK = 128;            // tiles one bin_scatter workgroup walks at most
static constexpr int BIN_REDUCE_THREADS = 1024; // two such workgroups fill a CU: the panels take 64 KB of LDS each
// The LDS panels hold floats as doubles: on gfx950 ds_add_f32 handles one lane at a time (80 ns per wave
// instruction against 9 ns for ds_add_f64 and 6 ns for ds_add_u32, measured with random addresses), so a
// float sum through LDS atomics is ten times faster in fp64, and rounds once at the end.
template <class Z> struct bin_panel { using type = Z; };
template <> struct bin_panel<float> { using type = double; };
// Bins of 2^14 destinations whatever the type (64 KB panels of 4-byte sums, 128 KB of 8-byte ones): the
// narrower the bins, the shorter the runs bin_scatter writes, and short runs are what it pays for.
template <class Z> struct bin_shift { static constexpr int v = 14; };

#ifdef GBA_PANEL_TICKS                            // tuning build: per-phase stopwatch of bin_scatter (shader clocks)
__device__ unsigned long long bin_ticks[16];
#define BIN_TICK0 unsigned long long tk_ = __builtin_amdgcn_s_memtime()
#define BIN_TICK(n) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        atomicAdd(&bin_ticks[n], t_ - tk_); tk_ = t_; } } while (0)
#else
#define BIN_TICK0
#define BIN_TICK(n)
#endif

// Loads that the compiler does not see as memory instructions, each followed by its own wait.  For rarely
// taken paths inside a software-pipelined loop: a compiler-visible load there makes every later register
// reuse wait for all outstanding loads (it cannot tell whether that path ran), the prefetch included.
__device__ inline void load_i64_pair_blocking(const int64_t *p, int64_t &lo, int64_t &hi) {
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    u4 r;
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p) : "memory");
    lo = (int64_t)(((uint64_t)r.y << 32) | r.x);
    hi = (int64_t)(((uint64_t)r.w << 32) | r.z);
}
template <class T> __device__ inline T load_blocking(const T *p) {
    if constexpr (sizeof(T) == 4) {
        unsigned int r;
        asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p) : "memory");
        T t; __builtin_memcpy(&t, &r, 4); return t;
    } else if constexpr (sizeof(T) == 8) {
        unsigned long long r;
        asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p) : "memory");
        T t; __builtin_memcpy(&t, &r, 8); return t;
    } else {
        return *p;
    }
}

__device__ inline int64_t ceil_div_dev(int64_t a, int64_t b) { return (a + b - 1) / b; }

// workgroup barrier that waits for this wave's LDS traffic only: loads from memory issued earlier (a
// prefetch for the next tile) stay in flight across it, which __syncthreads() would not allow
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// exclusive prefix sum of v[0..n) in LDS by all NT threads of the workgroup (wsum: NT / 64 ints); v must be
// complete (barrier before the call); returns the total after a closing barrier
template <int NT>
__device__ inline int block_scan_lds(int *v, int n, int *wsum) {
    const int tid = threadIdx.x, per = (n + NT - 1) / NT;
    const int lo = tid * per, hi = (lo + per < n) ? lo + per : n;
    int s = 0;
    for (int q = lo; q < hi; q++) s += v[q];
    int inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d, 64); if ((tid & 63) >= d) inc += t; }
    if ((tid & 63) == 63) wsum[tid >> 6] = inc;
    lds_barrier();
    int run = inc - s, total = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; w++) { if (w < (tid >> 6)) run += wsum[w]; total += wsum[w]; }
    for (int q = lo; q < hi; q++) { const int c = v[q]; v[q] = run; run += c; }
    lds_barrier();
    return total;
}

// One workgroup counts BIN_COUNT_TILES consecutive tiles, so that for every bin it writes that many adjacent
// cells of the bin-major table at once (a full 64-byte line) instead of one 8-byte cell per tile.
static constexpr int BIN_COUNT_TILES = 8;
template <int BIN_ITEMS>
__global__ void __launch_bounds__(BIN_THREADS)
bin_count_kernel(int64_t nvecA, const int64_t *__restrict__ Ap, const int64_t *__restrict__ Ai, int shift, int nb,
                 int64_t nwg, int64_t *__restrict__ table) {
    constexpr int BIN_TILE = BIN_THREADS * BIN_ITEMS;
    extern __shared__ int bin_dyn[];
    int *h = bin_dyn;                                         // h[t * nb + b]: neighbouring bins in neighbouring banks
    const int tid = threadIdx.x;
    const int64_t wg0 = (int64_t)blockIdx.x * BIN_COUNT_TILES;
    const int64_t p0 = Ap[0], pN = Ap[nvecA];
    const int *Ai_lo = reinterpret_cast<const int *>(Ai);    // indices are below 2^31 here
    for (int q = tid; q < nb * BIN_COUNT_TILES; q += BIN_THREADS) h[q] = 0;
    __syncthreads();
    for (int t = 0; t < BIN_COUNT_TILES; t++) {
        const int64_t start = p0 + (wg0 + t) * BIN_TILE;
        if (start >= pN) break;                               // uniform
        int k[BIN_ITEMS];
#pragma unroll
        for (int it = 0; it < BIN_ITEMS; it++) {
            const int64_t p = start + it * BIN_THREADS + tid;
            k[it] = (p < pN) ? __builtin_nontemporal_load(&Ai_lo[2 * p]) : -1;
        }
#pragma unroll
        for (int it = 0; it < BIN_ITEMS; it++) if (k[it] >= 0) atomicAdd(&h[t * nb + (k[it] >> shift)], 1);
    }
    __syncthreads();
    for (int q = tid; q < nb * BIN_COUNT_TILES; q += BIN_THREADS) {
        const int b = q / BIN_COUNT_TILES, t = q - b * BIN_COUNT_TILES;
        if (wg0 + t < nwg) table[(int64_t)b * nwg + wg0 + t] = h[t * nb + b];      // bin-major
    }
}

// A workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...; while it sorts one tile in LDS the loads of
// its next tile are already in flight (the memory round trip is several microseconds under load, longer
// than the LDS work on a tile, and too few workgroups fit a CU to hide it by occupancy alone).
template <class S, int BIN_ITEMS>
__global__ void __launch_bounds__(BIN_THREADS, 4)      // four waves per SIMD = two workgroups per CU: at most 128 registers
bin_scatter_kernel(S sr, int64_t nvecA, const int64_t *__restrict__ Ap, const int64_t *__restrict__ Ai,
                   const void *__restrict__ Ax_, bool A_iso, DenseView B, int64_t j,
                   const int64_t *__restrict__ tile_row, int shift, int nb, int64_t nwg,
                   const int64_t *__restrict__ table, uint16_t *__restrict__ pair_idx,
                   typename S::z_t *__restrict__ pair_val, const int32_t *__restrict__ narrow) {
    using X = typename S::x_t; using Z = typename S::z_t;
    constexpr int BIN_TILE = BIN_THREADS * BIN_ITEMS;
    constexpr int NLONG = BIN_TILE / DOT_SEG_SERIAL + 2, NGB = BIN_MAX / BIN_THREADS;
    extern __shared__ int bin_dyn[];
    int *lstart = bin_dyn, *delta = bin_dyn + nb;            // per bin: first slot in the sorted tile; global - local
    __shared__ int wsum[BIN_THREADS / 64];
    __shared__ int scol[BIN_TILE];
    __shared__ __align__(8) unsigned char vbuf[BIN_TILE * (sizeof(X) > sizeof(Z) ? sizeof(X) : sizeof(Z))];
    __shared__ int longlo[NLONG], longhi[NLONG];
    __shared__ X longx[NLONG];
    __shared__ int nlongq;
    X *xs = reinterpret_cast<X *>(vbuf);                      // B(k,j) per entry, then (after a barrier) ...
    Z *sval = reinterpret_cast<Z *>(vbuf);                    // ... the products in bin order
    const X *Ax = reinterpret_cast<const X *>(Ax_);
    const X *Bx = reinterpret_cast<const X *>(B.x);
    const int tid = threadIdx.x;
    // Which tiles this workgroup walks.  Workgroups go to the eight XCDs in turn (blockIdx.x % 8), each XCD
    // with its own L2: the tiles are cut into eight contiguous ranges, one per XCD, so that neighbouring
    // tiles -- whose runs are neighbours inside every bin -- complete each other's cache lines in one L2.
    int64_t tile, G, tile_end;
    if (gridDim.x % 8 == 0) {
        const int64_t per_xcd = ceil_div_dev(nwg, 8), xcd = blockIdx.x & 7;
        tile = xcd * per_xcd + (blockIdx.x >> 3);
        G = gridDim.x >> 3;
        tile_end = (xcd + 1) * per_xcd < nwg ? (xcd + 1) * per_xcd : nwg;
    } else {
        tile = blockIdx.x; G = gridDim.x; tile_end = nwg;
    }
    if (tile >= tile_end) return;                             // uniform
    const int64_t p0 = Ap[0], pN = Ap[nvecA];
    const bool apat = sr.a_is_pattern(), bpat = sr.b_is_pattern();
    const int64_t boff = j * B.vlen;
    const int low = (1 << shift) - 1;
    // the first vector of each tile this workgroup will walk, fetched once: the loads of a tile start from it
    __shared__ int64_t myrow[BIN_WALK];
    for (int64_t k = tid; k < BIN_WALK; k += BIN_THREADS) {
        const int64_t t = tile + k * G;
        myrow[k] = (t < tile_end) ? tile_row[t] : 0;
    }
    __syncthreads();
    // indices are below 2^31 here: only their low words are loaded (a load whose upper half is dead makes the
    // compiler reuse that register at once, and the reuse then has to wait for the load)
    // (`narrow`: the 32-bit copy of the indices kept with a pattern seen before, entry e at narrow[e - Ap[0]])
    const int *Ai_lo = narrow ? narrow - p0 : reinterpret_cast<const int *>(Ai);
    const int istride = narrow ? 1 : 2;
    const int64_t isomask = A_iso ? 0 : -1;                   // an iso A reads Ax[0] through the same vector loads
    auto issue = [&](int64_t t, int64_t tr0, int (&c)[BIN_ITEMS], X (&a)[BIN_ITEMS], int64_t &a0, int64_t &a1, X &xv) {
        const int64_t st = p0 + t * BIN_TILE;
        const int64_t rm = tr0 + tid, rc = (rm < nvecA) ? rm : nvecA - 1;
        a0 = Ap[rc]; a1 = Ap[rc + 1]; xv = load_x<X>(Bx, boff + rc, B.iso, bpat);
#pragma unroll
        for (int it = 0; it < BIN_ITEMS; it++) {
            const int64_t p = st + it * BIN_THREADS + tid;
            c[it] = __builtin_nontemporal_load(&Ai_lo[istride * (p < pN ? p : pN - 1)]);
        }
        if (!apat) {
#pragma unroll
            for (int it = 0; it < BIN_ITEMS; it++) {
                const int64_t p = st + it * BIN_THREADS + tid;
                a[it] = __builtin_nontemporal_load(&Ax[(p < pN ? p : pN - 1) & isomask]);
            }
        } else {
#pragma unroll
            for (int it = 0; it < BIN_ITEMS; it++) a[it] = X();
        }
    };

    int walk = 0;
    int col[BIN_ITEMS], ncol[BIN_ITEMS]; X av[BIN_ITEMS], nav[BIN_ITEMS];
    int64_t a0_first, a1_first, na0_first, na1_first; X x_first, nx_first;
    int64_t r0 = myrow[0];
    issue(tile, r0, col, av, a0_first, a1_first, x_first);
    while (true) {
        BIN_TICK0;
        const int64_t start = p0 + tile * BIN_TILE;
        const int64_t end = (start + BIN_TILE < pN) ? start + BIN_TILE : pN;
        const int ntile = (int)(end - start);
        const bool more = tile + G < tile_end;
        // this tile's place in every bin (needed late), then the next tile's loads: in that order, because a
        // wait for a load also waits for every load issued before it.  The last tile fetches itself again.
        int gb[NGB];
#pragma unroll
        for (int u = 0; u < NGB; u++) {
            const int b = u * BIN_THREADS + tid;
            gb[u] = (int)table[(int64_t)(b < nb ? b : nb - 1) * nwg + tile];
        }
        const int64_t nr0 = myrow[more ? walk + 1 : walk];
        issue(more ? tile + G : tile, nr0, ncol, nav, na0_first, na1_first, nx_first);
        for (int b = tid; b < nb; b += BIN_THREADS) lstart[b] = 0;
        if (tid == 0) nlongq = 0;
        lds_barrier();
        BIN_TICK(0);
        // xs[idx] = B(k,j) of the vector k that owns entry idx: a thread per vector, whole waves for long
        // segments.  The first vector of each thread comes from the registers loaded a tile ago.
        const int64_t rmine = r0 + tid;
        auto spread = [&](int64_t a0, int64_t a1, X xv) {
            const int lo = (int)((a0 > start ? a0 : start) - start), hi = (int)((a1 < end ? a1 : end) - start);
            if (hi <= lo) return;
            if (hi - lo > DOT_SEG_SERIAL) {
                const int q = atomicAdd(&nlongq, 1);
                longlo[q] = lo; longhi[q] = hi; longx[q] = xv;
                return;
            }
            for (int idx = lo; idx < hi; idx++) xs[idx] = xv;
        };
        if (rmine < nvecA && a0_first < end) spread(a0_first, a1_first, x_first);
        if (rmine + BIN_THREADS < nvecA && a1_first < end) {           // a tile of more than BIN_THREADS vectors
            for (int64_t r = rmine + BIN_THREADS; r < nvecA; r += BIN_THREADS) {
                int64_t a0, a1;
                load_i64_pair_blocking(Ap + r, a0, a1);
                if (a0 >= end) break;
                spread(a0, a1, bpat ? X() : load_blocking<X>(&Bx[B.iso ? 0 : boff + r]));
            }
        }
        lds_barrier();
        BIN_TICK(1);
        {
            const int nq = nlongq, lane = tid & 63;
            for (int q = tid >> 6; q < nq; q += BIN_THREADS / 64) {
                const int hi = longhi[q];
                const X xv = longx[q];
                for (int idx = longlo[q] + lane; idx < hi; idx += 64) xs[idx] = xv;
            }
        }
        lds_barrier();
        BIN_TICK(2);
        int rank[BIN_ITEMS]; Z pv[BIN_ITEMS];
#pragma unroll
        for (int it = 0; it < BIN_ITEMS; it++) {
            rank[it] = 0; pv[it] = Z();
            if (it * BIN_THREADS + tid < ntile) {
                rank[it] = atomicAdd(&lstart[col[it] >> shift], 1);
                pv[it] = sr.mult(av[it], xs[it * BIN_THREADS + tid], 0, 0, j);
            }
        }
        lds_barrier();
        BIN_TICK(3);
        block_scan_lds<BIN_THREADS>(lstart, nb, wsum);
        BIN_TICK(4);
#pragma unroll
        for (int u = 0; u < NGB; u++) {
            const int b = u * BIN_THREADS + tid;
            if (b < nb) delta[b] = gb[u] - lstart[b];
        }
#pragma unroll
        for (int it = 0; it < BIN_ITEMS; it++)
            if (it * BIN_THREADS + tid < ntile) { const int q = lstart[col[it] >> shift] + rank[it]; scol[q] = col[it]; sval[q] = pv[it]; }
        lds_barrier();
        BIN_TICK(5);
        for (int q = tid; q < ntile; q += BIN_THREADS) {
            const int c = scol[q];
            const int64_t dst = (int64_t)delta[c >> shift] + q;
            pair_idx[dst] = (uint16_t)(c & low);
            pair_val[dst] = sval[q];
        }
        BIN_TICK(6);
        if (!more) break;
        lds_barrier();                                        // the next tile reuses every LDS array
        tile += G; walk++; r0 = nr0;
        a0_first = na0_first; a1_first = na1_first; x_first = nx_first;
#pragma unroll
        for (int it = 0; it < BIN_ITEMS; it++) { col[it] = ncol[it]; av[it] = nav[it]; }
    }
}

// chunk_base[b] = number of bin_reduce workgroups of the bins before b; chunk_base[nb] = all of them
__global__ void __launch_bounds__(256)
bin_chunks_kernel(int nb, int64_t nwg, const int64_t *__restrict__ table, int *__restrict__ chunk_base) {
    __shared__ int v[BIN_MAX];
    __shared__ int wsum[4];
    for (int b = threadIdx.x; b < nb; b += 256) {
        const int64_t cnt = table[(int64_t)(b + 1) * nwg] - table[(int64_t)b * nwg];
        v[b] = (int)((cnt + BIN_CHUNK - 1) / BIN_CHUNK);
    }
    __syncthreads();
    const int total = block_scan_lds<256>(v, nb, wsum);
    for (int b = threadIdx.x; b < nb; b += 256) chunk_base[b] = v[b];
    if (threadIdx.x == 0) chunk_base[nb] = total;
}

template <class S>
__global__ void __launch_bounds__(BIN_REDUCE_THREADS)
bin_reduce_kernel(S sr, int shift, int nb, int64_t nwg, const int64_t *__restrict__ table,
                  const int *__restrict__ chunk_base, const uint16_t *__restrict__ pair_idx,
                  const typename S::z_t *__restrict__ pair_val, typename S::z_t *__restrict__ Cx, int64_t cvlen,
                  typename S::z_t *__restrict__ parked) {
    using Z = typename S::z_t; using P = typename bin_panel<Z>::type;
    extern __shared__ __align__(8) unsigned char bin_panel_lds[];
    P *panel = reinterpret_cast<P *>(bin_panel_lds);
    const int add_code = sr.add_code();
    auto fold = [&](P *p, Z v) { atomic_monoid<P>(add_code, p, (P)v); };
    constexpr int NT = BIN_REDUCE_THREADS;
    const int w = blockIdx.x, tid = threadIdx.x;
    if (w >= chunk_base[nb]) return;                          // uniform
    int lo = 0, hi = nb - 1;                                  // the bin of chunk w: largest b with chunk_base[b] <= w
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (chunk_base[mid] <= w) lo = mid; else hi = mid - 1; }
    const int b = lo, first = chunk_base[b], nch = chunk_base[b + 1] - first;
    const int64_t s = table[(int64_t)b * nwg] + (int64_t)(w - first) * BIN_CHUNK;
    const int64_t bend = table[(int64_t)(b + 1) * nwg];
    const int64_t e = (s + BIN_CHUNK < bend) ? s + BIN_CHUNK : bend;
    const int width = 1 << shift;
    const P id = (P)sr.identity();
    for (int c = tid; c < width; c += NT) panel[c] = id;
    __syncthreads();
    // pairs in groups of eight at aligned positions (one 16-byte load of indices, wide loads of values),
    // two groups in flight per thread; the ragged ends of the chunk one pair at a time
    const int64_t s8 = (s + 7) & ~(int64_t)7, e8 = e & ~(int64_t)7;
    if (s8 < e8) {
        for (int64_t q = s + tid; q < s8; q += NT) fold(&panel[pair_idx[q]], pair_val[q]);
        for (int64_t q = e8 + tid; q < e; q += NT) fold(&panel[pair_idx[q]], pair_val[q]);
        struct alignas(16) Idx8 { uint16_t c[8]; };
        struct alignas(16) Val8 { Z v[8]; };
        const Idx8 *gi = reinterpret_cast<const Idx8 *>(pair_idx);
        const Val8 *gv = reinterpret_cast<const Val8 *>(pair_val);
        const int64_t g0 = s8 >> 3, g1 = e8 >> 3;
        int64_t g = g0 + tid;
        for (; g + NT < g1; g += 2 * NT) {
            const Idx8 ia = gi[g], ib = gi[g + NT];
            const Val8 va = gv[g], vb = gv[g + NT];
#pragma unroll
            for (int u = 0; u < 8; u++) fold(&panel[ia.c[u]], va.v[u]);
#pragma unroll
            for (int u = 0; u < 8; u++) fold(&panel[ib.c[u]], vb.v[u]);
        }
        for (; g < g1; g += NT) {
            const Idx8 ia = gi[g];
            const Val8 va = gv[g];
#pragma unroll
            for (int u = 0; u < 8; u++) fold(&panel[ia.c[u]], va.v[u]);
        }
    } else {
        for (int64_t q = s + tid; q < e; q += NT) fold(&panel[pair_idx[q]], pair_val[q]);
    }
    __syncthreads();
    if (nch == 1) {
        const int64_t base = (int64_t)b << shift;
        for (int c = tid; c < width; c += NT) {
            const P v = panel[c];
            if (base + c < cvlen && !bits_equal<P>(v, id)) Cx[base + c] = sr.add(Cx[base + c], (Z)v);
        }
    } else {
        Z *out = parked + (int64_t)w * width;
        for (int c = tid; c < width; c += NT) out[c] = (Z)panel[c];
    }
}

template <class S>
__global__ void __launch_bounds__(256)
bin_combine_kernel(S sr, int shift, int64_t cvlen, const int *__restrict__ chunk_base,
                   const typename S::z_t *__restrict__ parked, typename S::z_t *__restrict__ Cx) {
    using Z = typename S::z_t;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t width = (int64_t)1 << shift;
    for (; i < cvlen; i += stride) {
        const int b = (int)(i >> shift);
        const int first = chunk_base[b], nch = chunk_base[b + 1] - first;
        if (nch <= 1) continue;
        const Z *src = parked + (int64_t)first * width + (i & (width - 1));
        Z acc = src[0];
        int ch = 1;
        for (; ch + 8 <= nch; ch += 8) {                      // eight loads in flight
            Z t[8];
#pragma unroll
            for (int u = 0; u < 8; u++) t[u] = src[(int64_t)(ch + u) * width];
#pragma unroll
            for (int u = 0; u < 8; u++) acc = sr.add(acc, t[u]);
        }
        for (; ch < nch; ch++) acc = sr.add(acc, src[(int64_t)ch * width]);
        Cx[i] = sr.add(Cx[i], acc);
    }
}

// C(:,j) += A * B(:,j) for every j, C full of the monoid's type, B full; false when the shape does not fit
template <class S>
static gba_info saxpy4_binned(S sr, const gba_matrix *A, const DenseView &Bv, int64_t cvlen, int64_t cvdim,
                              typename S::z_t *cx, hipStream_t st, bool *done, gba_report *rep) {
    using Z = typename S::z_t;
    *done = false;
    constexpr int shift = bin_shift<Z>::v, ITEMS = bin_items<S>::v, BIN_TILE = BIN_THREADS * ITEMS;
    const int64_t nnz = A->nvals, nb64 = ceil_div(cvlen, (int64_t)1 << shift);
    if (Bv.b || nnz < ((int64_t)1 << 20) || nnz >= ((int64_t)1 << 31) || cvlen >= ((int64_t)1 << 31) ||
        nb64 > BIN_MAX || cvdim > 8) return GBA_SUCCESS;
    const int nb = (int)nb64;
    const int64_t nwg = ceil_div(nnz, BIN_TILE), ncell = (int64_t)nb * nwg;
    const int64_t maxchunks = ceil_div(nnz, BIN_CHUNK) + nb;
    DevBuf table_buf, tile_rows_buf, chunk_buf, pidx, pval, parked;
    GBA_OK(pidx.alloc((size_t)nnz * 2));
    GBA_OK(pval.alloc((size_t)nnz * sizeof(Z)));
    GBA_OK(parked.alloc((size_t)maxchunks * sizeof(Z) << shift));
    // the plan (bin offsets of every tile, first vector of every tile, chunks per bin) depends on the pattern of A
    // only: from the second product with the same A on it is kept with the pattern and not computed again
    int64_t *table = nullptr, *tile_rows = nullptr; int *chunk_base = nullptr;
    const int32_t *narrow = nullptr;
    PatternAux *aux = pattern_aux(A);
    bool planned = false;
    if (aux && aux->sightings >= 2) {
        if (aux->bin_table && (aux->bin_shift != shift || aux->bin_tile != BIN_TILE || aux->bin_cvlen != cvlen)) {
            gba_free(aux->bin_table); gba_free(aux->bin_tile_rows); gba_free(aux->bin_chunks);
            aux->bin_table = aux->bin_tile_rows = nullptr; aux->bin_chunks = nullptr;
        }
        planned = aux->bin_table != nullptr;
        if (!planned) {
            aux->bin_table = reinterpret_cast<int64_t *>(gba_malloc((size_t)(ncell + 1) * 8));
            aux->bin_tile_rows = reinterpret_cast<int64_t *>(gba_malloc((size_t)nwg * 8));
            aux->bin_chunks = reinterpret_cast<int *>(gba_malloc((size_t)(nb + 1) * 4));
            if (!aux->bin_table || !aux->bin_tile_rows || !aux->bin_chunks) {
                gba_free(aux->bin_table); gba_free(aux->bin_tile_rows); gba_free(aux->bin_chunks);
                aux->bin_table = aux->bin_tile_rows = nullptr; aux->bin_chunks = nullptr;
            } else {
                aux->bin_shift = shift; aux->bin_tile = BIN_TILE; aux->bin_cvlen = cvlen;
            }
        }
        table = aux->bin_table; tile_rows = aux->bin_tile_rows; chunk_base = aux->bin_chunks;
        if (!aux->i32 && A->vlen < (int64_t)0x7fffffff && A->nvals <= PATTERN_I32_MAX_ENTRIES) {
            aux->i32 = reinterpret_cast<int32_t *>(gba_malloc((size_t)A->nvals * 4));
            if (aux->i32) {
                int64_t nbk = ceil_div(A->nvals, 1024); if (nbk > 16384) nbk = 16384;
                hipLaunchKernelGGL(narrow_indices_kernel, dim3((unsigned)nbk), dim3(256), 0, st, A->p, A->nvec, A->i, aux->i32);
            }
        }
        narrow = aux->i32;
    }
    if (!table) {
        GBA_OK(table_buf.alloc((size_t)(ncell + 1) * 8));
        GBA_OK(tile_rows_buf.alloc((size_t)nwg * 8));
        GBA_OK(chunk_buf.alloc((size_t)(nb + 1) * 4));
        table = table_buf.as<int64_t>(); tile_rows = tile_rows_buf.as<int64_t>(); chunk_base = chunk_buf.as<int>();
    }
    if (!planned) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(bin_count_kernel<ITEMS>), dim3((unsigned)ceil_div(nwg, BIN_COUNT_TILES)), dim3(BIN_THREADS),
                           (size_t)nb * 4 * BIN_COUNT_TILES, st, A->nvec, A->p, A->i, shift, nb, nwg, table);
        GBA_OK(dev_cumsum_i64(table, ncell, st));
        int64_t rb = ceil_div(A->nvec, 256); if (rb > 8192) rb = 8192; if (rb < 1) rb = 1;
        hipLaunchKernelGGL(dot_tile_rows_kernel, dim3((unsigned)rb), dim3(256), 0, st, A->nvec, A->p, (int64_t)BIN_TILE, tile_rows);
        hipLaunchKernelGGL(bin_chunks_kernel, dim3(1), dim3(256), 0, st, nb, nwg, table, chunk_base);
    }
    // as many bin_scatter workgroups as stay resident together, each walking tiles (at most BIN_WALK of them)
    int per_cu = 1;
    GBA_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, bin_scatter_kernel<S, ITEMS>, BIN_THREADS, (size_t)nb * 8));
    int64_t sgrid = (int64_t)ctx().num_cu * (per_cu > 0 ? per_cu : 1);
    sgrid = (sgrid + 7) & ~(int64_t)7;                        // a multiple of the eight XCDs
    const int64_t need = 8 * ceil_div(ceil_div(nwg, 8), BIN_WALK);    // no workgroup walks more than BIN_WALK tiles
    if (sgrid < need) sgrid = need;
    for (int64_t j = 0; j < cvdim; j++) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(bin_scatter_kernel<S, ITEMS>), dim3((unsigned)sgrid), dim3(BIN_THREADS), (size_t)nb * 8, st,
                           sr, A->nvec, A->p, A->i, A->x, A->iso, Bv, j, tile_rows, shift, nb, nwg,
                           table, pidx.as<uint16_t>(), pval.as<Z>(), narrow);
        GBA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bin_reduce_kernel<S>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(typename bin_panel<Z>::type) << shift)));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(bin_reduce_kernel<S>), dim3((unsigned)maxchunks), dim3(BIN_REDUCE_THREADS), sizeof(typename bin_panel<Z>::type) << shift, st,
                           sr, shift, nb, nwg, table, chunk_base, pidx.as<uint16_t>(), pval.as<Z>(),
                           cx + j * cvlen, cvlen, parked.as<Z>());
        hipLaunchKernelGGL(HIP_KERNEL_NAME(bin_combine_kernel<S>), dim3(grid_for(cvlen)), dim3(256), 0, st, sr, shift, cvlen,
                           chunk_base, parked.as<Z>(), cx + j * cvlen);
    }
    GBA_HIP(hipGetLastError());
    GBA_HIP(hipStreamSynchronize(st));                        // the buffers above go back to the pool on return
#ifdef GBA_PANEL_TICKS
    if (getenv("GBA_PANEL_PROF")) {
        unsigned long long t[16], z[16] = {0};
        GBA_HIP(hipMemcpyFromSymbol(t, HIP_SYMBOL(bin_ticks), sizeof(t)));
        GBA_HIP(hipMemcpyToSymbol(HIP_SYMBOL(bin_ticks), z, sizeof(z)));
        fprintf(stderr, "bin_scatter ticks/tile: loads %.0f fill %.0f longfill %.0f rank %.0f scan %.0f sort %.0f write %.0f (tiles %ld)\n",
                (double)t[0] / nwg, (double)t[1] / nwg, (double)t[2] / nwg, (double)t[3] / nwg, (double)t[4] / nwg,
                (double)t[5] / nwg, (double)t[6] / nwg, (long)nwg);
    }
#endif
    if (rep) rep->index_bytes = narrow ? 4 : 8;               // bytes per index the scatter fetched
    *done = true;
    return GBA_SUCCESS;
}

// ---- saxpy4 / saxbit over tiles of entries -------------------------------------------------------
// The kernel above gives every vector of A to a small group of lanes, so one hub vector (10^5 entries in an
// RMAT graph) keeps its group busy long after the rest of the grid has drained: the BFS step with a bitmap
// frontier spent 11 ms that way.  Here every workgroup takes DOT_TILE consecutive ENTRIES of A, first notes
// for each entry which vector owns it (a thread per vector, whole waves for long segments; -1 where B(k,j)
// is absent), then handles the entries one per lane.  Used with every semiring, positional ones included.
template <class S, int MODE>
__global__ void __launch_bounds__(DOT_TILE_THREADS)
saxpy_tile_kernel(S sr, int64_t nvecA, const int64_t *__restrict__ Ap, const int64_t *__restrict__ Ai,
                  const void *__restrict__ Ax_, bool A_iso, DenseView B, MaskViewD M,
                  typename S::acc_t *__restrict__ Cx, int8_t *__restrict__ Cb, int64_t cvlen,
                  const int64_t *__restrict__ tile_row) {
    using X = typename S::x_t; using Z = typename S::z_t;
    constexpr int NLONG = DOT_TILE / DOT_SEG_SERIAL + 2;
    __shared__ int owner[DOT_TILE];
    __shared__ int longlo[NLONG], longhi[NLONG], longv[NLONG];
    __shared__ int nlongq;
    const X *Ax = reinterpret_cast<const X *>(Ax_);
    const X *Bx = reinterpret_cast<const X *>(B.x);
    const int tid = threadIdx.x;
    const int64_t j = blockIdx.y;
    const int64_t pN = Ap[nvecA];
    const int64_t start = Ap[0] + (int64_t)blockIdx.x * DOT_TILE;
    if (start >= pN) return;                                  // uniform
    const int64_t end = (start + DOT_TILE < pN) ? start + DOT_TILE : pN;
    const int64_t r0 = tile_row[blockIdx.x];
    const int64_t boff = j * B.vlen, coff = j * cvlen;
    if (tid == 0) nlongq = 0;
    __syncthreads();
    for (int64_t r = r0 + tid; r < nvecA; r += DOT_TILE_THREADS) {
        const int64_t a0 = Ap[r], a1 = Ap[r + 1];
        if (a0 >= end) break;
        const int lo = (int)((a0 > start ? a0 : start) - start), hi = (int)((a1 < end ? a1 : end) - start);
        if (hi <= lo) continue;
        const int v = (B.b && !B.b[boff + r]) ? -1 : (int)(r - r0);
        if (hi - lo > DOT_SEG_SERIAL) {
            const int q = atomicAdd(&nlongq, 1);
            longlo[q] = lo; longhi[q] = hi; longv[q] = v;
            continue;
        }
        for (int idx = lo; idx < hi; idx++) owner[idx] = v;
    }
    __syncthreads();
    {
        const int nq = nlongq, lane = tid & 63;
        for (int q = tid >> 6; q < nq; q += DOT_TILE_THREADS / 64) {
            const int hi = longhi[q], v = longv[q];
            for (int idx = longlo[q] + lane; idx < hi; idx += 64) owner[idx] = v;
        }
    }
    __syncthreads();
    const bool apat = sr.a_is_pattern(), bpat = sr.b_is_pattern();
#pragma unroll
    for (int it = 0; it < DOT_TILE_ITEMS; it++) {
        const int idx = it * DOT_TILE_THREADS + tid;
        const int64_t p = start + idx;
        if (p >= end) continue;
        const int v = owner[idx];
        if (v < 0) continue;
        const int64_t k = r0 + v, i = Ai[p];
        if (MODE == 1 && !maskd_allows(M, i, j)) continue;
        if (MODE == 1 && sr.add_is_any() && Cb[coff + i]) continue;       // ANY keeps what an earlier product left
        const Z t = sr.mult(load_x<X>(Ax, p, A_iso, apat), load_x<X>(Bx, boff + k, B.iso, bpat), i, k, j);
        sr.atomic(&Cx[coff + i], t);
        if (MODE == 1) Cb[coff + i] = 1;
    }
}

// widen / narrow between the stored type and the accumulator type (8/16-bit and bool)
template <class Z, class ACC>
__global__ void widen_kernel(const Z *__restrict__ in, ACC *__restrict__ out, int64_t n) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; idx < n; idx += stride) out[idx] = (ACC)in[idx];
}
template <class Z, class ACC>
__global__ void narrow_kernel(const ACC *__restrict__ in, Z *__restrict__ out, int64_t n) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; idx < n; idx += stride) out[idx] = (Z)in[idx];
}
template <class ACC>
__global__ void fill_kernel(ACC *p, int64_t n, ACC v) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; idx < n; idx += stride) p[idx] = v;
}

template <int MODE>
static gba_info saxpy_sparse_dense(gba_matrix *C, const gba_matrix *M, bool Mask_comp, bool Mask_struct,
                                   const gba_matrix *A, const gba_matrix *B, const Canon &c, gba_report *rep) {
    GBA_OK(require_init());
    if (!is_sparse(A) || !is_dense_fmt(B))
        GBA_FAIL(GBA_NOT_IMPLEMENTED, "%s on device: A must be sparse and B bitmap/full (got %d, %d)",
                 MODE == 0 ? "saxpy4" : "saxbit", A->sparsity, B->sparsity);
    if (A->vdim != B->vlen) GBA_FAIL(GBA_DIMENSION_MISMATCH, "saxpy: A->vdim %ld != B->vlen %ld", (long)A->vdim, (long)B->vlen);
    if (A->nzombies) GBA_FAIL(GBA_INVALID_OBJECT, "saxpy: A has zombies");
    hipStream_t st = ctx().stream;
    const int64_t cvlen = A->vlen, cvdim = B->vdim;
    if (cvdim > 65535) GBA_FAIL(GBA_NO_VALUE, "saxpy: more than 65535 dense vectors (declined: the caller takes the general path)");
    const int64_t cn = cvlen * cvdim;
    MaskViewD mv; GBA_OK(make_maskd(M, Mask_comp, Mask_struct, &mv));
    if (MODE == 0) {
        if (!is_full(C) || C->type != c.ztype || C->vlen != cvlen || C->vdim != cvdim || C->iso)
            GBA_FAIL(GBA_INVALID_VALUE, "saxpy4: C must be a full non-iso matrix of the monoid type, %ld-by-%ld", (long)cvlen, (long)cvdim);
    }
    const int lpr = pick_group(A);
    DenseView Bv{B->x, is_bitmap(B) ? B->b : nullptr, B->vlen, B->iso};
    DevBuf Cb, Cx;
    const size_t zsize = type_size(c.ztype);
    if (MODE == 1) {
        GBA_OK(Cb.alloc((size_t)(cn > 0 ? cn : 1)));
        GBA_OK(Cx.alloc((size_t)(cn > 0 ? cn : 1) * zsize));
        GBA_HIP(hipMemsetAsync(Cb.p, 0, (size_t)(cn > 0 ? cn : 1), st));
    }
    KernelTimer timer(rep != nullptr);
    if (cn > 0) {
        GBA_OK(with_semiring(c, [&](auto sr) -> gba_info {
            using S = decltype(sr); using Z = typename S::z_t; using ACC = typename S::acc_t;
            Z *cx = reinterpret_cast<Z *>(MODE == 0 ? C->x : Cx.p);
            ACC *acc = reinterpret_cast<ACC *>(cx);
            DevBuf wide;
            constexpr bool WIDEN = !std::is_same<Z, ACC>::value;
            if (WIDEN) {
                GBA_OK(wide.alloc((size_t)cn * sizeof(ACC)));
                acc = wide.as<ACC>();
                if (MODE == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(widen_kernel<Z, ACC>), dim3(grid_for(cn)), dim3(256), 0, st, cx, acc, cn);
            }
            if (MODE == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(fill_kernel<ACC>), dim3(grid_for(cn)), dim3(256), 0, st, acc, cn, (ACC)sr.identity());
            if constexpr (MODE == 0 && dot_tile_ok<S>::v) {
                bool done = false;
                GBA_OK(saxpy4_binned<S>(sr, A, Bv, cvlen, cvdim, cx, st, &done, rep));
                if (done) return GBA_SUCCESS;
            }
            DevBuf tile_rows;
            if (A->nvals >= 4 * DOT_TILE) {                    // entry-balanced tiles; small products keep a group per vector
                const int64_t ntiles = ceil_div(A->nvals, DOT_TILE);
                GBA_OK(tile_rows.alloc((size_t)ntiles * 8));
                int64_t rb = ceil_div(A->nvec, 256); if (rb > 8192) rb = 8192; if (rb < 1) rb = 1;
                hipLaunchKernelGGL(dot_tile_rows_kernel, dim3((unsigned)rb), dim3(256), 0, st, A->nvec, A->p, (int64_t)DOT_TILE,
                                   tile_rows.as<int64_t>());
                hipLaunchKernelGGL(HIP_KERNEL_NAME(saxpy_tile_kernel<S, MODE>), dim3((unsigned)ntiles, (unsigned)cvdim),
                                   dim3(DOT_TILE_THREADS), 0, st, sr, A->nvec, A->p, A->i, A->x, A->iso, Bv, mv, acc,
                                   Cb.as<int8_t>(), cvlen, tile_rows.as<int64_t>());
                GBA_HIP(hipGetLastError());
                GBA_HIP(hipStreamSynchronize(st));             // tile_rows goes back to the pool on return
            } else {
            dim3 grid((unsigned)ceil_div(A->nvec * lpr, 256), (unsigned)cvdim);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(saxpy_sd_kernel<S, MODE>), grid, dim3(256), 0, st, sr, A->nvec, A->p, A->i,
                               A->x, A->iso, Bv, mv, lpr, acc, Cb.as<int8_t>(), cvlen);
            }
            if (WIDEN) {
                hipLaunchKernelGGL(HIP_KERNEL_NAME(narrow_kernel<Z, ACC>), dim3(grid_for(cn)), dim3(256), 0, st, acc, cx, cn);
                GBA_HIP(hipStreamSynchronize(st));
            }
            GBA_HIP(hipGetLastError());
            return GBA_SUCCESS;
        }));
    }
    if (rep) rep->kernel_ms = timer.stop_ms();
    if (MODE == 1) {
        int64_t nvals = 0;
        GBA_OK(count_ones(Cb.as<int8_t>(), cn, &nvals));
        C->vlen = cvlen; C->vdim = cvdim; C->nvec = cvdim; C->nvals = nvals;
        C->p = nullptr; C->h = nullptr; C->i = nullptr;
        C->b = Cb.as<int8_t>(); Cb.release();
        C->x = Cx.release();
        C->nzombies = 0; C->type = c.ztype; C->sparsity = GBA_BITMAP; C->mem = GBA_MEM_DEVICE;
        C->iso = false; C->jumbled = false; C->is_csc = B->is_csc;
    } else {
        GBA_HIP(hipStreamSynchronize(st));
    }
    return GBA_SUCCESS;
}

gba_info saxpy4(gba_matrix *C, const gba_matrix *A, const gba_matrix *B, const Canon &c, gba_report *rep) {
    if (c.add == GBA_ANY) return GBA_NO_VALUE;      // Source/GB_AxB_saxpy4.c: the ANY monoid is declined
    return saxpy_sparse_dense<0>(C, nullptr, false, false, A, B, c, rep);
}

// ---------------------------------------------------------- dense x dense --
// saxbit with both operands dense: one thread per C(i,j), k runs over the inner dimension
template <class S>
__global__ void __launch_bounds__(256)
saxbit_dd_kernel(S sr, DenseView A, DenseView B, MaskViewD M, int64_t cvlen, int64_t cvdim, int64_t inner,
                 typename S::z_t *__restrict__ Cx, int8_t *__restrict__ Cb) {
    using X = typename S::x_t; using Z = typename S::z_t;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= cvlen * cvdim) return;
    const int64_t j = idx / cvlen, i = idx - j * cvlen;
    if (!maskd_allows(M, i, j)) { Cb[idx] = 0; return; }
    const X *Ax = reinterpret_cast<const X *>(A.x);
    const X *Bx = reinterpret_cast<const X *>(B.x);
    const bool apat = sr.a_is_pattern(), bpat = sr.b_is_pattern();
    Z acc = sr.identity(); int found = 0;
    for (int64_t k = 0; k < inner; k++) {
        const int64_t pA = k * A.vlen + i, pB = j * B.vlen + k;
        if (A.b && !A.b[pA]) continue;
        if (B.b && !B.b[pB]) continue;
        const Z t = sr.mult(load_x<X>(Ax, pA, A.iso, apat), load_x<X>(Bx, pB, B.iso, bpat), i, k, j);
        acc = found ? sr.add(acc, t) : t; found = 1;
    }
    Cb[idx] = (int8_t)found;
    if (found) Cx[idx] = acc;
}

gba_info saxbit(gba_matrix *C, const gba_matrix *M, bool Mask_comp, bool Mask_struct,
                const gba_matrix *A, const gba_matrix *B, const Canon &c, gba_report *rep) {
    GBA_OK(require_init());
    if (M && (M->vlen != A->vlen || M->vdim != B->vdim)) GBA_FAIL(GBA_DIMENSION_MISMATCH, "saxbit: mask dimensions");
    if (is_sparse(A) && is_dense_fmt(B)) return saxpy_sparse_dense<1>(C, M, Mask_comp, Mask_struct, A, B, c, rep);
    if (is_dense_fmt(A) && is_dense_fmt(B)) {
        if (A->vdim != B->vlen) GBA_FAIL(GBA_DIMENSION_MISMATCH, "saxbit: inner dimensions");
        hipStream_t st = ctx().stream;
        const int64_t cvlen = A->vlen, cvdim = B->vdim, cn = cvlen * cvdim;
        MaskViewD mv; GBA_OK(make_maskd(M, Mask_comp, Mask_struct, &mv));
        DevBuf Cb, Cx;
        GBA_OK(Cb.alloc((size_t)(cn > 0 ? cn : 1)));
        GBA_OK(Cx.alloc((size_t)(cn > 0 ? cn : 1) * type_size(c.ztype)));
        KernelTimer timer(rep != nullptr);
        if (cn > 0) {
            DenseView Av{A->x, is_bitmap(A) ? A->b : nullptr, A->vlen, A->iso};
            DenseView Bv{B->x, is_bitmap(B) ? B->b : nullptr, B->vlen, B->iso};
            GBA_OK(with_semiring(c, [&](auto sr) -> gba_info {
                using S = decltype(sr); using Z = typename S::z_t;
                hipLaunchKernelGGL(HIP_KERNEL_NAME(saxbit_dd_kernel<S>), dim3((unsigned)ceil_div(cn, 256)), dim3(256), 0, st,
                                   sr, Av, Bv, mv, cvlen, cvdim, A->vdim, Cx.as<Z>(), Cb.as<int8_t>());
                GBA_HIP(hipGetLastError());
                return GBA_SUCCESS;
            }));
        }
        if (rep) rep->kernel_ms = timer.stop_ms();
        int64_t nvals = 0;
        GBA_OK(count_ones(Cb.as<int8_t>(), cn, &nvals));
        C->vlen = cvlen; C->vdim = cvdim; C->nvec = cvdim; C->nvals = nvals;
        C->p = nullptr; C->h = nullptr; C->i = nullptr;
        C->b = Cb.as<int8_t>(); Cb.release();
        C->x = Cx.release();
        C->nzombies = 0; C->type = c.ztype; C->sparsity = GBA_BITMAP; C->mem = GBA_MEM_DEVICE;
        C->iso = false; C->jumbled = false; C->is_csc = B->is_csc;
        return GBA_SUCCESS;
    }
    GBA_FAIL(GBA_NOT_IMPLEMENTED, "saxbit on device: A sparse or dense, B dense (got %d, %d)", A->sparsity, B->sparsity);
}

// ------------------------------------------------------------------ saxpy5 --
// C(:,j) += sum_k A(:,k) (x) B(k,j): A dense (column k contiguous), B sparse.
// One thread per row i of a column tile; the column of B is walked by all
// threads together, so A is read coalesced and nothing is atomic
// (Source/Template/GB_AxB_saxpy5_unrolled.c:58-140 keeps 16-row register panels).
template <class S>
__global__ void __launch_bounds__(256)
saxpy5_kernel(S sr, DenseView A, const int64_t *__restrict__ Bp, const int64_t *__restrict__ Bi,
              const void *__restrict__ Bx_, bool B_iso, int64_t cvlen, typename S::z_t *__restrict__ Cx) {
    using X = typename S::x_t; using Z = typename S::z_t;
    const X *Ax = reinterpret_cast<const X *>(A.x);
    const X *Bx = reinterpret_cast<const X *>(Bx_);
    const int64_t j = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cvlen) return;
    const bool apat = sr.a_is_pattern(), bpat = sr.b_is_pattern();
    const int64_t pB1 = Bp[j + 1];
    Z acc = Cx[j * cvlen + i];
    for (int64_t pB = Bp[j]; pB < pB1; pB++) {
        const int64_t k = Bi[pB];
        const int64_t pA = k * A.vlen + i;
        if (A.b && !A.b[pA]) continue;
        const Z t = sr.mult(load_x<X>(Ax, pA, A.iso, apat), load_x<X>(Bx, pB, B_iso, bpat), i, k, j);
        acc = sr.add(acc, t);
    }
    Cx[j * cvlen + i] = acc;
}

gba_info saxpy5(gba_matrix *C, const gba_matrix *A, const gba_matrix *B, const Canon &c, gba_report *rep) {
    GBA_OK(require_init());
    if (c.add == GBA_ANY) return GBA_NO_VALUE;
    if (!is_dense_fmt(A) || !is_sparse(B)) GBA_FAIL(GBA_NOT_IMPLEMENTED, "saxpy5 on device: A bitmap/full, B sparse");
    if (A->vdim != B->vlen) GBA_FAIL(GBA_DIMENSION_MISMATCH, "saxpy5: inner dimensions");
    const int64_t cvlen = A->vlen, cvdim = B->vdim;
    if (!is_full(C) || C->type != c.ztype || C->vlen != cvlen || C->vdim != cvdim || C->iso)
        GBA_FAIL(GBA_INVALID_VALUE, "saxpy5: C must be a full non-iso matrix of the monoid type");
    if (cvdim > 65535) GBA_FAIL(GBA_NO_VALUE, "saxpy5: more than 65535 vectors in B (declined: the caller takes the general path)");
    hipStream_t st = ctx().stream;
    KernelTimer timer(rep != nullptr);
    if (cvlen > 0 && cvdim > 0) {
        DenseView Av{A->x, is_bitmap(A) ? A->b : nullptr, A->vlen, A->iso};
        GBA_OK(with_semiring(c, [&](auto sr) -> gba_info {
            using S = decltype(sr); using Z = typename S::z_t;
            dim3 grid((unsigned)ceil_div(cvlen, 256), (unsigned)cvdim);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(saxpy5_kernel<S>), grid, dim3(256), 0, st, sr, Av, B->p, B->i, B->x, B->iso,
                               cvlen, reinterpret_cast<Z *>(C->x));
            GBA_HIP(hipGetLastError());
            return GBA_SUCCESS;
        }));
    }
    if (rep) rep->kernel_ms = timer.stop_ms();
    GBA_HIP(hipStreamSynchronize(st));
    return GBA_SUCCESS;
}

}  // namespace gba
