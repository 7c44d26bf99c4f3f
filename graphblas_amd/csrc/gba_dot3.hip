// C<M> = A'*B with M sparse and not complemented: the device counterpart of
// GB_AxB_dot3 (Source/GB_AxB_dot3.c:37) and its kernel bodies
// Source/Template/GB_AxB_dot3_template.c:16-168 and GB_AxB_dot_cij.c:543-679.
//
// C takes the pattern of M.  Every mask entry M(i,j) is one dot product
// A(:,i)'B(:,j); an entry whose dot product has no term becomes a zombie
// (index replaced by GBA_FLIP(i), Source/Template/GB_AxB_dot3_template.c:159-164).
// The CPU estimates the work of every entry (1 + min(ainz,bjnz), phase 1,
// Source/Template/GB_AxB_dot3_phase1_template.c:97-125) and slices it over
// threads.  On the GPU the same estimate routes an entry either to one thread
// (short lists: the CPU's three-way merge / gallop) or to a whole wavefront
// (long lists: 64 lanes stride the shorter list and binary-search the longer
// one, results combined with shuffles).
#include "gba_internal.hpp"

namespace gba {

struct OperandView {            // sparse (kind 1) or bitmap (2) / full (3)
    int kind; const int64_t *p; const int64_t *i; const void *x; const int8_t *b; int64_t vlen; bool iso;
};
static OperandView make_operand(const gba_matrix *A) {
    OperandView v{};
    v.kind = is_sparse(A) ? 1 : (is_bitmap(A) ? 2 : 3);
    v.p = A->p; v.i = A->i; v.x = A->x; v.b = is_bitmap(A) ? A->b : nullptr; v.vlen = A->vlen; v.iso = A->iso;
    return v;
}

struct Dot3Mask { const int64_t *p; const int64_t *i; const uint8_t *x; int msize; bool structural, iso; };

__device__ inline bool dot3_mask_value(const Dot3Mask &m, int64_t p) {
    if (m.structural) return true;
    const uint8_t *v = m.x + (m.iso ? 0 : p * m.msize);
    bool nz = false;
    for (int q = 0; q < m.msize; q++) nz |= (v[q] != 0);
    return nz;
}

// vector id of every mask entry (the CPU gets it from its task slicing)
__global__ void __launch_bounds__(256)
expand_vectors_kernel(int64_t nvec, const int64_t *__restrict__ Mp, int64_t *__restrict__ Jof) {
    const int lane = threadIdx.x & 63;
    int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (; j < nvec; j += stride) {
        const int64_t p1 = Mp[j + 1];
        for (int64_t p = Mp[j] + lane; p < p1; p += 64) Jof[p] = j;
    }
}

template <class S>
__device__ inline typename S::z_t dot_term(const S &sr, const OperandView &A, const OperandView &B, bool apat, bool bpat,
                                           int64_t i, int64_t k, int64_t j, int64_t pA, int64_t pB) {
    using X = typename S::x_t;
    const X a = load_x<X>(reinterpret_cast<const X *>(A.x), pA, A.iso, apat);
    const X b = load_x<X>(reinterpret_cast<const X *>(B.x), pB, B.iso, bpat);
    return sr.mult(a, b, i, k, j);
}

// first position in [lo,hi] whose index is >= target (GB_TRIM_BINARY_SEARCH)
__device__ inline int64_t lower_bound(const int64_t *__restrict__ X, int64_t lo, int64_t hi, int64_t target) {
    while (lo <= hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (X[mid] < target) lo = mid + 1; else hi = mid - 1;
    }
    return lo;
}

// one thread, whole dot product; follows Source/Template/GB_AxB_dot_cij.c
template <class S>
__device__ inline void dot_serial(const S &sr, const OperandView &A, const OperandView &B, int64_t i, int64_t j,
                                  typename S::z_t &cij, int &found) {
    using Z = typename S::z_t;
    const bool apat = sr.a_is_pattern(), bpat = sr.b_is_pattern();
    const bool any = sr.add_is_any();
    found = 0; cij = sr.identity();
    auto term = [&](int64_t k, int64_t pA, int64_t pB) {
        const Z t = dot_term(sr, A, B, apat, bpat, i, k, j, pA, pB);
        cij = found ? sr.add(cij, t) : t; found = 1;
    };
    if (A.kind == 1 && B.kind == 1) {
        int64_t pA = A.p[i], pA_end = A.p[i + 1], pB = B.p[j], pB_end = B.p[j + 1];
        const int64_t ainz = pA_end - pA, bjnz = pB_end - pB;
        if (ainz == 0 || bjnz == 0) return;
        if (A.i[pA_end - 1] < B.i[pB] || B.i[pB_end - 1] < A.i[pA]) return;     // no overlap (dot_cij.c:555)
        if (ainz > 8 * bjnz) {                                                   // gallop in A (dot_cij.c:565-603)
            while (pA < pA_end && pB < pB_end) {
                const int64_t ia = A.i[pA], ib = B.i[pB];
                if (ia < ib) pA = lower_bound(A.i, pA + 1, pA_end - 1, ib);
                else if (ib < ia) pB++;
                else { term(ia, pA, pB); if (any) return; pA++; pB++; }
            }
        } else if (bjnz > 8 * ainz) {                                            // gallop in B (dot_cij.c:605-643)
            while (pA < pA_end && pB < pB_end) {
                const int64_t ia = A.i[pA], ib = B.i[pB];
                if (ia < ib) pA++;
                else if (ib < ia) pB = lower_bound(B.i, pB + 1, pB_end - 1, ia);
                else { term(ia, pA, pB); if (any) return; pA++; pB++; }
            }
        } else {                                                                 // two-pointer merge (dot_cij.c:645-678)
            while (pA < pA_end && pB < pB_end) {
                const int64_t ia = A.i[pA], ib = B.i[pB];
                if (ia < ib) pA++;
                else if (ib < ia) pB++;
                else { term(ia, pA, pB); if (any) return; pA++; pB++; }
            }
        }
    } else if (A.kind == 1) {                                                    // A sparse, B dense
        const int64_t pA_end = A.p[i + 1], boff = j * B.vlen;
        for (int64_t pA = A.p[i]; pA < pA_end; pA++) {
            const int64_t k = A.i[pA];
            if (B.b && !B.b[boff + k]) continue;
            term(k, pA, boff + k); if (any) return;
        }
    } else if (B.kind == 1) {                                                    // A dense, B sparse
        const int64_t pB_end = B.p[j + 1], aoff = i * A.vlen;
        for (int64_t pB = B.p[j]; pB < pB_end; pB++) {
            const int64_t k = B.i[pB];
            if (A.b && !A.b[aoff + k]) continue;
            term(k, aoff + k, pB); if (any) return;
        }
    } else {                                                                     // both dense
        const int64_t aoff = i * A.vlen, boff = j * B.vlen;
        for (int64_t k = 0; k < A.vlen; k++) {
            if (A.b && !A.b[aoff + k]) continue;
            if (B.b && !B.b[boff + k]) continue;
            term(k, aoff + k, boff + k); if (any) return;
        }
    }
}

__device__ inline int64_t list_len(const OperandView &V, int64_t v) {
    return (V.kind == 1) ? (V.p[v + 1] - V.p[v]) : V.vlen;
}

template <class S>
__global__ void __launch_bounds__(256)
dot3_thread_kernel(S sr, int64_t mnz, Dot3Mask M, const int64_t *__restrict__ Jof, OperandView A, OperandView B,
                   int64_t long_work, int64_t *__restrict__ Ci, typename S::z_t *__restrict__ Cx,
                   unsigned long long *__restrict__ nzombies, unsigned int *__restrict__ nlong,
                   int64_t *__restrict__ long_list) {
    using Z = typename S::z_t;
    const int64_t pC = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int zombie = 0;
    if (pC < mnz) {
        const int64_t i = M.i[pC], j = Jof[pC];
        if (!dot3_mask_value(M, pC)) {
            Ci[pC] = GBA_FLIP(i); zombie = 1;
        } else {
            const int64_t ainz = list_len(A, i), bjnz = list_len(B, j);
            const int64_t work = (A.kind == 1 && B.kind == 1) ? (ainz < bjnz ? ainz : bjnz) * 4 + 1
                                                              : (ainz < bjnz ? ainz : bjnz);
            if (work > long_work) {
                const unsigned int q = atomicAdd(nlong, 1u);
                long_list[q] = pC;
            } else {
                Z cij; int found;
                dot_serial(sr, A, B, i, j, cij, found);
                if (found) { Ci[pC] = i; Cx[pC] = cij; }
                else { Ci[pC] = GBA_FLIP(i); zombie = 1; }
            }
        }
    }
    unsigned long long z = (unsigned long long)zombie;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) z += __shfl_down(z, d, 64);
    if ((threadIdx.x & 63) == 0 && z) atomicAdd(nzombies, z);
}

// one wavefront per long dot product
template <class S>
__global__ void __launch_bounds__(256)
dot3_wave_kernel(S sr, Dot3Mask M, const int64_t *__restrict__ Jof, OperandView A, OperandView B,
                 int64_t *__restrict__ Ci, typename S::z_t *__restrict__ Cx,
                 unsigned long long *__restrict__ nzombies, const unsigned int *__restrict__ nlong,
                 const int64_t *__restrict__ long_list) {
    using Z = typename S::z_t;
    const int lane = threadIdx.x & 63;
    const unsigned int n = *nlong;
    const bool apat = sr.a_is_pattern(), bpat = sr.b_is_pattern();
    unsigned int q = (unsigned int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const unsigned int stride = (unsigned int)(((int64_t)gridDim.x * blockDim.x) >> 6);
    for (; q < n; q += stride) {
        const int64_t pC = long_list[q];
        const int64_t i = M.i[pC], j = Jof[pC];
        Z cij = sr.identity(); int found = 0;
        auto term = [&](int64_t k, int64_t pA, int64_t pB) {
            const Z t = dot_term(sr, A, B, apat, bpat, i, k, j, pA, pB);
            cij = found ? sr.add(cij, t) : t; found = 1;
        };
        if (A.kind == 1 && B.kind == 1) {
            const int64_t pA0 = A.p[i], pA1 = A.p[i + 1], pB0 = B.p[j], pB1 = B.p[j + 1];
            if (pA1 - pA0 <= pB1 - pB0) {           // stride A(:,i), search B(:,j)
                for (int64_t pA = pA0 + lane; pA < pA1; pA += 64) {
                    const int64_t k = A.i[pA];
                    const int64_t pB = lower_bound(B.i, pB0, pB1 - 1, k);
                    if (pB < pB1 && B.i[pB] == k) term(k, pA, pB);
                }
            } else {
                for (int64_t pB = pB0 + lane; pB < pB1; pB += 64) {
                    const int64_t k = B.i[pB];
                    const int64_t pA = lower_bound(A.i, pA0, pA1 - 1, k);
                    if (pA < pA1 && A.i[pA] == k) term(k, pA, pB);
                }
            }
        } else if (A.kind == 1) {
            const int64_t pA1 = A.p[i + 1], boff = j * B.vlen;
            for (int64_t pA = A.p[i] + lane; pA < pA1; pA += 64) {
                const int64_t k = A.i[pA];
                if (B.b && !B.b[boff + k]) continue;
                term(k, pA, boff + k);
            }
        } else if (B.kind == 1) {
            const int64_t pB1 = B.p[j + 1], aoff = i * A.vlen;
            for (int64_t pB = B.p[j] + lane; pB < pB1; pB += 64) {
                const int64_t k = B.i[pB];
                if (A.b && !A.b[aoff + k]) continue;
                term(k, aoff + k, pB);
            }
        } else {
            const int64_t aoff = i * A.vlen, boff = j * B.vlen;
            for (int64_t k = lane; k < A.vlen; k += 64) {
                if (A.b && !A.b[aoff + k]) continue;
                if (B.b && !B.b[boff + k]) continue;
                term(k, aoff + k, boff + k);
            }
        }
        group_reduce(sr, cij, found, 64);
        if (lane == 0) {
            if (found) { Ci[pC] = i; Cx[pC] = cij; }
            else { Ci[pC] = GBA_FLIP(i); atomicAdd(nzombies, 1ull); }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Both operands sparse: one workgroup per chunk of 1024 consecutive mask entries.
//
// All entries of one mask vector j share the list B(:,j).  Where a chunk holds enough entries of a vector whose B(:,j)
// is long, the workgroup stages B(:,j) in an LDS hash set (D3_SEG keys at a time) and every dot product of that run
// becomes a stream over A(:,i) with one LDS probe per index: coalesced reads of A's list instead of one binary search
// through global memory per index.  A list A(:,i) far longer than the staged keys is still searched (the keys of
// B(:,j) are looked up in it); a list of at most D3_SMALL entries is walked by a single thread.  Entries of every other
// vector take the serial dot product of dot_serial, or go to the long list of dot3_wave_kernel, as the work estimate
// of Source/Template/GB_AxB_dot3_phase1_template.c:97-125 says.
constexpr int D3_THREADS = 1024;            // threads of a workgroup = mask entries of a chunk
#ifndef GBA_D3_SLOTS
#define GBA_D3_SLOTS 16384
#endif
#ifndef GBA_D3_WGS
#define GBA_D3_WGS 1
#endif
constexpr int D3_SLOTS = GBA_D3_SLOTS;       // slots of the LDS hash set
constexpr int D3_SEG = D3_SLOTS / 2;        // keys of B(:,j) staged at a time
#ifndef GBA_D3_HEAVY
#define GBA_D3_HEAVY 128
#endif
#ifndef GBA_D3_SMALL
#define GBA_D3_SMALL 8
#endif
#ifndef GBA_D3_UNROLL
#define GBA_D3_UNROLL 8
#endif
constexpr int D3_HEAVY = GBA_D3_HEAVY;       // shortest B(:,j) worth staging
constexpr int D3_SMALL = GBA_D3_SMALL;       // longest A(:,i) one thread walks alone
constexpr uint32_t D3_EMPTY = 0xffffffffu;
constexpr int D3_UNROLL = GBA_D3_UNROLL;     // indices a lane loads per step of a streamed list

// The hash set of the team kernel: buckets of four 32-bit keys, filled front to back, so one 16-byte LDS read
// settles most lookups and a bucket whose last slot is empty ends the search.  Returns the slot of k, or -1.
__device__ inline int d3_probe(const uint32_t *tab, uint32_t b, uint32_t bmask, uint32_t k) {
    for (;;) {
        const uint4 w = *reinterpret_cast<const uint4 *>(tab + 4 * b);
        if (w.x == k) return (int)(4 * b);
        if (w.y == k) return (int)(4 * b + 1);
        if (w.z == k) return (int)(4 * b + 2);
        if (w.w == k) return (int)(4 * b + 3);
        if (w.w == D3_EMPTY) return -1;
        b = (b + 1) & bmask;
    }
}

static size_t dot3_team_lds_bytes(size_t zsize) {
    return (size_t)D3_THREADS * 12 + (size_t)D3_SLOTS * 4 + (size_t)D3_THREADS * zsize + (size_t)D3_SLOTS * 2 + (size_t)D3_THREADS * 5;
}

template <class S>
__global__ void __launch_bounds__(D3_THREADS, GBA_D3_WGS)
dot3_team_kernel(S sr, int64_t mnz, Dot3Mask M, const int64_t *__restrict__ Jof, OperandView A, OperandView B,
                 int64_t long_work, int64_t *__restrict__ Ci, typename S::z_t *__restrict__ Cx,
                 unsigned long long *__restrict__ nzombies, unsigned int *__restrict__ nlong,
                 int64_t *__restrict__ long_list, unsigned int *__restrict__ next_chunk,
                 unsigned int *__restrict__ nshort, int64_t *__restrict__ short_list) {
    using Z = typename S::z_t;
    extern __shared__ __align__(16) unsigned char d3_lds[];
    int64_t *list_at = reinterpret_cast<int64_t *>(d3_lds);                 // A(:,i) of every entry of the chunk: start
    uint32_t *list_n = reinterpret_cast<uint32_t *>(list_at + D3_THREADS);  // and length (below 2^32, as its indices are)
    uint32_t *tab = list_n + D3_THREADS;
    Z *acc = reinterpret_cast<Z *>(tab + D3_SLOTS);
    uint16_t *tpos = reinterpret_cast<uint16_t *>(reinterpret_cast<unsigned char *>(acc) + (size_t)D3_THREADS * sizeof(Z));
    uint16_t *queue = tpos + D3_SLOTS;
    uint16_t *runs = queue + D3_THREADS;
    uint8_t *state = reinterpret_cast<uint8_t *>(runs + D3_THREADS);     // 0: not a team entry, 1: no term yet, 2: acc holds a sum
    __shared__ int s_nruns, s_qn, s_qhead;
    __shared__ unsigned int s_chunk;
    const int tid = threadIdx.x, lane = tid & 63;
    const bool apat = sr.a_is_pattern(), bpat = sr.b_is_pattern();
    const int64_t nchunks = (mnz + D3_THREADS - 1) / D3_THREADS;
    for (;;) {
        __syncthreads();
        if (tid == 0) { s_chunk = atomicAdd(next_chunk, 1u); s_nruns = 0; }
        __syncthreads();
        const int64_t c = s_chunk;
        if (c >= nchunks) break;
        const int64_t c0 = c * D3_THREADS, c1 = (c0 + D3_THREADS < mnz) ? c0 + D3_THREADS : mnz;
        const int64_t e = c0 + tid;
        int64_t i = 0, pA0 = 0, la = 0;
        int zombie = 0, mine = 0, lone = 0;
        if (e < c1) {
            i = M.i[e];
            const int64_t j = Jof[e];
            const int64_t m0 = M.p[j], m1 = M.p[j + 1];
            const int64_t r0 = m0 > c0 ? m0 : c0, r1 = m1 < c1 ? m1 : c1;
            const int64_t lb = B.p[j + 1] - B.p[j];
            const bool team = lb >= D3_HEAVY && ((r1 - r0) >= 16 || (r1 - r0) * 64 >= lb);
            if (team && e == r0) runs[atomicAdd(&s_nruns, 1)] = (uint16_t)tid;
            pA0 = A.p[i]; la = A.p[i + 1] - pA0;
            if (!dot3_mask_value(M, e) || la == 0 || lb == 0) {
                Ci[e] = GBA_FLIP(i); zombie = 1;
            } else if (team) {
                mine = 1;
            } else if ((la < lb ? la : lb) * 4 + 1 > long_work) {
                long_list[atomicAdd(nlong, 1u)] = e;
            } else {
                lone = 1;
            }
        }
        {   // entries outside every run go to dot3_group_kernel: one slot claim per wavefront
            const unsigned long long votes = __ballot(lone);
            if (votes) {
                unsigned int at = 0;
                if (lane == __ffsll((long long)votes) - 1) at = atomicAdd(nshort, (unsigned int)__popcll(votes));
                at = __shfl(at, __ffsll((long long)votes) - 1, 64);
                if (lone) short_list[at + __popcll(votes & ((1ull << lane) - 1ull))] = e;
            }
        }
        state[tid] = (uint8_t)mine;
        list_at[tid] = pA0; list_n[tid] = (uint32_t)la;
        __syncthreads();
        const int nruns = s_nruns;
        for (int q = 0; q < nruns; q++) {
            const int t0 = runs[q];
            const int64_t j = Jof[c0 + t0];
            const int64_t m1 = M.p[j + 1];
            const int t1 = (int)((m1 < c1 ? m1 : c1) - c0);                  // the run is entries [t0, t1) of the chunk
            const int64_t pB0 = B.p[j], lb = B.p[j + 1] - pB0;
            const bool multi = lb > D3_SEG;
            for (int64_t seg0 = 0; seg0 < lb; seg0 += D3_SEG) {
                const int segn = (int)((lb - seg0 < D3_SEG) ? lb - seg0 : D3_SEG);
                int bits = 6;                                                 // slots: four per key where they fit, never under two
                while ((1 << bits) < 4 * segn && (1 << bits) < D3_SLOTS) bits++;
                const int bshift = 32 - (bits - 2);
                const uint32_t bmask = (1u << (bits - 2)) - 1u;
                for (int s = tid; s < (1 << bits); s += D3_THREADS) tab[s] = D3_EMPTY;
                if (tid == 0) { s_qn = 0; s_qhead = 0; }
                __syncthreads();
                const int64_t pBs = pB0 + seg0;
                for (int t = tid; t < segn; t += D3_THREADS) {
                    const uint32_t k = (uint32_t)B.i[pBs + t];
                    uint32_t s = 4 * ((k * 2654435761u) >> bshift);
                    while (atomicCAS(&tab[s], D3_EMPTY, k) != D3_EMPTY) s = (s + 1) & (4 * bmask + 3);
                    tpos[s] = (uint16_t)t;
                }
                int64_t kfirst = 0, klast = INT64_MAX;
                if (multi) { kfirst = B.i[pBs]; klast = B.i[pBs + segn - 1]; }
                __syncthreads();
                // a thread walks its own short list
                if (mine && tid >= t0 && tid < t1) {
                    if (la <= D3_SMALL) {
                        Z cij = sr.identity(); int found = 0;
                        for (int64_t pA = pA0; pA < pA0 + la; pA++) {
                            const int64_t k = A.i[pA];
                            if (k < kfirst || k > klast) continue;
                            const int s = d3_probe(tab, ((uint32_t)k * 2654435761u) >> bshift, bmask, (uint32_t)k);
                            if (s >= 0) {
                                const Z t = dot_term(sr, A, B, apat, bpat, i, k, j, pA, pBs + tpos[s]);
                                cij = found ? sr.add(cij, t) : t; found = 1;
                            }
                        }
                        if (found) {
                            if (state[tid] == 2) acc[tid] = sr.add(acc[tid], cij);
                            else { acc[tid] = cij; state[tid] = 2; }
                        }
                    } else {
                        queue[atomicAdd(&s_qn, 1)] = (uint16_t)tid;
                    }
                }
                __syncthreads();
                // a wavefront per longer list
                const int qn = s_qn;
                for (;;) {
                    int t = 0;
                    if (lane == 0) t = atomicAdd(&s_qhead, 1);
                    t = __builtin_amdgcn_readfirstlane(t);
                    if (t >= qn) break;
                    const int ent = queue[t];
                    const int64_t ii = M.i[c0 + ent];
                    int64_t lo = list_at[ent], hi = lo + list_n[ent];
                    if (multi) {
                        const int64_t last = hi - 1;
                        lo = lower_bound(A.i, lo, last, kfirst);
                        hi = lower_bound(A.i, lo, last, klast + 1);
                    }
                    const int64_t n = hi - lo;
                    Z cij = sr.identity(); int found = 0;
                    if (n > 0) {
                        const int lg = 64 - __clzll((unsigned long long)n);
                        if (n <= (int64_t)segn * lg * 2) {                     // stream A(:,i), probe the staged keys
                            for (int64_t base = lo + lane; base < hi; base += 64 * D3_UNROLL) {
                                int64_t kk[D3_UNROLL];
#pragma unroll
                                for (int u = 0; u < D3_UNROLL; u++) {          // all loads of the step in flight together
                                    const int64_t pA = base + 64 * u;
                                    kk[u] = (pA < hi) ? A.i[pA] : -1;
                                }
                                uint4 w[D3_UNROLL];
#pragma unroll
                                for (int u = 0; u < D3_UNROLL; u++)            // and then the first bucket of every lookup
                                    w[u] = *reinterpret_cast<const uint4 *>(tab + 4 * (((uint32_t)kk[u] * 2654435761u) >> bshift));
#pragma unroll
                                for (int u = 0; u < D3_UNROLL; u++) {
                                    const int64_t k = kk[u];
                                    if (k < 0) continue;
                                    const uint32_t k32 = (uint32_t)k, b = (k32 * 2654435761u) >> bshift;
                                    int s = (w[u].x == k32) ? 0 : (w[u].y == k32) ? 1 : (w[u].z == k32) ? 2 : (w[u].w == k32) ? 3 : -1;
                                    if (s >= 0) s += (int)(4 * b);
                                    else if (w[u].w != D3_EMPTY) s = d3_probe(tab, (b + 1) & bmask, bmask, k32);
                                    if (s >= 0) {
                                        const Z t2 = dot_term(sr, A, B, apat, bpat, ii, k, j, base + 64 * u, pBs + tpos[s]);
                                        cij = found ? sr.add(cij, t2) : t2; found = 1;
                                    }
                                }
                            }
                        } else {                                               // few keys, long list: look the keys up in A(:,i)
                            for (int t2 = lane; t2 < segn; t2 += 64) {
                                const int64_t pB = pBs + t2, k = B.i[pB];
                                const int64_t pA = lower_bound(A.i, lo, hi - 1, k);
                                if (pA < hi && A.i[pA] == k) {
                                    const Z t3 = dot_term(sr, A, B, apat, bpat, ii, k, j, pA, pB);
                                    cij = found ? sr.add(cij, t3) : t3; found = 1;
                                }
                            }
                        }
                    }
                    group_reduce(sr, cij, found, 64);
                    if (lane == 0 && found) {
                        if (state[ent] == 2) acc[ent] = sr.add(acc[ent], cij);
                        else { acc[ent] = cij; state[ent] = 2; }
                    }
                }
                __syncthreads();
            }
        }
        if (mine) {
            if (state[tid] == 2) { Ci[e] = i; Cx[e] = acc[tid]; }
            else { Ci[e] = GBA_FLIP(i); zombie = 1; }
        }
        unsigned long long z = (unsigned long long)zombie;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) z += __shfl_down(z, d, 64);
        if (lane == 0 && z) atomicAdd(nzombies, z);
    }
}

// Entries the team kernel leaves alone (their mask vector has a short B(:,j), or too few entries in the chunk): the
// shorter list has at most 63 entries.  Eight lanes share one dot product: they stride the shorter list and look each
// index up in the longer one, the gallop of Source/Template/GB_AxB_dot_cij.c:565-643 spread over the lanes.
template <class S>
__global__ void __launch_bounds__(256)
dot3_group_kernel(S sr, Dot3Mask M, const int64_t *__restrict__ Jof, OperandView A, OperandView B,
                  int64_t *__restrict__ Ci, typename S::z_t *__restrict__ Cx, unsigned long long *__restrict__ nzombies,
                  const unsigned int *__restrict__ nshort, const int64_t *__restrict__ short_list) {
    using Z = typename S::z_t;
    const bool apat = sr.a_is_pattern(), bpat = sr.b_is_pattern();
    const int sub = threadIdx.x & 7;
    const int64_t n = *nshort;
    int zombie = 0;
    for (int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; q < n; q += ((int64_t)gridDim.x * blockDim.x) >> 3) {
        const int64_t e = short_list[q];
        const int64_t i = M.i[e], j = Jof[e];
        const int64_t pA0 = A.p[i], pA1 = A.p[i + 1], pB0 = B.p[j], pB1 = B.p[j + 1];
        Z cij = sr.identity(); int found = 0;
        if (pA1 - pA0 <= pB1 - pB0) {
            for (int64_t pA = pA0 + sub; pA < pA1; pA += 8) {
                const int64_t k = A.i[pA];
                const int64_t pB = lower_bound(B.i, pB0, pB1 - 1, k);
                if (pB < pB1 && B.i[pB] == k) {
                    const Z t = dot_term(sr, A, B, apat, bpat, i, k, j, pA, pB);
                    cij = found ? sr.add(cij, t) : t; found = 1;
                }
            }
        } else {
            for (int64_t pB = pB0 + sub; pB < pB1; pB += 8) {
                const int64_t k = B.i[pB];
                const int64_t pA = lower_bound(A.i, pA0, pA1 - 1, k);
                if (pA < pA1 && A.i[pA] == k) {
                    const Z t = dot_term(sr, A, B, apat, bpat, i, k, j, pA, pB);
                    cij = found ? sr.add(cij, t) : t; found = 1;
                }
            }
        }
        group_reduce(sr, cij, found, 8);
        if (sub == 0) {
            if (found) { Ci[e] = i; Cx[e] = cij; }
            else { Ci[e] = GBA_FLIP(i); zombie++; }
        }
    }
    unsigned long long z = (unsigned long long)zombie;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) z += __shfl_down(z, d, 64);
    if ((threadIdx.x & 63) == 0 && z) atomicAdd(nzombies, z);
}

gba_info dot3(gba_matrix *C, const gba_matrix *M, bool Mask_struct, const gba_matrix *A,
              const gba_matrix *B, const Canon &c, gba_report *rep) {
    GBA_OK(require_init());
    if (!M || !is_sparse(M)) GBA_FAIL(GBA_NOT_IMPLEMENTED, "dot3 on device needs a sparse mask");
    if (is_hyper(A) || is_hyper(B)) GBA_FAIL(GBA_NOT_IMPLEMENTED, "dot3 on device: hypersparse operand");
    if ((is_sparse(A) && A->jumbled) || (is_sparse(B) && B->jumbled))
        GBA_FAIL(GBA_NOT_IMPLEMENTED, "dot3: operands must not be jumbled (wait on them first, Source/GB_AxB_dot.c:78-87)");
    if (A->nzombies || B->nzombies || M->nzombies) GBA_FAIL(GBA_INVALID_OBJECT, "dot3: zombies in an input");
    if (A->vlen != B->vlen) GBA_FAIL(GBA_DIMENSION_MISMATCH, "dot3: A->vlen %ld != B->vlen %ld", (long)A->vlen, (long)B->vlen);
    if (M->vlen != A->vdim || M->vdim != B->vdim) GBA_FAIL(GBA_DIMENSION_MISMATCH, "dot3: mask is %ld-by-%ld, need %ld-by-%ld",
                                                           (long)M->vlen, (long)M->vdim, (long)A->vdim, (long)B->vdim);
    hipStream_t st = ctx().stream;
    const int64_t mnz = M->nvals, nvec = M->nvec;
    const size_t zsize = type_size(c.ztype);
    DevBuf Cp, Ci, Cx, Jof, cnt, long_list, short_list;
    GBA_OK(Cp.alloc((size_t)(nvec + 1) * 8));
    GBA_OK(Ci.alloc((size_t)(mnz > 0 ? mnz : 1) * 8));
    GBA_OK(Cx.alloc((size_t)(mnz > 0 ? mnz : 1) * zsize));
    GBA_OK(Jof.alloc((size_t)(mnz > 0 ? mnz : 1) * 8));
    GBA_OK(long_list.alloc((size_t)(mnz > 0 ? mnz : 1) * 8));
    GBA_OK(cnt.alloc(32));
    GBA_HIP(hipMemsetAsync(cnt.p, 0, 32, st));
    GBA_HIP(hipMemcpyAsync(Cp.p, M->p, (size_t)(nvec + 1) * 8, hipMemcpyDeviceToDevice, st));
    Dot3Mask mv{M->p, M->i, reinterpret_cast<const uint8_t *>(M->x), (int)type_size(M->type), Mask_struct, M->iso};
    if (!Mask_struct && (!M->x || mv.msize == 0)) GBA_FAIL(GBA_INVALID_OBJECT, "dot3: valued mask without values");
    unsigned long long *nzomb = cnt.as<unsigned long long>();
    unsigned int *nlong = reinterpret_cast<unsigned int *>(cnt.as<unsigned long long>() + 1);
    KernelTimer timer(rep != nullptr);
    if (mnz > 0) {
        int64_t eb = ceil_div(nvec * 64, 256); if (eb > 8192) eb = 8192;
        hipLaunchKernelGGL(expand_vectors_kernel, dim3((unsigned)eb), dim3(256), 0, st, nvec, M->p, Jof.as<int64_t>());
        OperandView Av = make_operand(A), Bv = make_operand(B);
        // the hash set of the team kernel keeps 32-bit keys
        const bool team = is_sparse(A) && is_sparse(B) && A->vlen < (int64_t)D3_EMPTY;
        if (team) GBA_OK(short_list.alloc((size_t)mnz * 8));
        GBA_OK(with_semiring(c, [&](auto sr) -> gba_info {
            using S = decltype(sr); using Z = typename S::z_t;
            if (team) {
                const size_t lds = dot3_team_lds_bytes(sizeof(Z));
                GBA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&dot3_team_kernel<S>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                int64_t nchunks = ceil_div(mnz, (int64_t)D3_THREADS);
                if (nchunks > 1024) nchunks = 1024;
                hipLaunchKernelGGL(HIP_KERNEL_NAME(dot3_team_kernel<S>), dim3((unsigned)nchunks), dim3(D3_THREADS), lds, st,
                                   sr, mnz, mv, Jof.as<int64_t>(), Av, Bv, (int64_t)256, Ci.as<int64_t>(), Cx.as<Z>(),
                                   nzomb, nlong, long_list.as<int64_t>(), nlong + 1, nlong + 2, short_list.as<int64_t>());
                hipLaunchKernelGGL(HIP_KERNEL_NAME(dot3_group_kernel<S>), dim3(8192), dim3(256), 0, st,
                                   sr, mv, Jof.as<int64_t>(), Av, Bv, Ci.as<int64_t>(), Cx.as<Z>(), nzomb, nlong + 2,
                                   short_list.as<int64_t>());
            } else {
                hipLaunchKernelGGL(HIP_KERNEL_NAME(dot3_thread_kernel<S>), dim3((unsigned)ceil_div(mnz, 256)), dim3(256), 0, st,
                                   sr, mnz, mv, Jof.as<int64_t>(), Av, Bv, (int64_t)256, Ci.as<int64_t>(), Cx.as<Z>(),
                                   nzomb, nlong, long_list.as<int64_t>());
            }
            hipLaunchKernelGGL(HIP_KERNEL_NAME(dot3_wave_kernel<S>), dim3(2048), dim3(256), 0, st,
                               sr, mv, Jof.as<int64_t>(), Av, Bv, Ci.as<int64_t>(), Cx.as<Z>(), nzomb, nlong,
                               long_list.as<int64_t>());
            GBA_HIP(hipGetLastError());
            return GBA_SUCCESS;
        }));
    }
    if (rep) rep->kernel_ms = timer.stop_ms();
    int64_t nz = 0;
    GBA_OK(dev_read_i64(cnt.as<int64_t>(), &nz));
    C->vlen = M->vlen; C->vdim = M->vdim; C->nvec = nvec; C->nvals = mnz;
    C->p = reinterpret_cast<int64_t *>(Cp.release());
    C->h = nullptr;
    C->i = reinterpret_cast<int64_t *>(Ci.release());
    C->x = Cx.release();
    C->b = nullptr;
    C->nzombies = nz;
    C->type = c.ztype; C->sparsity = GBA_SPARSE; C->mem = GBA_MEM_DEVICE;
    C->iso = false; C->jumbled = M->jumbled; C->is_csc = M->is_csc;
    return GBA_SUCCESS;
}

}  // namespace gba
