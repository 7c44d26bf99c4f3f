// Library context, device memory, prefix sums and the semiring canonicaliser.
#include "gba_internal.hpp"
#include <mutex>
#include <map>
#include <unordered_map>
#include <iterator>
#include <cstdlib>
#include <cstring>

namespace gba {

// ------------------------------------------------------------------ errors --
static thread_local char g_err[1024] = "";
std::string last_error() { return std::string(g_err); }
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static Context g_ctx;
Context &ctx() { return g_ctx; }

gba_info require_init() {
    if (!g_ctx.inited) GBA_FAIL(GBA_UNINITIALIZED_OBJECT, "gba_init has not been called");
    return GBA_SUCCESS;
}

// ------------------------------------------------------------- prefix sum --
// Three-step exclusive scan over n+1 slots (slot n holds the total):
// per-tile sums, scan of the tile sums (recursively), per-tile scan + offset.
static constexpr int SCAN_THREADS = 256;
static constexpr int SCAN_ITEMS = 8;
static constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ inline int64_t wave_incl_scan(int64_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int64_t u = __shfl_up(v, d, 64);
        if (lane >= d) v += u;
    }
    return v;
}

__global__ void __launch_bounds__(SCAN_THREADS)
scan_tile_sums(const int64_t *__restrict__ a, int64_t n, int64_t *__restrict__ tile_sum) {
    __shared__ int64_t wsum[SCAN_THREADS / 64];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
    int64_t s = 0;
#pragma unroll
    for (int it = 0; it < SCAN_ITEMS; it++) {
        int64_t idx = base + it * SCAN_THREADS + threadIdx.x;
        if (idx < n) s += a[idx];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t t = 0;
        for (int w = 0; w < SCAN_THREADS / 64; w++) t += wsum[w];
        tile_sum[blockIdx.x] = t;
    }
}

// each thread owns SCAN_ITEMS consecutive slots of its tile
__global__ void __launch_bounds__(SCAN_THREADS)
scan_tiles(int64_t *__restrict__ a, int64_t n, const int64_t *__restrict__ tile_off) {
    __shared__ int64_t wsum[SCAN_THREADS / 64];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int64_t v[SCAN_ITEMS];
    int64_t s = 0;
#pragma unroll
    for (int it = 0; it < SCAN_ITEMS; it++) {
        int64_t idx = base + it;
        v[it] = (idx < n) ? a[idx] : 0;
        s += v[it];
    }
    int64_t incl = wave_incl_scan(s);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) wsum[w] = incl;
    __syncthreads();
    int64_t off = tile_off[blockIdx.x];
    for (int q = 0; q < w; q++) off += wsum[q];
    int64_t run = off + incl - s;
#pragma unroll
    for (int it = 0; it < SCAN_ITEMS; it++) {
        int64_t idx = base + it;
        if (idx <= n) a[idx] = run;
        run += v[it];
    }
}

gba_info dev_cumsum_i64(int64_t *count, int64_t n, hipStream_t st) {
    if (n < 0) GBA_FAIL(GBA_INVALID_VALUE, "cumsum: n < 0");
    const int64_t slots = n + 1;
    const int64_t ntiles = ceil_div(slots, SCAN_TILE);
    DevBuf sums;
    GBA_OK(sums.alloc((size_t)(ntiles + 1) * sizeof(int64_t)));
    hipLaunchKernelGGL(scan_tile_sums, dim3((unsigned)ntiles), dim3(SCAN_THREADS), 0, st, count, n, sums.as<int64_t>());
    if (ntiles > 1) {
        GBA_OK(dev_cumsum_i64(sums.as<int64_t>(), ntiles, st));
    } else {
        GBA_HIP(hipMemsetAsync(sums.p, 0, sizeof(int64_t), st));
    }
    hipLaunchKernelGGL(scan_tiles, dim3((unsigned)ntiles), dim3(SCAN_THREADS), 0, st, count, n, sums.as<int64_t>());
    GBA_HIP(hipGetLastError());
    GBA_HIP(hipStreamSynchronize(st));   // sums is freed on return
    return GBA_SUCCESS;
}

// ---- prefix sum without any allocation -------------------------------------------------------------
// The tile sums of the three-step scan live in a device array that is part of the library's code object:
// each GPU the library runs on holds its own copy at its own address until the process ends.  Host code
// never passes the array's name to a kernel -- only the address hipGetSymbolAddress returned for the
// device in use (taken in gba_init, with that device current).
static constexpr int64_t NOALLOC_L1 = 65536;                     // level-1 tile sums (+ their total)
static constexpr int64_t NOALLOC_L2 = 64, NOALLOC_L3 = 64;       // sums of the sums, twice
static constexpr int64_t NOALLOC_SLOTS = NOALLOC_L1 + NOALLOC_L2 + NOALLOC_L3;
static_assert((NOALLOC_L1 - 1 + SCAN_TILE) / SCAN_TILE + 1 <= NOALLOC_L2, "level 2 holds the tile sums of level 1");
static __device__ __align__(256) int64_t scratch[NOALLOC_SLOTS];  // 525 312 bytes per GPU
static constexpr int NOALLOC_MAX_DEVICES = 64;
static int64_t *g_scratch_ptr[NOALLOC_MAX_DEVICES] = {nullptr};   // device address of scratch[], per device

static gba_info noalloc_bind_scratch(int device) {
    if (device < 0 || device >= NOALLOC_MAX_DEVICES) return GBA_SUCCESS;       // such a device gets no allocation-free scan
    if (g_scratch_ptr[device]) return GBA_SUCCESS;
    void *ptr = nullptr;
    GBA_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));                 // the current device's copy
    g_scratch_ptr[device] = reinterpret_cast<int64_t *>(ptr);
    return GBA_SUCCESS;
}

// a[0..n] -> exclusive prefix sums in place; the tile sums of level k go to the k-th part of sc[]
static gba_info noalloc_scan(int64_t *count, int64_t n, int64_t *sc, hipStream_t st) {
    int64_t *lev[3] = {sc, sc + NOALLOC_L1, sc + NOALLOC_L1 + NOALLOC_L2};
    const int64_t cap[3] = {NOALLOC_L1, NOALLOC_L2, NOALLOC_L3};
    int64_t *a[3]; int64_t nn[3];
    a[0] = count; nn[0] = n;
    int depth = 0;
    while (true) {
        const int64_t ntiles = ceil_div(nn[depth] + 1, SCAN_TILE);
        if (ntiles + 1 > cap[depth]) GBA_FAIL(GBA_PANIC, "cumsum: scratch level %d too small for %ld tiles", depth, (long)ntiles);
        hipLaunchKernelGGL(scan_tile_sums, dim3((unsigned)ntiles), dim3(SCAN_THREADS), 0, st, a[depth], nn[depth], lev[depth]);
        if (ntiles == 1) {
            GBA_HIP(hipMemsetAsync(lev[depth], 0, sizeof(int64_t), st));    // the only tile starts at zero; the array keeps the last call's contents
            break;
        }
        if (depth == 2) GBA_FAIL(GBA_PANIC, "cumsum: scratch levels exhausted");
        a[depth + 1] = lev[depth]; nn[depth + 1] = ntiles;
        depth++;
    }
    for (int d = depth; d >= 0; d--) {
        const int64_t ntiles = ceil_div(nn[d] + 1, SCAN_TILE);
        hipLaunchKernelGGL(scan_tiles, dim3((unsigned)ntiles), dim3(SCAN_THREADS), 0, st, a[d], nn[d], lev[d]);
    }
    GBA_HIP(hipGetLastError());
    return GBA_SUCCESS;
}

__global__ void fill_i64_kernel(int64_t *p, int64_t n, int64_t v) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; idx < n; idx += stride) p[idx] = v;
}
gba_info dev_fill_i64(int64_t *p, int64_t n, int64_t v, hipStream_t st) {
    if (n <= 0) return GBA_SUCCESS;
    int64_t blocks = ceil_div(n, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fill_i64_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, n, v);
    GBA_HIP(hipGetLastError());
    return GBA_SUCCESS;
}
gba_info dev_fill_bytes(void *p, size_t n, int byte, hipStream_t st) {
    if (n == 0) return GBA_SUCCESS;
    GBA_HIP(hipMemsetAsync(p, byte, n, st));
    return GBA_SUCCESS;
}

__global__ void max_diff_kernel(const int64_t *__restrict__ p, int64_t n, unsigned long long *out) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t m = 0;
    for (; idx < n; idx += stride) { int64_t d = p[idx + 1] - p[idx]; m = d > m ? d : m; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { int64_t o = __shfl_down(m, d, 64); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(out, (unsigned long long)m);
}
gba_info dev_max_diff_i64(const int64_t *p, int64_t n, int64_t *out_host) {
    *out_host = 0;
    if (n <= 0) return GBA_SUCCESS;
    DevBuf o; GBA_OK(o.alloc(8));
    hipStream_t st = ctx().stream;
    GBA_HIP(hipMemsetAsync(o.p, 0, 8, st));
    int64_t blocks = ceil_div(n, 256); if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(max_diff_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, n, o.as<unsigned long long>());
    GBA_HIP(hipMemcpyAsync(out_host, o.p, 8, hipMemcpyDeviceToHost, st));
    GBA_HIP(hipStreamSynchronize(st));
    return GBA_SUCCESS;
}
gba_info dev_read_i64(const int64_t *p, int64_t *out_host) {
    hipStream_t st = ctx().stream;
    GBA_HIP(hipMemcpyAsync(out_host, p, 8, hipMemcpyDeviceToHost, st));
    GBA_HIP(hipStreamSynchronize(st));
    return GBA_SUCCESS;
}

__global__ void count_nonempty_kernel(const int64_t *__restrict__ p, int64_t n, unsigned long long *out) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long c = 0;
    for (; idx < n; idx += stride) c += (p[idx + 1] > p[idx]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}
// GB_nvec_nonempty (Source/GB_nvec_nonempty.c)
gba_info dev_count_nonempty(const int64_t *p, int64_t nvec, int64_t *out_host) {
    *out_host = 0;
    if (nvec <= 0) return GBA_SUCCESS;
    DevBuf o; GBA_OK(o.alloc(8));
    hipStream_t st = ctx().stream;
    GBA_HIP(hipMemsetAsync(o.p, 0, 8, st));
    int64_t blocks = ceil_div(nvec, 256); if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(count_nonempty_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, nvec, o.as<unsigned long long>());
    GBA_HIP(hipMemcpyAsync(out_host, o.p, 8, hipMemcpyDeviceToHost, st));
    GBA_HIP(hipStreamSynchronize(st));
    return GBA_SUCCESS;
}

// GB_is_diagonal (Source/GB_is_diagonal.c): square, nvals == n, every vector j holds exactly entry j
__global__ void is_diag_kernel(const int64_t *__restrict__ p, const int64_t *__restrict__ i, int64_t n, int *bad) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    int64_t s = p[j];
    if (p[j + 1] - s != 1 || i[s] != j) *bad = 1;
}
gba_info dev_is_diagonal(const gba_matrix *A, bool *out) {
    *out = false;
    if (!is_sparse(A) || A->vlen != A->vdim || A->nvals != A->vdim || A->nzombies != 0) return GBA_SUCCESS;
    const int64_t n = A->vdim;
    if (n == 0) { return GBA_SUCCESS; }
    DevBuf o; GBA_OK(o.alloc(4));
    hipStream_t st = ctx().stream;
    GBA_HIP(hipMemsetAsync(o.p, 0, 4, st));
    hipLaunchKernelGGL(is_diag_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, A->p, A->i, n, o.as<int>());
    int bad = 0;
    GBA_HIP(hipMemcpyAsync(&bad, o.p, 4, hipMemcpyDeviceToHost, st));
    GBA_HIP(hipStreamSynchronize(st));
    *out = (bad == 0);
    return GBA_SUCCESS;
}

// ------------------------------------------------------ semiring canonical --
// GB_flip_binop (Source/GB_flip_binop.c:355-375)
int flip_binop(int op, bool *flipxy) {
    if (!*flipxy) return op;
    *flipxy = false;
    switch (op) {
        case GBA_FIRST: return GBA_SECOND;   case GBA_SECOND: return GBA_FIRST;
        case GBA_MINUS: return GBA_RMINUS;   case GBA_RMINUS: return GBA_MINUS;
        case GBA_DIV: return GBA_RDIV;       case GBA_RDIV: return GBA_DIV;
        case GBA_GT: return GBA_LT;          case GBA_LT: return GBA_GT;
        case GBA_GE: return GBA_LE;          case GBA_LE: return GBA_GE;
        case GBA_ISGT: return GBA_ISLT;      case GBA_ISLT: return GBA_ISGT;
        case GBA_ISGE: return GBA_ISLE;      case GBA_ISLE: return GBA_ISGE;
        case GBA_FIRSTI: return GBA_SECONDJ;   case GBA_SECONDJ: return GBA_FIRSTI;
        case GBA_FIRSTI1: return GBA_SECONDJ1; case GBA_SECONDJ1: return GBA_FIRSTI1;
        case GBA_FIRSTJ: return GBA_SECONDI;   case GBA_SECONDI: return GBA_FIRSTJ;
        case GBA_FIRSTJ1: return GBA_SECONDI1; case GBA_SECONDI1: return GBA_FIRSTJ1;
        // commutative: unchanged
        case GBA_ANY: case GBA_PAIR: case GBA_MIN: case GBA_MAX: case GBA_PLUS: case GBA_TIMES:
        case GBA_ISEQ: case GBA_ISNE: case GBA_EQ: case GBA_NE:
        case GBA_LOR: case GBA_LAND: case GBA_LXOR:
        case GBA_BOR: case GBA_BAND: case GBA_BXOR: case GBA_BXNOR:
            return op;
        default:
            *flipxy = true;   // cannot be flipped: not a built-in semiring any more
            return op;
    }
}

// GB_boolean_rename (Source/GB_boolean_rename.c:45-97)
int boolean_rename(int op) {
    switch (op) {
        case GBA_DIV: return GBA_FIRST;
        case GBA_RDIV: return GBA_SECOND;
        case GBA_MIN: case GBA_TIMES: return GBA_LAND;
        case GBA_MAX: case GBA_PLUS: return GBA_LOR;
        case GBA_MINUS: case GBA_RMINUS: case GBA_NE: case GBA_ISNE: return GBA_LXOR;
        case GBA_ISEQ: return GBA_EQ;
        case GBA_ISGT: return GBA_GT;
        case GBA_ISLT: return GBA_LT;
        case GBA_POW: case GBA_ISGE: return GBA_GE;
        case GBA_ISLE: return GBA_LE;
        default: return op;
    }
}

static bool is_uint_type(int t) { return t == GBA_UINT8 || t == GBA_UINT16 || t == GBA_UINT32 || t == GBA_UINT64; }

// GB_AxB_semiring_builtin (Source/GB_AxB_semiring_builtin.c:20-123) +
// GB_binop_builtin (Source/GB_binop_builtin.c:68-88)
gba_info canonicalize(const gba_semiring *s, bool flipxy, int atype, int btype, Canon *out) {
    if (!s) GBA_FAIL(GBA_NULL_POINTER, "semiring is NULL");
    int mult = s->mult, add = s->add;
    if (mult == GBA_USER_BINOP || add == GBA_USER_BINOP || mult < GBA_FIRST || mult > GBA_SECONDJ1)
        GBA_FAIL(GBA_NOT_IMPLEMENTED, "user-defined or unknown operator in semiring (add %d mult %d)", add, mult);
    mult = flip_binop(mult, &flipxy);
    if (flipxy) GBA_FAIL(GBA_NOT_IMPLEMENTED, "multiplier %d cannot be flipped", mult);
    Canon c;
    if (op_is_positional(mult)) {
        if (mult == GBA_SECONDI) mult = GBA_FIRSTJ;
        if (mult == GBA_SECONDI1) mult = GBA_FIRSTJ1;
        if (s->ztype != GBA_INT32 && s->ztype != GBA_INT64)
            GBA_FAIL(GBA_DOMAIN_MISMATCH, "positional multiplier needs an int32/int64 monoid");
        if (!(add == GBA_MIN || add == GBA_MAX || add == GBA_TIMES || add == GBA_PLUS || add == GBA_ANY))
            GBA_FAIL(GBA_NOT_IMPLEMENTED, "monoid %d with a positional multiplier", add);
        c.kind = KIND_POS; c.add = add; c.mult = mult; c.xtype = GBA_UINT8; c.ztype = s->ztype;
        c.a_pattern = c.b_pattern = true;
        *out = c;
        return GBA_SUCCESS;
    }
    int xtype = s->xtype, ztype = s->ztype;
    if (!is_real_type(ztype)) GBA_FAIL(GBA_NOT_IMPLEMENTED, "ztype %d (complex/user types stay on the CPU)", ztype);
    const bool a_pat = (mult == GBA_SECOND || mult == GBA_PAIR);
    const bool b_pat = (mult == GBA_FIRST || mult == GBA_PAIR);
    if (mult == GBA_PAIR) xtype = ztype;
    if (!is_real_type(xtype)) GBA_FAIL(GBA_NOT_IMPLEMENTED, "xtype %d", xtype);
    if (!a_pat && !b_pat && s->ytype != s->xtype) GBA_FAIL(GBA_NOT_IMPLEMENTED, "multiplier with different x and y types");
    if (a_pat && !b_pat) xtype = s->ytype;
    // boolean renames
    if (xtype == GBA_BOOL) mult = boolean_rename(mult);
    if (ztype == GBA_BOOL) {
        if (add == GBA_MIN || add == GBA_TIMES) add = GBA_LAND;
        else if (add == GBA_MAX || add == GBA_PLUS) add = GBA_LOR;
    }
    const bool mult_is_cmp = (mult >= GBA_EQ && mult <= GBA_LE);
    if (mult_is_cmp && xtype != GBA_BOOL) {
        if (ztype != GBA_BOOL) GBA_FAIL(GBA_DOMAIN_MISMATCH, "comparator multiplier needs a boolean monoid");
        c.kind = KIND_CMP;
    } else {
        if (xtype != ztype) GBA_FAIL(GBA_NOT_IMPLEMENTED, "multiplier output type differs from monoid type (typecast)");
        c.kind = KIND_TT;
    }
    // monoid must be a built-in one for its type
    if (ztype == GBA_BOOL) {
        if (!(add == GBA_LOR || add == GBA_LAND || add == GBA_LXOR || add == GBA_EQ || add == GBA_ANY))
            GBA_FAIL(GBA_NOT_IMPLEMENTED, "boolean monoid %d", add);
    } else if (add == GBA_BOR || add == GBA_BAND || add == GBA_BXOR || add == GBA_BXNOR) {
        if (!is_uint_type(ztype)) GBA_FAIL(GBA_NOT_IMPLEMENTED, "bitwise monoid on type %d", ztype);
    } else if (!(add == GBA_MIN || add == GBA_MAX || add == GBA_PLUS || add == GBA_TIMES || add == GBA_ANY)) {
        GBA_FAIL(GBA_NOT_IMPLEMENTED, "monoid %d on type %d", add, ztype);
    }
    if ((mult == GBA_BOR || mult == GBA_BAND || mult == GBA_BXOR || mult == GBA_BXNOR) && !is_uint_type(xtype))
        GBA_FAIL(GBA_NOT_IMPLEMENTED, "bitwise multiplier on type %d", xtype);
    if (mult == GBA_ANY || mult == GBA_POW) GBA_FAIL(GBA_NOT_IMPLEMENTED, "multiplier %d", mult);
    // ANY_PAIR collapse (Source/GB_AxB_semiring_builtin.c:108-120): the result is 1 everywhere
    if (mult == GBA_PAIR && (add == GBA_EQ || add == GBA_LAND || add == GBA_BAND || add == GBA_LOR ||
                             add == GBA_BOR || add == GBA_MAX || add == GBA_MIN || add == GBA_TIMES))
        add = GBA_ANY;
    // no typecasting of the operands on the device (Source/GB_binop_builtin.c:68-88)
    if (!a_pat && atype != xtype) GBA_FAIL(GBA_NOT_IMPLEMENTED, "A has type %d but the multiplier wants %d (typecast)", atype, xtype);
    if (!b_pat && btype != xtype) GBA_FAIL(GBA_NOT_IMPLEMENTED, "B has type %d but the multiplier wants %d (typecast)", btype, xtype);
    c.add = add; c.mult = mult; c.xtype = xtype; c.ztype = ztype; c.a_pattern = a_pat; c.b_pattern = b_pat;
    *out = c;
    return GBA_SUCCESS;
}

// device address of an array in pinned, mapped host memory (hipHostMalloc / gba_host_malloc)
static gba_info device_alias(const void *host, void **dev) {
    *dev = nullptr;
    if (!host) return GBA_SUCCESS;
    GBA_HIP(hipHostGetDevicePointer(dev, const_cast<void *>(host), 0));
    return GBA_SUCCESS;
}

gba_info stage_to_device(const gba_matrix *in, gba_matrix *out, std::vector<void *> &owned, bool in_place) {
    *out = *in;
    if (in->mem == GBA_MEM_DEVICE) return GBA_SUCCESS;
    if (in->mem == GBA_MEM_HOST_VISIBLE && in_place) {
        // the kernels read the caller's arrays through the device's mapping of them: nothing is copied
        GBA_OK(device_alias(in->p, (void **)&out->p));
        GBA_OK(device_alias(in->h, (void **)&out->h));
        GBA_OK(device_alias(in->i, (void **)&out->i));
        GBA_OK(device_alias(in->x, &out->x));
        GBA_OK(device_alias(in->b, (void **)&out->b));
        out->mem = GBA_MEM_DEVICE;
        return GBA_SUCCESS;
    }
    hipStream_t st = ctx().stream;
    auto up = [&](const void *src, size_t bytes, void **dst) -> gba_info {
        *dst = nullptr;
        if (!src) return GBA_SUCCESS;
        void *d = gba_malloc(bytes ? bytes : 1);
        if (!d) GBA_FAIL(GBA_OUT_OF_MEMORY, "staging %zu bytes", bytes);
        owned.push_back(d);
        GBA_HIP(hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, st));
        *dst = d;
        return GBA_SUCCESS;
    };
    const size_t xs = type_size(in->type);
    const int64_t dense = in->vlen * in->vdim;
    // a sparse view may start in the middle of its arrays (a row slice keeps the absolute offsets in p): only
    // the entries p[0] .. p[nvec) travel, and the device arrays are addressed by the same offsets
    int64_t e0 = 0, held = nnz_held(in);
    if (!is_dense_fmt(in) && in->p) { e0 = in->p[0]; held = in->p[in->nvec] - e0; }
    GBA_OK(up(in->p, (size_t)(in->nvec + 1) * 8, (void **)&out->p));
    GBA_OK(up(in->h, (size_t)in->nvec * 8, (void **)&out->h));
    GBA_OK(up(in->i ? in->i + e0 : nullptr, (size_t)held * 8, (void **)&out->i));
    if (out->i) out->i -= e0;
    GBA_OK(up(in->b, (size_t)dense, (void **)&out->b));
    if (in->iso || is_dense_fmt(in)) {
        GBA_OK(up(in->x, (size_t)(in->iso ? 1 : dense) * xs, &out->x));
    } else {
        GBA_OK(up(in->x ? reinterpret_cast<const uint8_t *>(in->x) + (size_t)e0 * xs : nullptr, (size_t)held * xs, &out->x));
        if (out->x) out->x = reinterpret_cast<uint8_t *>(out->x) - (size_t)e0 * xs;
    }
    GBA_HIP(hipStreamSynchronize(st));
    out->mem = GBA_MEM_DEVICE;
    return GBA_SUCCESS;
}

// ------------------------------------------------------------ device pool --
// hipFree of a multi-gigabyte block unmaps its pages and can stall for most of
// a second; outputs of that size are produced and dropped on every call of a
// graph algorithm's inner loop.  Freed blocks are therefore kept, sorted by
// capacity, and handed out again (best fit) to requests they fit with at most 2x slack
// (the role the RMM pool plays for the reference's CUDA build,
// rmm_wrap/rmm_wrap.h:76-83).  Blocks are reused in stream order: a block may
// be given back while kernels that touch it are still queued on the library
// stream, because whatever uses it next is queued behind them.
class Pool {
    struct Idle { void *p; uint64_t stamp; };
    std::mutex mu_;
    std::multimap<size_t, Idle> idle_;                // capacity -> block
    std::unordered_map<void *, size_t> capacity_;     // every block this pool created and has not released
    size_t cached_ = 0, live_ = 0, limit_ = (size_t)8 << 30;
    uint64_t clock_ = 0;

    // capacities keep four significant bits, so requests of similar size share blocks
    static size_t round_up(size_t n) {
        if (n < 4096) return (n + 511) / 512 * 512;
        int top = 63 - __builtin_clzll((unsigned long long)n);
        const size_t q = (size_t)1 << (top - 3);
        return (n + q - 1) / q * q;
    }
    void release_locked(size_t keep) {                // least recently returned first
        while (cached_ > keep && !idle_.empty()) {
            auto victim = idle_.begin();
            for (auto it = idle_.begin(); it != idle_.end(); ++it)
                if (it->second.stamp < victim->second.stamp) victim = it;
            (void)hipFree(victim->second.p);
            capacity_.erase(victim->second.p);
            cached_ -= victim->first;
            idle_.erase(victim);
        }
    }

public:
    void *take(size_t n) {
        const size_t cap = round_up(n);
        std::lock_guard<std::mutex> lock(mu_);
        auto it = idle_.lower_bound(cap);
        if (it != idle_.end() && it->first <= 2 * cap) {
            void *p = it->second.p;
            cached_ -= it->first; live_ += it->first;
            idle_.erase(it);
            return p;
        }
        void *p = nullptr;
        if (hipMalloc(&p, cap) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipStreamSynchronize(ctx().stream);
            release_locked(0);
            if (hipMalloc(&p, cap) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        }
        capacity_[p] = cap;
        live_ += cap;
        return p;
    }
    void give(void *p) {
        std::lock_guard<std::mutex> lock(mu_);
        auto it = capacity_.find(p);
        if (it == capacity_.end()) { (void)hipFree(p); return; }   // not ours: a caller's own hipMalloc
        const size_t cap = it->second;
        live_ -= cap;
        if (cap > limit_) { (void)hipFree(p); capacity_.erase(it); return; }
        idle_.emplace(cap, Idle{p, ++clock_});
        cached_ += cap;
        if (cached_ > limit_) release_locked(limit_);
    }
    void release_all() { std::lock_guard<std::mutex> lock(mu_); release_locked(0); }
    void set_limit(size_t n) { std::lock_guard<std::mutex> lock(mu_); limit_ = n; release_locked(limit_); }
    void stats(size_t *cached, size_t *live) {
        std::lock_guard<std::mutex> lock(mu_);
        if (cached) *cached = cached_;
        if (live) *live = live_;
    }
};
static Pool &pool() { static Pool p; return p; }

// ------------------------------------------------------ pattern auxiliaries --
static constexpr int PATTERN_AUX_SLOTS = 8;
static std::mutex g_aux_mu;
static PatternAux g_aux[PATTERN_AUX_SLOTS];
static uint64_t g_aux_clock = 0;

static void aux_drop_locked(PatternAux &e) {
    void *bufs[6] = {e.i32, e.tile_row[0], e.tile_row[1], e.bin_table, e.bin_tile_rows, e.bin_chunks};
    e = PatternAux();
    for (void *b : bufs) if (b) pool().give(b);         // (straight to the pool: gba_free would come back here)
}

PatternAux *pattern_aux(const gba_matrix *A) {
    static const bool on = [] { const char *e = getenv("GBA_PATTERN_CACHE"); return !(e && atoi(e) == 0); }();
    if (!on || !A || A->mem != GBA_MEM_DEVICE || !A->p || !A->i || A->nzombies || A->nvals < (int64_t)1 << 16) return nullptr;
    std::lock_guard<std::mutex> lock(g_aux_mu);
    PatternAux *lru = &g_aux[0];
    for (PatternAux &e : g_aux) {
        if (e.p == A->p && e.i == A->i && e.nvals == A->nvals && e.nvec == A->nvec && e.vlen == A->vlen) {
            e.sightings++; e.stamp = ++g_aux_clock;
            return &e;
        }
        if (e.stamp < lru->stamp) lru = &e;
    }
    aux_drop_locked(*lru);
    lru->p = A->p; lru->i = A->i; lru->nvals = A->nvals; lru->nvec = A->nvec; lru->vlen = A->vlen;
    lru->sightings = 1; lru->stamp = ++g_aux_clock;
    return lru;
}

void pattern_aux_forget_pointer(const void *ptr) {
    std::lock_guard<std::mutex> lock(g_aux_mu);
    for (PatternAux &e : g_aux) if (e.p && (e.p == ptr || e.i == ptr)) aux_drop_locked(e);
}

static void pattern_aux_clear() {
    std::lock_guard<std::mutex> lock(g_aux_mu);
    for (PatternAux &e : g_aux) aux_drop_locked(e);
}

}  // namespace gba

// ============================================================== C ABI ======
using namespace gba;

extern "C" {

gba_info gba_init(int device) {
    Context &c = ctx();
    if (c.inited && c.device == device) return GBA_SUCCESS;
    if (c.inited) gba_finalize();
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        GBA_FAIL(GBA_PANIC, "no HIP device is visible (%s): libgraphblas_amd has no CPU path", hipGetErrorString(e));
    if (device < 0 || device >= count) GBA_FAIL(GBA_INVALID_VALUE, "device %d out of range (0..%d)", device, count - 1);
    GBA_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    GBA_HIP(hipGetDeviceProperties(&prop, device));
    c.num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    GBA_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    GBA_HIP(hipEventCreate(&c.ev0));
    GBA_HIP(hipEventCreate(&c.ev1));
    GBA_HIP(hipEventCreate(&c.tm0));
    GBA_HIP(hipEventCreate(&c.tm1));
    GBA_OK(noalloc_bind_scratch(device));
    c.device = device;
    c.inited = true;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) { c.total_mem = total_b; pool().set_limit(total_b / 2 + total_b / 8); }
    if (const char *e = getenv("GBA_POOL_LIMIT_MB")) pool().set_limit((size_t)strtoull(e, nullptr, 10) << 20);
    return GBA_SUCCESS;
}

gba_info gba_finalize(void) {
    Context &c = ctx();
    if (!c.inited) return GBA_SUCCESS;
    hipStreamSynchronize(c.stream);
    pattern_aux_clear();
    pool().release_all();
    jit_unload_all();
    hipEventDestroy(c.ev0);
    hipEventDestroy(c.ev1);
    hipEventDestroy(c.tm0);
    hipEventDestroy(c.tm1);
    hipStreamDestroy(c.stream);
    c = Context();
    return GBA_SUCCESS;
}

const char *gba_last_error(void) { return g_err; }
const char *gba_version(void) { return "graphblas_amd 0.1 (gfx950); mirrors SuiteSparse:GraphBLAS 9.1.0 mxm/mxv/vxm"; }

void *gba_malloc(size_t nbytes) {
    if (!ctx().inited) { set_error("gba_malloc before gba_init"); return nullptr; }
    return pool().take(nbytes ? nbytes : 1);
}
void gba_free(void *p) { if (p) { pattern_aux_forget_pointer(p); pool().give(p); } }
gba_info gba_pool_release(void) { pool().release_all(); return GBA_SUCCESS; }
gba_info gba_pool_set_limit(size_t nbytes) { pool().set_limit(nbytes); return GBA_SUCCESS; }
gba_info gba_pool_stats(size_t *cached_bytes, size_t *live_bytes) {
    pool().stats(cached_bytes, live_bytes);
    return GBA_SUCCESS;
}


// ---- host memory the device can address: what GxB_init is given so that the arrays inside the CPU library's
// handles can be handed to this library without a copy through pageable memory (the role of rmm_wrap's managed
// pool for the reference's CUDA build, Source/GB_init.c:84-95; without XNACK a managed allocation stays in host
// memory, so pinned + mapped host memory is the same thing stated plainly).  A 64-byte header in front of every
// block keeps its size for realloc.
namespace {
struct HostHeader { uint64_t magic; uint64_t size; uint64_t pad[6]; };
static_assert(sizeof(HostHeader) == 64, "payload stays 64-byte aligned");
constexpr uint64_t HOST_MAGIC = 0x6762615f686f7374ull;      // "gba_host"
HostHeader *host_header(void *p) { return reinterpret_cast<HostHeader *>(static_cast<char *>(p) - sizeof(HostHeader)); }
}

void *gba_host_malloc(size_t nbytes) {
    void *raw = nullptr;
    if (hipHostMalloc(&raw, nbytes + sizeof(HostHeader), hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    HostHeader *h = static_cast<HostHeader *>(raw);
    h->magic = HOST_MAGIC; h->size = nbytes;
    return static_cast<char *>(raw) + sizeof(HostHeader);
}
void *gba_host_calloc(size_t count, size_t size) {
    if (size != 0 && count > SIZE_MAX / size) return nullptr;
    void *p = gba_host_malloc(count * size);
    if (p) memset(p, 0, count * size);
    return p;
}
void gba_host_free(void *p) {
    if (!p) return;
    HostHeader *h = host_header(p);
    if (h->magic != HOST_MAGIC) return;                         // not a block of this allocator: left alone
    h->magic = 0;
    (void)hipHostFree(h);
}
void *gba_host_realloc(void *p, size_t nbytes) {
    if (!p) return gba_host_malloc(nbytes);
    HostHeader *h = host_header(p);
    if (h->magic != HOST_MAGIC) return nullptr;
    if (nbytes <= h->size) { h->size = nbytes; return p; }
    void *q = gba_host_malloc(nbytes);
    if (!q) return nullptr;                                     // the old block stays valid, like realloc
    memcpy(q, p, h->size);
    gba_host_free(p);
    return q;
}
int gba_is_device_visible(const void *p) {
    if (!p) return 0;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return (a.type == hipMemoryTypeHost || a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged) ? 1 : 0;
}

gba_info gba_memcpy_h2d(void *dst, const void *src, size_t nbytes) {
    GBA_OK(require_init());
    if (nbytes == 0) return GBA_SUCCESS;
    GBA_HIP(hipMemcpyAsync(dst, src, nbytes, hipMemcpyHostToDevice, ctx().stream));
    GBA_HIP(hipStreamSynchronize(ctx().stream));
    return GBA_SUCCESS;
}
gba_info gba_memcpy_d2h(void *dst, const void *src, size_t nbytes) {
    GBA_OK(require_init());
    if (nbytes == 0) return GBA_SUCCESS;
    GBA_HIP(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToHost, ctx().stream));
    GBA_HIP(hipStreamSynchronize(ctx().stream));
    return GBA_SUCCESS;
}
gba_info gba_memset(void *dst, int byte, size_t nbytes) {
    GBA_OK(require_init());
    if (nbytes == 0) return GBA_SUCCESS;
    GBA_HIP(hipMemsetAsync(dst, byte, nbytes, ctx().stream));
    return GBA_SUCCESS;
}
gba_info gba_sync(void) {
    GBA_OK(require_init());
    GBA_HIP(hipStreamSynchronize(ctx().stream));
    return GBA_SUCCESS;
}
void *gba_stream(void) { return (void *)ctx().stream; }

gba_info gba_timer_start(void) {
    GBA_OK(require_init());
    GBA_HIP(hipEventRecord(ctx().tm0, ctx().stream));
    return GBA_SUCCESS;
}
gba_info gba_timer_stop(double *ms) {
    GBA_OK(require_init());
    GBA_HIP(hipEventRecord(ctx().tm1, ctx().stream));
    GBA_HIP(hipEventSynchronize(ctx().tm1));
    float f = 0.f;
    GBA_HIP(hipEventElapsedTime(&f, ctx().tm0, ctx().tm1));
    if (ms) *ms = (double)f;
    return GBA_SUCCESS;
}

void gba_matrix_free_content(gba_matrix *A) {
    if (!A) return;
    if (A->mem == GBA_MEM_DEVICE) {
        gba_free(A->p); gba_free(A->h); gba_free(A->i); gba_free(A->x); gba_free(A->b);
    }
    A->p = A->h = A->i = nullptr; A->x = nullptr; A->b = nullptr;
    A->nvals = 0; A->nzombies = 0;
}

gba_info gba_matrix_forget(const gba_matrix *A) {
    if (!A) return GBA_SUCCESS;
    if (A->p) pattern_aux_forget_pointer(A->p);
    if (A->i) pattern_aux_forget_pointer(A->i);
    return GBA_SUCCESS;
}

gba_info gba_cumsum(int64_t *count_dev, int64_t n) {
    GBA_OK(require_init());
    if (!count_dev) GBA_FAIL(GBA_NULL_POINTER, "count is NULL");
    return dev_cumsum_i64(count_dev, n, ctx().stream);
}

gba_info gba_cumsum_noalloc(int64_t *count_dev, int64_t n) {
    GBA_OK(require_init());
    if (!count_dev) GBA_FAIL(GBA_NULL_POINTER, "count is NULL");
    if (n < 0) GBA_FAIL(GBA_INVALID_VALUE, "cumsum: n < 0");
    if (n > GBA_CUMSUM_NOALLOC_MAX)
        GBA_FAIL(GBA_INVALID_VALUE, "gba_cumsum_noalloc: n = %ld exceeds the limit of %ld slots the built-in scratch array serves "
                 "(use gba_cumsum)", (long)n, (long)GBA_CUMSUM_NOALLOC_MAX);
    const int dev = ctx().device;
    int64_t *sc = (dev >= 0 && dev < NOALLOC_MAX_DEVICES) ? g_scratch_ptr[dev] : nullptr;
    if (!sc) GBA_FAIL(GBA_PANIC, "gba_cumsum_noalloc: no scratch array bound for device %d", dev);
    return noalloc_scan(count_dev, n, sc, ctx().stream);
}

}  // extern "C"
