// Host-side plumbing shared by every translation unit of libgraphblas_amd.so:
// error reporting, the per-process device context, device buffers, the
// canonical semiring and its dispatch onto kernel template instances.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <cstdio>
#include <cstdarg>
#include "../../include/graphblas_amd.h"
#include "gba_ops.hpp"

namespace gba {

// ------------------------------------------------------------------ errors --
void set_error(const char *fmt, ...);
std::string last_error();              // the calling thread's message (a copy: cleanup code may overwrite it)
#define GBA_FAIL(code, ...) do { ::gba::set_error(__VA_ARGS__); return (code); } while (0)
#define GBA_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    ::gba::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
    return (e_ == hipErrorOutOfMemory) ? GBA_OUT_OF_MEMORY : GBA_PANIC; } } while (0)
#define GBA_OK(call) do { gba_info i_ = (call); if (i_ != GBA_SUCCESS) return i_; } while (0)

// ----------------------------------------------------------------- context --
struct Context {
    size_t udt_size = 0;            // element size of GBA_UDT matrices while a user-defined call is running
    bool inited = false;
    int device = -1;
    hipStream_t stream = nullptr;
    int num_cu = 256;
    size_t total_mem = 0;           // bytes of device memory (read once in gba_init)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;        // KernelTimer (inside the library)
    hipEvent_t tm0 = nullptr, tm1 = nullptr;        // gba_timer_start / gba_timer_stop (the caller's stopwatch)
};
Context &ctx();
gba_info require_init();
void jit_unload_all();          // gba_jit.hip: unload every hipRTC module (gba_finalize)
// bytes per entry of a matrix, user-defined types included
inline size_t esize_of(int type) { return type == GBA_UDT ? ctx().udt_size : type_size(type); }

// RAII device buffer (freed with gba_free unless released)
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() {}
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { reset(); }
    gba_info alloc(size_t n) {
        reset();
        bytes = n;
        p = gba_malloc(n ? n : 1);
        if (!p) { set_error("device allocation of %zu bytes failed", n); return GBA_OUT_OF_MEMORY; }
        return GBA_SUCCESS;
    }
    void reset() { if (p) gba_free(p); p = nullptr; bytes = 0; }
    void *release() { void *q = p; p = nullptr; bytes = 0; return q; }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// device timer over the library stream
struct KernelTimer {
    bool on;
    explicit KernelTimer(bool enable) : on(enable) { if (on) hipEventRecord(ctx().ev0, ctx().stream); }
    double stop_ms() {
        if (!on) return 0.0;
        hipEventRecord(ctx().ev1, ctx().stream);
        hipEventSynchronize(ctx().ev1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, ctx().ev0, ctx().ev1);
        return (double)ms;
    }
};

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// matrix predicates, named after the reference macros GB_IS_SPARSE/BITMAP/FULL
inline bool is_sparse(const gba_matrix *A) { return A && A->sparsity == GBA_SPARSE; }
inline bool is_real_type(int t) { return t >= GBA_BOOL && t <= GBA_FP64; }     // built-in, not complex
inline bool is_hyper(const gba_matrix *A)  { return A && A->sparsity == GBA_HYPERSPARSE; }
inline bool is_bitmap(const gba_matrix *A) { return A && A->sparsity == GBA_BITMAP; }
inline bool is_full(const gba_matrix *A)   { return A && A->sparsity == GBA_FULL; }
inline bool is_dense_fmt(const gba_matrix *A) { return is_bitmap(A) || is_full(A); }
// all entries present (GB_as_if_full, Source/GB_is_dense.c)
inline bool as_if_full(const gba_matrix *A) {
    if (!A) return false;
    if (is_full(A)) return true;
    if (A->nzombies > 0) return false;
    double total = (double)A->vlen * (double)A->vdim;
    return (double)A->nvals == total;
}
inline int64_t nnz_held(const gba_matrix *A) {
    if (is_full(A)) return A->vlen * A->vdim;
    return A->nvals;
}

// ------------------------------------------------------- canonical semiring --
// A semiring after GB_flip_binop (Source/GB_flip_binop.c:355-375), the boolean
// renames (Source/GB_boolean_rename.c:45-97) and the SECONDI==FIRSTJ folding
// (Source/Factories/GB_AxB_factory.c:332-351).
struct Canon {
    int add = 0, mult = 0;
    int xtype = 0, ztype = 0;
    int kind = KIND_TT;
    bool a_pattern = false, b_pattern = false;
};

// returns GBA_NOT_IMPLEMENTED when the semiring is not in the device set
gba_info canonicalize(const gba_semiring *s, bool flipxy, int atype, int btype, Canon *out);
int flip_binop(int op, bool *flipxy);
int boolean_rename(int op);

// Call f(S{}) with the kernel semiring class for c.  HOT semirings get
// compile-time operators, everything else the run-time class of its types.
template <class F>
gba_info with_semiring(const Canon &c, F &&f) {
    const int add = c.add, mult = c.mult;
#ifdef GBA_DEV_FAST         // tuning builds: one semiring only, so that a kernel change compiles in seconds
    if (c.kind == KIND_TT && c.ztype == GBA_FP64 && add == GBA_PLUS && mult == GBA_TIMES) return f(Sct<GBA_PLUS, GBA_TIMES, double, double>{});
    GBA_FAIL(GBA_NOT_IMPLEMENTED, "tuning build: PLUS_TIMES_FP64 only");
#else
    if (c.kind == KIND_POS) {
        if (c.ztype == GBA_INT64) {
            if (add == GBA_ANY && mult == GBA_FIRSTJ) return f(Sct<GBA_ANY, GBA_FIRSTJ, uint8_t, int64_t>{});
            return f(Srt<uint8_t, int64_t, KIND_POS>{add, mult, true, true});
        }
        if (c.ztype == GBA_INT32) return f(Srt<uint8_t, int32_t, KIND_POS>{add, mult, true, true});
        GBA_FAIL(GBA_NOT_IMPLEMENTED, "positional semiring with ztype %d", c.ztype);
    }
    if (c.kind == KIND_CMP) {
        switch (c.xtype) {
#define GBA_CMP_CASE(code, T) case code: return f(Srt<T, bool, KIND_CMP>{add, mult, false, false});
            GBA_CMP_CASE(GBA_INT8, int8_t) GBA_CMP_CASE(GBA_UINT8, uint8_t)
            GBA_CMP_CASE(GBA_INT16, int16_t) GBA_CMP_CASE(GBA_UINT16, uint16_t)
            GBA_CMP_CASE(GBA_INT32, int32_t) GBA_CMP_CASE(GBA_UINT32, uint32_t)
            GBA_CMP_CASE(GBA_INT64, int64_t) GBA_CMP_CASE(GBA_UINT64, uint64_t)
            GBA_CMP_CASE(GBA_FP32, float) GBA_CMP_CASE(GBA_FP64, double)
#undef GBA_CMP_CASE
            default: GBA_FAIL(GBA_NOT_IMPLEMENTED, "comparator semiring with xtype %d", c.xtype);
        }
    }
    // KIND_TT
    if (c.ztype == GBA_FP64) {
        if (add == GBA_PLUS && mult == GBA_TIMES) return f(Sct<GBA_PLUS, GBA_TIMES, double, double>{});
        if (add == GBA_MIN && mult == GBA_PLUS) return f(Sct<GBA_MIN, GBA_PLUS, double, double>{});
    }
    if (c.ztype == GBA_FP32) {
        if (add == GBA_PLUS && mult == GBA_TIMES) return f(Sct<GBA_PLUS, GBA_TIMES, float, float>{});
    }
    if (c.ztype == GBA_INT64) {
        if (add == GBA_PLUS && mult == GBA_PAIR) return f(Sct<GBA_PLUS, GBA_PAIR, int64_t, int64_t>{});
    }
    switch (c.ztype) {
#define GBA_TT_CASE(code, T) case code: return f(Srt<T, T, KIND_TT>{add, mult, c.a_pattern, c.b_pattern});
        GBA_TT_CASE(GBA_BOOL, bool)
        GBA_TT_CASE(GBA_INT8, int8_t) GBA_TT_CASE(GBA_UINT8, uint8_t)
        GBA_TT_CASE(GBA_INT16, int16_t) GBA_TT_CASE(GBA_UINT16, uint16_t)
        GBA_TT_CASE(GBA_INT32, int32_t) GBA_TT_CASE(GBA_UINT32, uint32_t)
        GBA_TT_CASE(GBA_INT64, int64_t) GBA_TT_CASE(GBA_UINT64, uint64_t)
        GBA_TT_CASE(GBA_FP32, float) GBA_TT_CASE(GBA_FP64, double)
#undef GBA_TT_CASE
        default: GBA_FAIL(GBA_NOT_IMPLEMENTED, "semiring with ztype %d", c.ztype);
    }
#endif
}

// ------------------------------------------------------- shared device ops --
gba_info dev_cumsum_i64(int64_t *count, int64_t n, hipStream_t st);       // exclusive, count[n] = total
gba_info dev_fill_i64(int64_t *p, int64_t n, int64_t v, hipStream_t st);
gba_info dev_fill_bytes(void *p, size_t n, int byte, hipStream_t st);
gba_info dev_max_diff_i64(const int64_t *p, int64_t n, int64_t *out_host);  // max over k of p[k+1]-p[k]
gba_info dev_read_i64(const int64_t *p, int64_t *out_host);
gba_info dev_count_nonempty(const int64_t *p, int64_t nvec, int64_t *out_host);
gba_info dev_is_diagonal(const gba_matrix *A, bool *out);
gba_info dev_count_ones(const int8_t *b, int64_t n, int64_t *out_host);

// method drivers (same contracts as the extern "C" entry points, operands on device)
gba_info saxpy3(gba_matrix *C, const gba_matrix *M, bool Mask_comp, bool Mask_struct,
                const gba_matrix *A, const gba_matrix *B, const Canon &c,
                int axb_method, int do_sort, bool *mask_applied, gba_report *rep, bool may_discard_mask = false);
gba_info saxpy3_flopcount(int64_t *Mwork, int64_t *Bflops, const gba_matrix *M, bool Mask_comp,
                          const gba_matrix *A, const gba_matrix *B);
gba_info dot3(gba_matrix *C, const gba_matrix *M, bool Mask_struct, const gba_matrix *A,
              const gba_matrix *B, const Canon &c, gba_report *rep);
gba_info dot2(gba_matrix *C, const gba_matrix *M, bool Mask_comp, bool Mask_struct,
              const gba_matrix *A, const gba_matrix *B, const Canon &c, gba_report *rep);
gba_info dot4(gba_matrix *C, const gba_matrix *A, const gba_matrix *B, const Canon &c, gba_report *rep);
gba_info saxpy4(gba_matrix *C, const gba_matrix *A, const gba_matrix *B, const Canon &c, gba_report *rep);
gba_info saxpy5(gba_matrix *C, const gba_matrix *A, const gba_matrix *B, const Canon &c, gba_report *rep);
gba_info saxbit(gba_matrix *C, const gba_matrix *M, bool Mask_comp, bool Mask_struct,
                const gba_matrix *A, const gba_matrix *B, const Canon &c, gba_report *rep);
// built-in complex semirings (gba_jit.hip): compiled at run time like a user-defined type
gba_info mxm_complex(gba_matrix *C, const gba_matrix *M, const gba_binop *accum, const gba_semiring *s,
                     const gba_matrix *A, bool A_tran, const gba_matrix *B, bool B_tran, bool flipxy,
                     bool C_replace, bool Mask_comp, bool Mask_struct, gba_report *report);
// gba_meta.hip: the accum/mask write-back on INT64 matrices that hold positions instead of values
gba_info writeback_positions(gba_matrix *Cpos, const gba_matrix *M, bool has_accum, bool C_replace, bool Mask_comp,
                             bool Mask_struct, gba_matrix *Tpos);
// gba_meta.hip: X with its values cast to another built-in type (real to real, anything to complex)
gba_info cast_operand(const gba_matrix *X, int to_type, gba_matrix *out, std::vector<void *> &owned);
// gba_meta.hip: C<M> = accum (C,T) for a product T of C's type (products computed in the other type family and cast)
gba_info mxm_assign_cast(gba_matrix *C, const gba_matrix *M, const gba_binop *accum, bool C_replace, bool Mask_comp,
                         bool Mask_struct, const gba_matrix *T, gba_report *report);
inline bool is_complex_type(int t) { return t == GBA_FC32 || t == GBA_FC64; }
gba_info transpose(gba_matrix *AT, const gba_matrix *A);
gba_info to_sparse(const gba_matrix *X, gba_matrix *S, std::vector<void *> &owned);   // bitmap/full -> sparse copy
// hypersparse -> sparse view over all vdim vectors (new p; i and x shared with X)
constexpr int64_t HYPER_EXPAND_MAX = (int64_t)1 << 31;
gba_info hyper_to_sparse(const gba_matrix *X, gba_matrix *S, std::vector<void *> &owned);
gba_info matrix_wait(gba_matrix *C);
gba_info ranks_in_list(const int64_t *idx, int64_t n, const int64_t *U, int64_t nu, int64_t *out);   // rank in the ascending U, or -1
gba_info indices_of_ranks(int64_t *idx, int64_t n, const int64_t *U);                              // the way back, in place
gba_info ranked_copy(const gba_matrix *X, const int64_t *U, int64_t nu, int64_t new_vlen, bool with_values,
                     gba_matrix *out, std::vector<void *> &owned);       // indices -> ranks in U, other entries dropped
gba_info distinct_sorted(const int64_t *idx, int64_t n, int bits, DevBuf *out, int64_t *count);   // ascending distinct values
gba_info nonempty_vectors(const gba_matrix *C, int64_t *k);
gba_info sparse_to_hyper(gba_matrix *C);     // C owns its arrays; h lists the non-empty vectors afterwards
gba_info sorted_copy(const gba_matrix *X, gba_matrix *out, std::vector<void *> &owned);   // sorted, zombie-free copy of a sparse X

// stage a host-resident view into device memory (PCIe); owned[] collects the
// buffers to free afterwards
// ---- what the library remembers about the PATTERN of sparse matrices it has multiplied by before -------------
// The reference keeps such auxiliaries inside the matrix object (A->Y, the hashed hyperlist,
// Source/Shared/GB_matrix.h:240-306: built on first use, dropped when the matrix changes).  Here the matrix is a
// caller's view, so they are kept beside it, keyed by the addresses of p and i, nvals, nvec and vlen.  Only the
// pattern is remembered (values may change freely).  An entry is dropped when one of its arrays is given back with
// gba_free, by gba_matrix_forget, or when it is the least recently used of PATTERN_AUX_SLOTS.
// a remembered 32-bit index copy is kept only for patterns of at most this many entries (4 GiB of copy)
static constexpr int64_t PATTERN_I32_MAX_ENTRIES = (int64_t)1 << 30;
struct PatternAux {
    const void *p = nullptr, *i = nullptr;
    int64_t nvals = 0, nvec = 0, vlen = 0;
    int32_t *i32 = nullptr;             // the indices in 32 bits (vlen < 2^31), or null
    int64_t *tile_row[2] = {nullptr, nullptr};      // first vector of every tile of tile_size[k] entries, or null
    int64_t tile_size[2] = {0, 0};
    // the plan of the destination-binned scatter (saxpy4 with a sparse A and full B, C): where every tile's pairs
    // go per bin, the first vector of every tile, the chunks of every bin.  Depends on the pattern only.
    int64_t *bin_table = nullptr, *bin_tile_rows = nullptr;
    int *bin_chunks = nullptr;
    int bin_shift = 0;
    int64_t bin_tile = 0, bin_cvlen = 0;
    uint64_t sightings = 0, stamp = 0;
};
// the entry of A, created (empty) when absent; null when the cache is off (GBA_PATTERN_CACHE=0) or A does not qualify
PatternAux *pattern_aux(const gba_matrix *A);
void pattern_aux_forget_pointer(const void *ptr);       // called by gba_free

// device view of a host operand.  GBA_MEM_HOST: copied.  GBA_MEM_HOST_VISIBLE (pinned memory mapped into the
// device's address space): read in place when `in_place` (each entry is read about once), copied otherwise.
gba_info stage_to_device(const gba_matrix *in, gba_matrix *out, std::vector<void *> &owned, bool in_place = false);

}  // namespace gba
