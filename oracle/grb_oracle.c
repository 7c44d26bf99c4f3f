/*
 * grb_oracle.c -- TEST INFRASTRUCTURE ONLY.  A plain-C, CPU restatement of the
 * algorithms on the sparse matrix-multiply path of SuiteSparse:GraphBLAS 9.1.0.
 * Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may
 * load this library; the product (libgraphblas_amd.so) never does.
 *
 * Parity pin: this restatement is checked against the reference's own
 * known-answer material for this path (Demo/Output/gauss_demo.out:223-405 --
 * saxbit, dot3, colscale, rowscale, dot4, saxpy4, saxpy5 on a user-defined type
 * -- replayed by oracle/gauss_replay.c; Demo/Output/wildtype_demo.out:142-330 --
 * saxpy3 and dot3 on a 192-byte user-defined type -- replayed by
 * oracle/wildtype_replay.c; tests/test_oracle.py compares both with fixtures
 * transcribed from those files), and against results of the reference library
 * itself for its built-in semirings: entry counts, checksums, multiply counts,
 * triangle counts and BFS frontier sizes it produced on RMAT inputs that
 * graphblas_amd/gen.py rebuilds edge for edge
 * (tests/golden/reference_run_checksums.json, tests/test_reference_runs.py).
 * The reference library itself cannot be built in this environment: its
 * sources need GB_config.h and GB_JITpackage.c, which only its cmake build
 * generates.
 *
 * Every function names the reference code it follows.  Matrices use the POD
 * view of include/graphblas_amd.h with host arrays; all operands are
 * CSC-agnostic "vector" matrices as the reference's method drivers see them.
 */
#include <stdint.h>
#include <stdbool.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include <complex.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/graphblas_amd.h"
#include "grb_oracle.h"

#define ORC_MAX_VALUE 256            /* bytes of the largest value the scratch buffers hold (user-defined types) */

/* ------------------------------------------------------------ operators -- */
/* z = f(x,y) on raw bytes, generated for every built-in type.  Integer
 * arithmetic wraps (the reference compiles with -fwrapv,
 * cmake_modules/GraphBLAS_compiler_options.cmake:23-29); integer division is
 * GB_idiv_* (Source/GB_math.h:27-260); fp min/max are fmin/fmax
 * (Source/FactoryKernels/GB_AxB__min_plus_fp64.c:16-19). */

#define SIGNED_TYPES(X)   X(INT8, int8_t, uint8_t, INT8_MIN, INT8_MAX) X(INT16, int16_t, uint16_t, INT16_MIN, INT16_MAX) \
                          X(INT32, int32_t, uint32_t, INT32_MIN, INT32_MAX) X(INT64, int64_t, uint64_t, INT64_MIN, INT64_MAX)
#define UNSIGNED_TYPES(X) X(UINT8, uint8_t, UINT8_MAX) X(UINT16, uint16_t, UINT16_MAX) X(UINT32, uint32_t, UINT32_MAX) X(UINT64, uint64_t, UINT64_MAX)

#define DEF_BIN(name, T, ZT, expr) \
    static void name(void *z_, const void *x_, const void *y_) { \
        const T x = *(const T *)x_, y = *(const T *)y_; (void)x; (void)y; *(ZT *)z_ = (ZT)(expr); }

/* ops common to all numeric types; PLUS/MINUS/TIMES/DIV supplied per family */
#define DEF_COMMON(TAG, T) \
    DEF_BIN(f_first_##TAG, T, T, x) DEF_BIN(f_second_##TAG, T, T, y) DEF_BIN(f_pair_##TAG, T, T, 1) \
    DEF_BIN(f_iseq_##TAG, T, T, x == y) DEF_BIN(f_isne_##TAG, T, T, x != y) DEF_BIN(f_isgt_##TAG, T, T, x > y) \
    DEF_BIN(f_islt_##TAG, T, T, x < y) DEF_BIN(f_isge_##TAG, T, T, x >= y) DEF_BIN(f_isle_##TAG, T, T, x <= y) \
    DEF_BIN(f_lor_##TAG, T, T, (x != 0) || (y != 0)) DEF_BIN(f_land_##TAG, T, T, (x != 0) && (y != 0)) \
    DEF_BIN(f_lxor_##TAG, T, T, (x != 0) != (y != 0)) \
    DEF_BIN(f_eq_##TAG, T, bool, x == y) DEF_BIN(f_ne_##TAG, T, bool, x != y) DEF_BIN(f_gt_##TAG, T, bool, x > y) \
    DEF_BIN(f_lt_##TAG, T, bool, x < y) DEF_BIN(f_ge_##TAG, T, bool, x >= y) DEF_BIN(f_le_##TAG, T, bool, x <= y)

#define DEF_SIGNED(TAG, T, U, TMIN, TMAX) \
    DEF_COMMON(TAG, T) \
    DEF_BIN(f_min_##TAG, T, T, x < y ? x : y) DEF_BIN(f_max_##TAG, T, T, x > y ? x : y) \
    DEF_BIN(f_plus_##TAG, T, T, (T)((U)x + (U)y)) DEF_BIN(f_minus_##TAG, T, T, (T)((U)x - (U)y)) \
    DEF_BIN(f_rminus_##TAG, T, T, (T)((U)y - (U)x)) DEF_BIN(f_times_##TAG, T, T, (T)((uint64_t)(U)x * (uint64_t)(U)y)) \
    static T idiv_##TAG(T x, T y) { if (y == -1) return (T)(0 - (U)x); if (y == 0) return (x == 0) ? 0 : ((x < 0) ? TMIN : TMAX); return (T)(x / y); } \
    DEF_BIN(f_div_##TAG, T, T, idiv_##TAG(x, y)) DEF_BIN(f_rdiv_##TAG, T, T, idiv_##TAG(y, x))
#define DEF_UNSIGNED(TAG, T, TMAX) \
    DEF_COMMON(TAG, T) \
    DEF_BIN(f_min_##TAG, T, T, x < y ? x : y) DEF_BIN(f_max_##TAG, T, T, x > y ? x : y) \
    DEF_BIN(f_plus_##TAG, T, T, (T)(x + y)) DEF_BIN(f_minus_##TAG, T, T, (T)(x - y)) \
    DEF_BIN(f_rminus_##TAG, T, T, (T)(y - x)) DEF_BIN(f_times_##TAG, T, T, (T)((uint64_t)x * (uint64_t)y)) \
    static T idiv_##TAG(T x, T y) { if (y == 0) return (x == 0) ? 0 : TMAX; return (T)(x / y); } \
    DEF_BIN(f_div_##TAG, T, T, idiv_##TAG(x, y)) DEF_BIN(f_rdiv_##TAG, T, T, idiv_##TAG(y, x)) \
    DEF_BIN(f_bor_##TAG, T, T, x | y) DEF_BIN(f_band_##TAG, T, T, x & y) DEF_BIN(f_bxor_##TAG, T, T, x ^ y) \
    DEF_BIN(f_bxnor_##TAG, T, T, ~(x ^ y))
#define DEF_FLOAT(TAG, T, FMIN, FMAX) \
    DEF_COMMON(TAG, T) \
    DEF_BIN(f_min_##TAG, T, T, FMIN(x, y)) DEF_BIN(f_max_##TAG, T, T, FMAX(x, y)) \
    DEF_BIN(f_plus_##TAG, T, T, x + y) DEF_BIN(f_minus_##TAG, T, T, x - y) DEF_BIN(f_rminus_##TAG, T, T, y - x) \
    DEF_BIN(f_times_##TAG, T, T, x * y) DEF_BIN(f_div_##TAG, T, T, x / y) DEF_BIN(f_rdiv_##TAG, T, T, y / x)

SIGNED_TYPES(DEF_SIGNED)
UNSIGNED_TYPES(DEF_UNSIGNED)
DEF_FLOAT(FP32, float, fminf, fmaxf)
DEF_FLOAT(FP64, double, fmin, fmax)
/* bool: after the renames of Source/GB_boolean_rename.c:45-97 only these remain */
DEF_BIN(f_first_BOOL, bool, bool, x) DEF_BIN(f_second_BOOL, bool, bool, y) DEF_BIN(f_pair_BOOL, bool, bool, 1)
DEF_BIN(f_lor_BOOL, bool, bool, x || y) DEF_BIN(f_land_BOOL, bool, bool, x && y) DEF_BIN(f_lxor_BOOL, bool, bool, x != y)
DEF_BIN(f_eq_BOOL, bool, bool, x == y) DEF_BIN(f_gt_BOOL, bool, bool, x > y) DEF_BIN(f_lt_BOOL, bool, bool, x < y)
DEF_BIN(f_ge_BOOL, bool, bool, x >= y) DEF_BIN(f_le_BOOL, bool, bool, x <= y)

/* complex: the C99 operators of the non-MSVC branch of Source/Shared/GB_complex.h:237-246 (x*y, x+y, x-y on
 * float/double complex), PAIR = 1+0i (Source/Factories/GB_ops_template.h, GxB_CMPLX (1,0)), and the
 * library's own division -- never the compiler's: GB_FC64_div, three cases after R. L. Smith, ACM Algorithm
 * 116 (Source/GB_math.h:301-340); single complex divides in double and rounds (Source/GB_math.h:379-389).
 * The semirings offered are (PLUS, TIMES, ANY) x (FIRST, SECOND, PAIR, PLUS, MINUS, RMINUS, TIMES, DIV,
 * RDIV), Source/Factories/GB_AxB_factory.c:46-243 with GB_COMPLEX, GB_AxB_type_factory.c:100-154. */
typedef float _Complex fc32_t;
typedef double _Complex fc64_t;
static fc64_t fc64_div(fc64_t x, fc64_t y) {
    const double xr = creal(x), xi = cimag(x), yr = creal(y), yi = cimag(y);
    const int yr_class = fpclassify(yr), yi_class = fpclassify(yi);
    if (yi_class == FP_ZERO) return CMPLX(xr / yr, xi / yr);
    if (yr_class == FP_ZERO) return CMPLX(xi / yi, -xr / yi);
    if (yi_class == FP_INFINITE && yr_class == FP_INFINITE) {
        const double r = (signbit(yr) == signbit(yi)) ? 1 : -1, d = yr + r * yi;
        return CMPLX((xr + xi * r) / d, (xi - xr * r) / d);
    }
    if (fabs(yr) >= fabs(yi)) { const double r = yi / yr, d = yr + r * yi; return CMPLX((xr + xi * r) / d, (xi - xr * r) / d); }
    const double r = yr / yi, d = r * yr + yi;
    return CMPLX((xr * r + xi) / d, (xi * r - xr) / d);
}
static fc32_t fc32_div(fc32_t x, fc32_t y) {
    const fc64_t z = fc64_div(CMPLX((double)crealf(x), (double)cimagf(x)), CMPLX((double)crealf(y), (double)cimagf(y)));
    return CMPLXF((float)creal(z), (float)cimag(z));
}
#define DEF_COMPLEX(TAG, T, ONE, DIV) \
    DEF_BIN(f_first_##TAG, T, T, x) DEF_BIN(f_second_##TAG, T, T, y) DEF_BIN(f_pair_##TAG, T, T, ONE) \
    DEF_BIN(f_plus_##TAG, T, T, x + y) DEF_BIN(f_minus_##TAG, T, T, x - y) DEF_BIN(f_rminus_##TAG, T, T, y - x) \
    DEF_BIN(f_times_##TAG, T, T, x * y) DEF_BIN(f_div_##TAG, T, T, DIV(x, y)) DEF_BIN(f_rdiv_##TAG, T, T, DIV(y, x))
DEF_COMPLEX(FC32, fc32_t, CMPLXF(1, 0), fc32_div)
DEF_COMPLEX(FC64, fc64_t, CMPLX(1, 0), fc64_div)

typedef void (*binfn)(void *, const void *, const void *);

size_t orc_type_size(int code) {
    switch (code) {
        case GBA_BOOL: case GBA_INT8: case GBA_UINT8: return 1;
        case GBA_INT16: case GBA_UINT16: return 2;
        case GBA_INT32: case GBA_UINT32: case GBA_FP32: return 4;
        case GBA_INT64: case GBA_UINT64: case GBA_FP64: case GBA_FC32: return 8;
        case GBA_FC64: return 16;
        default: return 0;
    }
}

#define CASE_NUM(OP, fname, TAG) case GBA_##TAG: return f_##fname##_##TAG;
#define NUM_SWITCH(OP, fname) switch (type) { \
    CASE_NUM(OP, fname, INT8) CASE_NUM(OP, fname, UINT8) CASE_NUM(OP, fname, INT16) CASE_NUM(OP, fname, UINT16) \
    CASE_NUM(OP, fname, INT32) CASE_NUM(OP, fname, UINT32) CASE_NUM(OP, fname, INT64) CASE_NUM(OP, fname, UINT64) \
    CASE_NUM(OP, fname, FP32) CASE_NUM(OP, fname, FP64) default: return NULL; }
#define UINT_SWITCH(fname) switch (type) { \
    case GBA_UINT8: return f_##fname##_UINT8; case GBA_UINT16: return f_##fname##_UINT16; \
    case GBA_UINT32: return f_##fname##_UINT32; case GBA_UINT64: return f_##fname##_UINT64; default: return NULL; }

/* binary operator on operand type `type` (already renamed if bool) */
static binfn lookup_binop(int op, int type) {
    if (type == GBA_BOOL) {
        switch (op) {
            case GBA_FIRST: return f_first_BOOL; case GBA_SECOND: return f_second_BOOL; case GBA_PAIR: return f_pair_BOOL;
            case GBA_ANY: return f_first_BOOL;
            case GBA_LOR: return f_lor_BOOL; case GBA_LAND: return f_land_BOOL; case GBA_LXOR: return f_lxor_BOOL;
            case GBA_EQ: return f_eq_BOOL; case GBA_GT: return f_gt_BOOL; case GBA_LT: return f_lt_BOOL;
            case GBA_GE: return f_ge_BOOL; case GBA_LE: return f_le_BOOL;
            default: return NULL;
        }
    }
    if (type == GBA_FC32 || type == GBA_FC64) {
#define CPLX(fname) return type == GBA_FC32 ? f_##fname##_FC32 : f_##fname##_FC64
        switch (op) {
            case GBA_FIRST: case GBA_ANY: CPLX(first);   case GBA_SECOND: CPLX(second);   case GBA_PAIR: CPLX(pair);
            case GBA_PLUS: CPLX(plus);   case GBA_MINUS: CPLX(minus);   case GBA_RMINUS: CPLX(rminus);
            case GBA_TIMES: CPLX(times); case GBA_DIV: CPLX(div);       case GBA_RDIV: CPLX(rdiv);
            default: return NULL;
        }
#undef CPLX
    }
    switch (op) {
        case GBA_FIRST: NUM_SWITCH(op, first)   case GBA_SECOND: NUM_SWITCH(op, second)
        case GBA_ANY: NUM_SWITCH(op, first)     case GBA_PAIR: NUM_SWITCH(op, pair)
        case GBA_MIN: NUM_SWITCH(op, min)       case GBA_MAX: NUM_SWITCH(op, max)
        case GBA_PLUS: NUM_SWITCH(op, plus)     case GBA_MINUS: NUM_SWITCH(op, minus)
        case GBA_RMINUS: NUM_SWITCH(op, rminus) case GBA_TIMES: NUM_SWITCH(op, times)
        case GBA_DIV: NUM_SWITCH(op, div)       case GBA_RDIV: NUM_SWITCH(op, rdiv)
        case GBA_ISEQ: NUM_SWITCH(op, iseq)     case GBA_ISNE: NUM_SWITCH(op, isne)
        case GBA_ISGT: NUM_SWITCH(op, isgt)     case GBA_ISLT: NUM_SWITCH(op, islt)
        case GBA_ISGE: NUM_SWITCH(op, isge)     case GBA_ISLE: NUM_SWITCH(op, isle)
        case GBA_LOR: NUM_SWITCH(op, lor)       case GBA_LAND: NUM_SWITCH(op, land)
        case GBA_LXOR: NUM_SWITCH(op, lxor)
        case GBA_EQ: NUM_SWITCH(op, eq)         case GBA_NE: NUM_SWITCH(op, ne)
        case GBA_GT: NUM_SWITCH(op, gt)         case GBA_LT: NUM_SWITCH(op, lt)
        case GBA_GE: NUM_SWITCH(op, ge)         case GBA_LE: NUM_SWITCH(op, le)
        case GBA_BOR: UINT_SWITCH(bor)          case GBA_BAND: UINT_SWITCH(band)
        case GBA_BXOR: UINT_SWITCH(bxor)        case GBA_BXNOR: UINT_SWITCH(bxnor)
        default: return NULL;
    }
}

/* GB_flip_binop (Source/GB_flip_binop.c:355-375) */
static int flip_op(int op) {
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
        default: return op;
    }
}

/* GB_boolean_rename (Source/GB_boolean_rename.c:45-97) */
static int bool_rename(int op) {
    switch (op) {
        case GBA_DIV: return GBA_FIRST;     case GBA_RDIV: return GBA_SECOND;
        case GBA_MIN: case GBA_TIMES: return GBA_LAND;
        case GBA_MAX: case GBA_PLUS: return GBA_LOR;
        case GBA_MINUS: case GBA_RMINUS: case GBA_NE: case GBA_ISNE: return GBA_LXOR;
        case GBA_ISEQ: return GBA_EQ;       case GBA_ISGT: return GBA_GT;   case GBA_ISLT: return GBA_LT;
        case GBA_POW: case GBA_ISGE: return GBA_GE;   case GBA_ISLE: return GBA_LE;
        default: return op;
    }
}

static bool op_positional(int op) { return op >= GBA_FIRSTI && op <= GBA_SECONDJ1; }

/* Build the working semiring from a built-in description; flipxy is removed by
 * flipping the multiplier (Source/GB_AxB_meta.c:456-468). */
int orc_semiring_builtin(orc_semiring *s, const gba_semiring *in, bool flipxy) {
    memset(s, 0, sizeof(*s));
    int mult = in->mult, add = in->add;
    if (flipxy) mult = flip_op(mult);
    s->ztype = in->ztype; s->zsize = orc_type_size(in->ztype);
    if (op_positional(mult)) {
        if (mult == GBA_SECONDI) mult = GBA_FIRSTJ;
        if (mult == GBA_SECONDI1) mult = GBA_FIRSTJ1;
        s->positional = mult; s->a_pattern = s->b_pattern = true;
        s->xtype = s->ytype = GBA_UINT8; s->xsize = s->ysize = 1;
        s->fadd = lookup_binop(add, in->ztype);
        s->add_op = add; s->mult_op = mult;
        return (s->fadd && (in->ztype == GBA_INT32 || in->ztype == GBA_INT64)) ? 0 : -1;
    }
    int xtype = in->xtype;
    s->a_pattern = (mult == GBA_SECOND || mult == GBA_PAIR);
    s->b_pattern = (mult == GBA_FIRST || mult == GBA_PAIR);
    if (mult == GBA_PAIR) xtype = in->ztype;
    if (s->a_pattern && !s->b_pattern) xtype = in->ytype;
    if (xtype == GBA_BOOL) mult = bool_rename(mult);
    if (in->ztype == GBA_BOOL) {
        if (add == GBA_MIN || add == GBA_TIMES) add = GBA_LAND;
        else if (add == GBA_MAX || add == GBA_PLUS) add = GBA_LOR;
    }
    s->xtype = s->ytype = xtype; s->xsize = s->ysize = orc_type_size(xtype);
    s->fmult = lookup_binop(mult, xtype);
    s->fadd = lookup_binop(add, in->ztype);
    s->add_op = add; s->mult_op = mult;
    return (s->fmult && s->fadd) ? 0 : -1;
}

void orc_semiring_user(orc_semiring *s, orc_binfn fadd, orc_binfn fmult, size_t xsize, size_t ysize, size_t zsize) {
    memset(s, 0, sizeof(*s));
    if (xsize > ORC_MAX_VALUE || ysize > ORC_MAX_VALUE || zsize > ORC_MAX_VALUE) { fprintf(stderr, "oracle: user type larger than %d bytes\n", ORC_MAX_VALUE); abort(); }
    s->fadd = fadd; s->fmult = fmult; s->xsize = xsize; s->ysize = ysize; s->zsize = zsize;
    s->xtype = s->ytype = s->ztype = GBA_UDT; s->add_op = GBA_USER_BINOP; s->mult_op = GBA_USER_BINOP;
}

/* t = A(i,k) (x) B(k,j) */
static inline void sr_mult(const orc_semiring *s, void *t, const void *a, const void *b, int64_t i, int64_t k, int64_t j) {
    if (s->positional) {
        int64_t v;
        switch (s->positional) {
            case GBA_FIRSTI: v = i; break;       case GBA_FIRSTI1: v = i + 1; break;
            case GBA_FIRSTJ: v = k; break;       case GBA_FIRSTJ1: v = k + 1; break;
            case GBA_SECONDJ: v = j; break;      case GBA_SECONDJ1: v = j + 1; break;
            default: v = 0;
        }
        if (s->zsize == 4) *(int32_t *)t = (int32_t)v; else *(int64_t *)t = v;
        return;
    }
    s->fmult(t, a, b);
}

static const uint8_t ZEROS[ORC_MAX_VALUE] = {0};
static inline const void *valp(const void *x, int64_t p, size_t size, bool iso, bool pattern) {
    if (pattern || !x) return ZEROS;
    return (const uint8_t *)x + (iso ? 0 : p) * size;
}

/* ----------------------------------------------------------------- mask -- */
/* GB_MCAST (Source/Template/GB_mcast.h): value of a mask entry as bool */
static bool mask_entry(const gba_matrix *M, int64_t p, bool structural) {
    if (structural) return true;
    size_t ms = orc_type_size(M->type);
    const uint8_t *v = (const uint8_t *)M->x + (M->iso ? 0 : p) * ms;
    for (size_t q = 0; q < ms; q++) if (v[q]) return true;
    return false;
}

/* dense 0/1 image of M(:,j) into m[0..vlen) (1 = mask value true) */
static void mask_scatter(const gba_matrix *M, int64_t j, bool structural, int8_t *m) {
    const int64_t vlen = M->vlen;
    memset(m, 0, (size_t)vlen);
    if (M->sparsity == GBA_SPARSE) {
        for (int64_t p = M->p[j]; p < M->p[j + 1]; p++) if (M->i[p] >= 0 && mask_entry(M, p, structural)) m[M->i[p]] = 1;
    } else {
        for (int64_t i = 0; i < vlen; i++) {
            int64_t p = j * vlen + i;
            if (M->sparsity == GBA_BITMAP && !M->b[p]) continue;
            if (mask_entry(M, p, structural)) m[i] = 1;
        }
    }
}

void orc_free(void *p) { free(p); }
void orc_matrix_free_content(gba_matrix *A) {
    if (!A) return;
    free(A->p); free(A->h); free(A->i); free(A->x); free(A->b);
    A->p = A->h = A->i = NULL; A->x = NULL; A->b = NULL; A->nvals = 0;
}

/* --------------------------------------------------------------- cumsum -- */
/* GB_cumsum (Source/GB_cumsum.c:23): count[0..n) -> exclusive sums, count[n] = total */
void orc_cumsum(int64_t *count, int64_t n) {
    int64_t s = 0;
    for (int64_t k = 0; k < n; k++) { int64_t c = count[k]; count[k] = s; s += c; }
    count[n] = s;
}

/* ------------------------------------------------------------ flopcount -- */
/* GB_AxB_saxpy3_flopcount (Source/GB_AxB_saxpy3_flopcount.c:81-477), as one
 * task (B_ntasks == 1): Bflops[kk] before the cumulative sum is
 *   0                               if B(:,j) is empty (or M(:,j) empty, M not complemented)
 *   (mjnz if M(:,j) sparse and non-empty, else bjnz) + sum over k in B(:,j) of aknz
 * with the binary-search estimate mjnz*(1+4*log2(aknz)) when C<M> and A(:,k)
 * is much longer than M(:,j) (lines 341-355). */
int orc_saxpy3_flopcount(int64_t *Mwork, int64_t *Bflops, const gba_matrix *M, bool Mask_comp,
                         const gba_matrix *A, const gba_matrix *B) {
    if (A->sparsity != GBA_SPARSE || B->sparsity != GBA_SPARSE) return -1;
    const int64_t bnvec = B->nvec;
    const bool mask_is_M = (M != NULL && !Mask_comp);
    const bool M_is_dense = M && (M->sparsity == GBA_BITMAP || M->sparsity == GBA_FULL ||
                                  (double)M->nvals == (double)M->vlen * (double)M->vdim);
    int64_t total_Mwork = 0;
    for (int64_t kk = 0; kk < bnvec; kk++) {
        Bflops[kk] = 0;
        const int64_t bjnz = B->p[kk + 1] - B->p[kk];
        if (bjnz == 0) continue;
        int64_t bjflops = bjnz, mjnz = 0;
        if (M != NULL && !M_is_dense) {
            mjnz = M->p[kk + 1] - M->p[kk];
            if (mjnz == 0 && !Mask_comp) continue;
            if (mjnz > 0) { bjflops = mjnz; total_Mwork += mjnz; }
        }
        const int64_t mjnz_much = 64 * mjnz;
        for (int64_t pB = B->p[kk]; pB < B->p[kk + 1]; pB++) {
            const int64_t k = B->i[pB];
            const int64_t aknz = A->p[k + 1] - A->p[k];
            if (aknz == 0) continue;
            double bkjflops;
            if (mask_is_M && aknz > 256 && mjnz_much < aknz && mjnz < M->vlen && aknz < A->vlen && !A->jumbled)
                bkjflops = mjnz * (1 + 4 * log2((double)aknz));
            else
                bkjflops = (double)aknz;
            bjflops += bkjflops;
        }
        Bflops[kk] = bjflops;
    }
    Bflops[bnvec] = 0;
    orc_cumsum(Bflops, bnvec);
    if (Mwork) *Mwork = total_Mwork;
    return 0;
}

/* --------------------------------------------------------------- saxpy3 -- */
/* C<#M> = A*B, C sparse: Gustavson's method as in the coarse tasks of
 * GB_AxB_saxpy3 (Source/GB_AxB_saxpy3.c:120; phase 1
 * Source/Template/GB_AxB_saxpy3_coarseGus_noM_phase1.c:14-61; phase 5
 * Source/Template/GB_AxB_saxpy3_coarseGus_noM_phase5.c:11-119 and the _M / _notM
 * variants).  Each C(i,j) is summed in increasing (pB, pA) order.  The pattern
 * of C(:,j) is emitted in first-touch order when C(:,j) is "very sparse"
 * (16*cjnz <= cvlen) and in index order otherwise, like the reference; C is
 * flagged jumbled.  Columns are processed independently (OpenMP over j). */
int orc_saxpy3(gba_matrix *C, const gba_matrix *M, bool Mask_comp, bool Mask_struct,
               const gba_matrix *A, const gba_matrix *B, const orc_semiring *s) {
    if (A->sparsity != GBA_SPARSE || B->sparsity != GBA_SPARSE) return -1;
    if (A->vdim != B->vlen) return -2;
    const int64_t cvlen = A->vlen, cvdim = B->vdim;
    const size_t zs = s->zsize;
    int64_t *Cp = (int64_t *)calloc((size_t)cvdim + 1, sizeof(int64_t));
    int nth = 1;
#ifdef _OPENMP
    nth = omp_get_max_threads();
#endif
    /* per-thread workspaces: Hf marks, Hx values, mask image, touch list */
    int64_t **Hf = (int64_t **)calloc((size_t)nth, sizeof(void *));
    uint8_t **Hx = (uint8_t **)calloc((size_t)nth, sizeof(void *));
    int8_t **Hm = (int8_t **)calloc((size_t)nth, sizeof(void *));
    int64_t **Hl = (int64_t **)calloc((size_t)nth, sizeof(void *));
    int64_t *marks = (int64_t *)calloc((size_t)nth, sizeof(int64_t));
    for (int t = 0; t < nth; t++) {
        Hf[t] = (int64_t *)calloc((size_t)cvlen + 1, sizeof(int64_t));
        Hx[t] = (uint8_t *)malloc(((size_t)cvlen + 1) * zs);
        Hm[t] = M ? (int8_t *)malloc((size_t)cvlen + 1) : NULL;
        Hl[t] = (int64_t *)malloc(((size_t)cvlen + 1) * sizeof(int64_t));
    }
    const bool fast_f64 = (s->mult_op == GBA_TIMES && s->add_op == GBA_PLUS && s->ztype == GBA_FP64 && !s->positional && !A->iso && !B->iso);
    /* pass 1 (symbolic) and pass 2 (numeric) share one column routine */
    int64_t *Ci = NULL; uint8_t *Cx = NULL;
    for (int pass = 0; pass < 2; pass++) {
        #pragma omp parallel for schedule(dynamic, 64)
        for (int64_t j = 0; j < cvdim; j++) {
            int t = 0;
#ifdef _OPENMP
            t = omp_get_thread_num();
#endif
            int64_t *hf = Hf[t]; uint8_t *hx = Hx[t]; int8_t *hm = Hm[t]; int64_t *hl = Hl[t];
            if (B->p[j + 1] == B->p[j]) { if (pass == 0) Cp[j] = 0; continue; }
            if (M) mask_scatter(M, j, Mask_struct, hm);
            const int64_t mark = ++marks[t];
            int64_t cjnz = 0;
            for (int64_t pB = B->p[j]; pB < B->p[j + 1]; pB++) {
                const int64_t k = B->i[pB];
                const void *bkj = valp(B->x, pB, s->ysize, B->iso, s->b_pattern);
                for (int64_t pA = A->p[k]; pA < A->p[k + 1]; pA++) {
                    const int64_t i = A->i[pA];
                    if (M && ((hm[i] != 0) == Mask_comp)) continue;
                    if (pass == 0) {
                        if (hf[i] != mark) { hf[i] = mark; cjnz++; }
                    } else if (fast_f64) {
                        const double tt = ((const double *)A->x)[pA] * (*(const double *)bkj);
                        if (hf[i] != mark) { hf[i] = mark; ((double *)hx)[i] = tt; hl[cjnz++] = i; }
                        else ((double *)hx)[i] += tt;
                    } else {
                        uint8_t tt[ORC_MAX_VALUE];
                        sr_mult(s, tt, valp(A->x, pA, s->xsize, A->iso, s->a_pattern), bkj, i, k, j);
                        if (hf[i] != mark) { hf[i] = mark; memcpy(hx + i * zs, tt, zs); hl[cjnz++] = i; }
                        else s->fadd(hx + i * zs, hx + i * zs, tt);
                    }
                }
            }
            if (pass == 0) { Cp[j] = cjnz; continue; }
            int64_t pC = Cp[j];
            if (16 * cjnz > cvlen) {                 /* not very sparse: gather in index order */
                for (int64_t i = 0; i < cvlen; i++) if (hf[i] == mark) { Ci[pC] = i; memcpy(Cx + pC * zs, hx + i * zs, zs); pC++; }
            } else {                                 /* very sparse: first-touch order (jumbled) */
                for (int64_t q = 0; q < cjnz; q++) { const int64_t i = hl[q]; Ci[pC] = i; memcpy(Cx + pC * zs, hx + i * zs, zs); pC++; }
            }
        }
        if (pass == 0) {
            Cp[cvdim] = 0;
            orc_cumsum(Cp, cvdim);
            const int64_t cnz = Cp[cvdim];
            Ci = (int64_t *)malloc((size_t)(cnz > 0 ? cnz : 1) * sizeof(int64_t));
            Cx = (uint8_t *)malloc((size_t)(cnz > 0 ? cnz : 1) * zs);
        }
    }
    for (int t = 0; t < nth; t++) { free(Hf[t]); free(Hx[t]); free(Hm[t]); free(Hl[t]); }
    free(Hf); free(Hx); free(Hm); free(Hl); free(marks);
    memset(C, 0, sizeof(*C));
    C->vlen = cvlen; C->vdim = cvdim; C->nvec = cvdim; C->nvals = Cp[cvdim];
    C->p = Cp; C->i = Ci; C->x = Cx; C->type = s->ztype; C->sparsity = GBA_SPARSE; C->mem = GBA_MEM_HOST;
    C->jumbled = true; C->is_csc = B->is_csc;
    return 0;
}

/* ----------------------------------------------------------------- dot3 -- */
/* one dot product A(:,i)'B(:,j), sparse x sparse, sorted indices:
 * Source/Template/GB_AxB_dot_cij.c:543-679 (no-overlap test, gallop when one
 * list is 8x longer, else two-pointer merge).  Terms are summed in increasing k. */
static int64_t trim_search(const int64_t *X, int64_t lo, int64_t hi, int64_t target) {
    while (lo <= hi) { int64_t mid = (lo + hi) / 2; if (X[mid] < target) lo = mid + 1; else hi = mid - 1; }
    return lo;
}
static bool dot_ss(const orc_semiring *s, const gba_matrix *A, const gba_matrix *B, int64_t i, int64_t j, void *cij) {
    int64_t pA = A->p[i], pA_end = A->p[i + 1], pB = B->p[j], pB_end = B->p[j + 1];
    const int64_t ainz = pA_end - pA, bjnz = pB_end - pB;
    bool found = false;
    uint8_t t[ORC_MAX_VALUE];
    if (ainz == 0 || bjnz == 0) return false;
    if (A->i[pA_end - 1] < B->i[pB] || B->i[pB_end - 1] < A->i[pA]) return false;
    while (pA < pA_end && pB < pB_end) {
        const int64_t ia = A->i[pA], ib = B->i[pB];
        if (ia < ib) { pA = (ainz > 8 * bjnz) ? trim_search(A->i, pA + 1, pA_end - 1, ib) : pA + 1; }
        else if (ib < ia) { pB = (bjnz > 8 * ainz) ? trim_search(B->i, pB + 1, pB_end - 1, ia) : pB + 1; }
        else {
            sr_mult(s, t, valp(A->x, pA, s->xsize, A->iso, s->a_pattern), valp(B->x, pB, s->ysize, B->iso, s->b_pattern), i, ia, j);
            if (found) s->fadd(cij, cij, t); else { memcpy(cij, t, s->zsize); found = true; }
            pA++; pB++;
        }
    }
    return found;
}

/* general dot product for any mix of sparse / bitmap / full operands */
static bool dot_any(const orc_semiring *s, const gba_matrix *A, const gba_matrix *B, int64_t i, int64_t j, void *cij) {
    const bool As = A->sparsity == GBA_SPARSE, Bs = B->sparsity == GBA_SPARSE;
    if (As && Bs) return dot_ss(s, A, B, i, j, cij);
    bool found = false;
    uint8_t t[ORC_MAX_VALUE];
    const int64_t vlen = A->vlen;
    #define TERM(k, pa, pb) do { \
        sr_mult(s, t, valp(A->x, (pa), s->xsize, A->iso, s->a_pattern), valp(B->x, (pb), s->ysize, B->iso, s->b_pattern), i, (k), j); \
        if (found) s->fadd(cij, cij, t); else { memcpy(cij, t, s->zsize); found = true; } } while (0)
    if (As) {
        for (int64_t pA = A->p[i]; pA < A->p[i + 1]; pA++) {
            const int64_t k = A->i[pA], pB = j * vlen + k;
            if (B->sparsity == GBA_BITMAP && !B->b[pB]) continue;
            TERM(k, pA, pB);
        }
    } else if (Bs) {
        for (int64_t pB = B->p[j]; pB < B->p[j + 1]; pB++) {
            const int64_t k = B->i[pB], pA = i * vlen + k;
            if (A->sparsity == GBA_BITMAP && !A->b[pA]) continue;
            TERM(k, pA, pB);
        }
    } else {
        for (int64_t k = 0; k < vlen; k++) {
            const int64_t pA = i * vlen + k, pB = j * vlen + k;
            if (A->sparsity == GBA_BITMAP && !A->b[pA]) continue;
            if (B->sparsity == GBA_BITMAP && !B->b[pB]) continue;
            TERM(k, pA, pB);
        }
    }
    #undef TERM
    return found;
}

/* C<M> = A'*B, M sparse, not complemented: GB_AxB_dot3 (Source/GB_AxB_dot3.c:37),
 * Source/Template/GB_AxB_dot3_template.c:16-168.  C has M's pattern; an entry
 * with no term is a zombie, Ci[p] = GBA_FLIP(i). */
int orc_dot3(gba_matrix *C, const gba_matrix *M, bool Mask_struct, const gba_matrix *A, const gba_matrix *B,
             const orc_semiring *s) {
    if (!M || M->sparsity != GBA_SPARSE) return -1;
    if (A->vlen != B->vlen || M->vlen != A->vdim || M->vdim != B->vdim) return -2;
    const int64_t mnz = M->nvals, nvec = M->nvec;
    const size_t zs = s->zsize;
    int64_t *Cp = (int64_t *)malloc((size_t)(nvec + 1) * sizeof(int64_t));
    int64_t *Ci = (int64_t *)malloc((size_t)(mnz > 0 ? mnz : 1) * sizeof(int64_t));
    uint8_t *Cx = (uint8_t *)calloc((size_t)(mnz > 0 ? mnz : 1), zs);
    memcpy(Cp, M->p, (size_t)(nvec + 1) * sizeof(int64_t));
    int64_t nzombies = 0;
    #pragma omp parallel for schedule(dynamic, 256) reduction(+:nzombies)
    for (int64_t j = 0; j < nvec; j++) {
        for (int64_t pC = M->p[j]; pC < M->p[j + 1]; pC++) {
            const int64_t i = M->i[pC];
            bool exists = false;
            if (mask_entry(M, pC, Mask_struct)) exists = dot_any(s, A, B, i, j, Cx + pC * zs);
            if (exists) Ci[pC] = i; else { Ci[pC] = GBA_FLIP(i); nzombies++; }
        }
    }
    memset(C, 0, sizeof(*C));
    C->vlen = M->vlen; C->vdim = M->vdim; C->nvec = nvec; C->nvals = mnz; C->nzombies = nzombies;
    C->p = Cp; C->i = Ci; C->x = Cx; C->type = s->ztype; C->sparsity = GBA_SPARSE; C->mem = GBA_MEM_HOST;
    C->jumbled = M->jumbled; C->is_csc = M->is_csc;
    return 0;
}

/* may C(i,j) be computed under mask M?  M of any format */
static bool mask_allows(const gba_matrix *M, bool comp, bool structural, int64_t i, int64_t j) {
    if (!M) return true;
    bool mij = false;
    if (M->sparsity == GBA_SPARSE) {
        for (int64_t p = M->p[j]; p < M->p[j + 1]; p++) if (M->i[p] == i) { mij = mask_entry(M, p, structural); break; }
    } else {
        const int64_t p = j * M->vlen + i;
        mij = (M->sparsity == GBA_BITMAP) ? (M->b[p] != 0) : true;
        if (mij) mij = mask_entry(M, p, structural);
    }
    return mij != comp;
}

/* C<#M> = A'*B, C bitmap: GB_AxB_dot2 (Source/GB_AxB_dot2.c:53,
 * Source/Template/GB_AxB_dot2_template.c): every C(i,j) the mask allows is one
 * dot product; Cb(i,j) = 1 iff it has a term. */
int orc_dot2(gba_matrix *C, const gba_matrix *M, bool Mask_comp, bool Mask_struct, const gba_matrix *A,
             const gba_matrix *B, const orc_semiring *s) {
    if (A->vlen != B->vlen) return -2;
    const int64_t cvlen = A->vdim, cvdim = B->vdim, cn = cvlen * cvdim;
    const size_t zs = s->zsize;
    int8_t *Cb = (int8_t *)calloc((size_t)(cn > 0 ? cn : 1), 1);
    uint8_t *Cx = (uint8_t *)calloc((size_t)(cn > 0 ? cn : 1), zs);
    int64_t nvals = 0;
    #pragma omp parallel for schedule(dynamic, 16) reduction(+:nvals)
    for (int64_t j = 0; j < cvdim; j++)
        for (int64_t i = 0; i < cvlen; i++) {
            if (!mask_allows(M, Mask_comp, Mask_struct, i, j)) continue;
            if (dot_any(s, A, B, i, j, Cx + (j * cvlen + i) * zs)) { Cb[j * cvlen + i] = 1; nvals++; }
        }
    memset(C, 0, sizeof(*C));
    C->vlen = cvlen; C->vdim = cvdim; C->nvec = cvdim; C->nvals = nvals; C->b = Cb; C->x = Cx;
    C->type = s->ztype; C->sparsity = GBA_BITMAP; C->mem = GBA_MEM_HOST; C->is_csc = B->is_csc;
    return 0;
}

/* C += A'*B in place, C full: GB_AxB_dot4 (Source/GB_AxB_dot4.c:49,
 * Source/Template/GB_AxB_dot4_template.c:316-711): cij = C(i,j); for k: cij += a*b; */
int orc_dot4(gba_matrix *C, const gba_matrix *A, const gba_matrix *B, const orc_semiring *s) {
    if (C->sparsity != GBA_FULL || A->vlen != B->vlen || C->vlen != A->vdim || C->vdim != B->vdim) return -2;
    const size_t zs = s->zsize;
    const int64_t cvlen = C->vlen;
    #pragma omp parallel for schedule(dynamic, 16)
    for (int64_t j = 0; j < C->vdim; j++)
        for (int64_t i = 0; i < cvlen; i++) {
            uint8_t cij[ORC_MAX_VALUE];
            if (dot_any(s, A, B, i, j, cij)) {
                uint8_t *c = (uint8_t *)C->x + (j * cvlen + i) * zs;
                s->fadd(c, c, cij);
            }
        }
    return 0;
}

/* ------------------------------------------------- saxpy4 / saxpy5 / saxbit -- */
/* C += A*B in place, C full, A sparse, B bitmap/full: GB_AxB_saxpy4
 * (Source/GB_AxB_saxpy4.c:48, Source/Template/GB_AxB_saxpy4_template.c), coarse
 * form: for j, for k with B(k,j) present, for i in A(:,k): C(i,j) += A(i,k)*B(k,j). */
int orc_saxpy4(gba_matrix *C, const gba_matrix *A, const gba_matrix *B, const orc_semiring *s) {
    if (C->sparsity != GBA_FULL || A->sparsity != GBA_SPARSE || A->vdim != B->vlen || C->vlen != A->vlen || C->vdim != B->vdim) return -2;
    const size_t zs = s->zsize;
    #pragma omp parallel for schedule(dynamic, 1)
    for (int64_t j = 0; j < B->vdim; j++)
        for (int64_t k = 0; k < B->vlen; k++) {
            const int64_t pB = j * B->vlen + k;
            if (B->sparsity == GBA_BITMAP && !B->b[pB]) continue;
            const void *bkj = valp(B->x, pB, s->ysize, B->iso, s->b_pattern);
            for (int64_t pA = A->p[k]; pA < A->p[k + 1]; pA++) {
                const int64_t i = A->i[pA];
                uint8_t t[ORC_MAX_VALUE];
                sr_mult(s, t, valp(A->x, pA, s->xsize, A->iso, s->a_pattern), bkj, i, k, j);
                uint8_t *c = (uint8_t *)C->x + (j * C->vlen + i) * zs;
                s->fadd(c, c, t);
            }
        }
    return 0;
}

/* C += A*B in place, C full, A bitmap/full, B sparse: GB_AxB_saxpy5
 * (Source/GB_AxB_saxpy5.c:55, Source/Factories/GB_AxB_saxpy5_meta.c):
 * for j, for k in B(:,j), for all i: C(i,j) += A(i,k)*B(k,j). */
int orc_saxpy5(gba_matrix *C, const gba_matrix *A, const gba_matrix *B, const orc_semiring *s) {
    if (C->sparsity != GBA_FULL || B->sparsity != GBA_SPARSE || A->vdim != B->vlen || C->vlen != A->vlen || C->vdim != B->vdim) return -2;
    const size_t zs = s->zsize;
    #pragma omp parallel for schedule(dynamic, 1)
    for (int64_t j = 0; j < B->vdim; j++)
        for (int64_t pB = B->p[j]; pB < B->p[j + 1]; pB++) {
            const int64_t k = B->i[pB];
            const void *bkj = valp(B->x, pB, s->ysize, B->iso, s->b_pattern);
            for (int64_t i = 0; i < A->vlen; i++) {
                const int64_t pA = k * A->vlen + i;
                if (A->sparsity == GBA_BITMAP && !A->b[pA]) continue;
                uint8_t t[ORC_MAX_VALUE];
                sr_mult(s, t, valp(A->x, pA, s->xsize, A->iso, s->a_pattern), bkj, i, k, j);
                uint8_t *c = (uint8_t *)C->x + (j * C->vlen + i) * zs;
                s->fadd(c, c, t);
            }
        }
    return 0;
}

/* C<#M> = A*B, C bitmap: GB_AxB_saxbit (Source/GB_AxB_saxbit.c:46,
 * Source/Template/GB_AxB_saxbit_template.c).  A sparse or bitmap/full, B
 * bitmap/full.  Cb(i,j) is raised by the first product the mask allows. */
int orc_saxbit(gba_matrix *C, const gba_matrix *M, bool Mask_comp, bool Mask_struct, const gba_matrix *A,
               const gba_matrix *B, const orc_semiring *s) {
    if (A->vdim != B->vlen) return -2;
    if (B->sparsity == GBA_SPARSE) return -1;
    const int64_t cvlen = A->vlen, cvdim = B->vdim, cn = cvlen * cvdim;
    const size_t zs = s->zsize;
    int8_t *Cb = (int8_t *)calloc((size_t)(cn > 0 ? cn : 1), 1);
    uint8_t *Cx = (uint8_t *)calloc((size_t)(cn > 0 ? cn : 1), zs);
    int64_t nvals = 0;
    #pragma omp parallel for schedule(dynamic, 1) reduction(+:nvals)
    for (int64_t j = 0; j < cvdim; j++) {
        int8_t *hm = M ? (int8_t *)malloc((size_t)cvlen + 1) : NULL;
        if (M) mask_scatter(M, j, Mask_struct, hm);
        for (int64_t k = 0; k < B->vlen; k++) {
            const int64_t pB = j * B->vlen + k;
            if (B->sparsity == GBA_BITMAP && !B->b[pB]) continue;
            const void *bkj = valp(B->x, pB, s->ysize, B->iso, s->b_pattern);
            const int64_t a0 = (A->sparsity == GBA_SPARSE) ? A->p[k] : 0;
            const int64_t a1 = (A->sparsity == GBA_SPARSE) ? A->p[k + 1] : cvlen;
            for (int64_t q = a0; q < a1; q++) {
                int64_t i, pA;
                if (A->sparsity == GBA_SPARSE) { pA = q; i = A->i[q]; }
                else { i = q; pA = k * cvlen + i; if (A->sparsity == GBA_BITMAP && !A->b[pA]) continue; }
                if (M && ((hm[i] != 0) == Mask_comp)) continue;
                uint8_t t[ORC_MAX_VALUE];
                sr_mult(s, t, valp(A->x, pA, s->xsize, A->iso, s->a_pattern), bkj, i, k, j);
                uint8_t *c = Cx + (j * cvlen + i) * zs;
                if (Cb[j * cvlen + i]) s->fadd(c, c, t);
                else { memcpy(c, t, zs); Cb[j * cvlen + i] = 1; nvals++; }
            }
        }
        free(hm);
    }
    memset(C, 0, sizeof(*C));
    C->vlen = cvlen; C->vdim = cvdim; C->nvec = cvdim; C->nvals = nvals; C->b = Cb; C->x = Cx;
    C->type = s->ztype; C->sparsity = GBA_BITMAP; C->mem = GBA_MEM_HOST; C->is_csc = B->is_csc;
    return 0;
}

/* ------------------------------------------------------------- spec mxm -- */
/* C<M> = accum (C, A*B) from the definition, as the reference's own test
 * oracle does with dense arrays (Test/GB_spec_mxm.m:99-117 for T = A*B,
 * Test/GB_spec_accum_mask.m for the write-back).  All matrices are addressed
 * by (row, col) whatever their storage; the result is returned as three dense
 * row-major arrays: present[nrows*ncols], value[nrows*ncols*zsize]. */
static bool get_entry(const gba_matrix *A, int64_t row, int64_t col, const void **val, size_t size) {
    const int64_t i = A->is_csc ? row : col, j = A->is_csc ? col : row;   /* i within vector j */
    if (A->sparsity == GBA_SPARSE) {
        for (int64_t p = A->p[j]; p < A->p[j + 1]; p++) if (A->i[p] == i) { *val = (const uint8_t *)A->x + (A->iso ? 0 : p) * size; return true; }
        return false;
    }
    if (A->sparsity == GBA_HYPERSPARSE) {
        /* vector j is present only if it is listed in h: binary search, the lookup the CPU library
           uses when a matrix has no hyper hash (Source/Shared/GB_hyper_hash_lookup.h:16-40) */
        int64_t lo = 0, hi = A->nvec - 1;
        while (lo < hi) { const int64_t mid = (lo + hi) / 2; if (A->h[mid] < j) lo = mid + 1; else hi = mid; }
        if (A->nvec == 0 || A->h[lo] != j) return false;
        for (int64_t p = A->p[lo]; p < A->p[lo + 1]; p++) if (A->i[p] == i) { *val = (const uint8_t *)A->x + (A->iso ? 0 : p) * size; return true; }
        return false;
    }
    const int64_t p = j * A->vlen + i;
    if (A->sparsity == GBA_BITMAP && !A->b[p]) return false;
    *val = A->x ? (const uint8_t *)A->x + (A->iso ? 0 : p) * size : ZEROS;
    return true;
}

/* Typecasting of built-in types (Source/GB_casting.h:65-143, 176-215): C casts, except that a
 * floating-point value becomes an integer by saturation with NaN -> 0, and anything becomes bool by
 * comparison with zero (so NaN -> true). */
void orc_cast(void *z, int ztype, const void *x, int xtype) {
    double d = 0; int64_t si = 0; uint64_t ui = 0; int kind;           /* 0 signed, 1 unsigned, 2 floating */
    if (ztype == GBA_FC32 || ztype == GBA_FC64) {
        /* to complex: the C conversion, real part = the value, imaginary part 0; complex to complex part by part */
        if (xtype == GBA_FC32) { const fc32_t c = *(const fc32_t *)x; if (ztype == GBA_FC32) *(fc32_t *)z = c; else *(fc64_t *)z = CMPLX((double)crealf(c), (double)cimagf(c)); return; }
        if (xtype == GBA_FC64) { const fc64_t c = *(const fc64_t *)x; if (ztype == GBA_FC64) *(fc64_t *)z = c; else *(fc32_t *)z = CMPLXF((float)creal(c), (float)cimag(c)); return; }
        if (ztype == GBA_FC32) { float r; orc_cast(&r, GBA_FP32, x, xtype); *(fc32_t *)z = CMPLXF(r, 0); }
        else { double r; orc_cast(&r, GBA_FP64, x, xtype); *(fc64_t *)z = CMPLX(r, 0); }
        return;
    }
    if (xtype == GBA_FC32 || xtype == GBA_FC64) {
        /* from complex (Source/GB_casting.h:211-446): bool is "either part non-zero", the rest casts the real part */
        const double re = (xtype == GBA_FC32) ? (double)crealf(*(const fc32_t *)x) : creal(*(const fc64_t *)x);
        const double im = (xtype == GBA_FC32) ? (double)cimagf(*(const fc32_t *)x) : cimag(*(const fc64_t *)x);
        if (ztype == GBA_BOOL) { *(uint8_t *)z = (re != 0 || im != 0); return; }
        if (xtype == GBA_FC32) { const float r = crealf(*(const fc32_t *)x); orc_cast(z, ztype, &r, GBA_FP32); }
        else orc_cast(z, ztype, &re, GBA_FP64);
        return;
    }
    switch (xtype) {
        case GBA_BOOL:   ui = *(const uint8_t *)x != 0; kind = 1; break;
        case GBA_INT8:   si = *(const int8_t *)x;   kind = 0; break;
        case GBA_INT16:  si = *(const int16_t *)x;  kind = 0; break;
        case GBA_INT32:  si = *(const int32_t *)x;  kind = 0; break;
        case GBA_INT64:  si = *(const int64_t *)x;  kind = 0; break;
        case GBA_UINT8:  ui = *(const uint8_t *)x;  kind = 1; break;
        case GBA_UINT16: ui = *(const uint16_t *)x; kind = 1; break;
        case GBA_UINT32: ui = *(const uint32_t *)x; kind = 1; break;
        case GBA_UINT64: ui = *(const uint64_t *)x; kind = 1; break;
        case GBA_FP32:   d = (double)*(const float *)x; kind = 2; break;
        default:         d = *(const double *)x;    kind = 2; break;
    }
#define ORC_TO_SIGNED(T, LO, HI) \
    *(T *)z = kind == 2 ? (isnan(d) ? 0 : d <= (double)(LO) ? (LO) : d >= (double)(HI) ? (HI) : (T)d) : kind == 0 ? (T)si : (T)ui; break
#define ORC_TO_UNSIGNED(T, HI) \
    *(T *)z = kind == 2 ? ((isnan(d) || d <= 0) ? 0 : d >= (double)(HI) ? (HI) : (T)d) : kind == 0 ? (T)si : (T)ui; break
    switch (ztype) {
        case GBA_BOOL:   *(uint8_t *)z = kind == 2 ? (d != 0) : kind == 0 ? (si != 0) : (ui != 0); break;
        case GBA_INT8:   ORC_TO_SIGNED(int8_t, INT8_MIN, INT8_MAX);
        case GBA_INT16:  ORC_TO_SIGNED(int16_t, INT16_MIN, INT16_MAX);
        case GBA_INT32:  ORC_TO_SIGNED(int32_t, INT32_MIN, INT32_MAX);
        case GBA_INT64:  ORC_TO_SIGNED(int64_t, INT64_MIN, INT64_MAX);
        case GBA_UINT8:  ORC_TO_UNSIGNED(uint8_t, UINT8_MAX);
        case GBA_UINT16: ORC_TO_UNSIGNED(uint16_t, UINT16_MAX);
        case GBA_UINT32: ORC_TO_UNSIGNED(uint32_t, UINT32_MAX);
        case GBA_UINT64: ORC_TO_UNSIGNED(uint64_t, UINT64_MAX);
        case GBA_FP32:   *(float *)z = kind == 2 ? (xtype == GBA_FP32 ? *(const float *)x : (float)d) : kind == 0 ? (float)si : (float)ui; break;
        default:         *(double *)z = kind == 2 ? d : kind == 0 ? (double)si : (double)ui; break;
    }
#undef ORC_TO_SIGNED
#undef ORC_TO_UNSIGNED
}

int orc_mxm_spec(int8_t *present, void *value, int64_t nrows, int64_t ncols, int64_t inner,
                 const gba_matrix *C, const gba_matrix *M, const gba_binop *accum, const gba_semiring *sr_in,
                 const gba_matrix *A, bool A_tran, const gba_matrix *B, bool B_tran, bool flipxy,
                 bool C_replace, bool Mask_comp, bool Mask_struct) {
    orc_semiring s;
    if (orc_semiring_builtin(&s, sr_in, false) != 0) return -1;
    /* value[] has the type of C.  The accumulator works in its own type: where C and T both have an entry, C's is
       cast to it, T's is cast to it from the monoid's type, and the result is cast to C's type; an entry of T
       alone is cast to C's type, an entry of C alone stays as it is (Test/GB_spec_accum.m:45-63).  The operands
       are cast to the multiplier's input types (Source/GB_mxm.c:84-99) */
    const size_t zs = orc_type_size(C->type);
    const bool cast_t = (C->type != s.ztype);
    const bool cast_a = !s.positional && A->type != s.xtype, cast_b = !s.positional && B->type != s.xtype;
    if (cast_t && s.ztype == GBA_UDT) return -1;
    binfn facc = NULL;
    if (accum && accum->op != GBA_NOP) {
        if (accum->type != C->type && (accum->type >= GBA_FC32 || C->type >= GBA_FC32 || s.ztype >= GBA_FC32)) return -1;
        facc = lookup_binop(accum->type == GBA_BOOL ? bool_rename(accum->op) : accum->op, accum->type); if (!facc) return -1;
    }
    #pragma omp parallel for schedule(dynamic, 4)
    for (int64_t r = 0; r < nrows; r++)
        for (int64_t c = 0; c < ncols; c++) {
            /* T(r,c) = sum_k A(r,k)*B(k,c), terms in increasing k */
            uint8_t t[ORC_MAX_VALUE], tk[ORC_MAX_VALUE], ca[ORC_MAX_VALUE], cb[ORC_MAX_VALUE]; bool t_in = false;
            for (int64_t k = 0; k < inner; k++) {
                const void *a, *b;
                if (!get_entry(A, A_tran ? k : r, A_tran ? r : k, &a, orc_type_size(A->type))) continue;
                if (!get_entry(B, B_tran ? c : k, B_tran ? k : c, &b, orc_type_size(B->type))) continue;
                if (cast_a) { orc_cast(ca, s.xtype, a, A->type); a = ca; }
                if (cast_b) { orc_cast(cb, s.xtype, b, B->type); b = cb; }
                if (s.positional) {
                    /* positional ops refer to the user's operands: i = row of A entry, etc. */
                    int64_t v; int op = sr_in->mult;
                    if (flipxy) op = flip_op(op);
                    switch (op) {
                        case GBA_FIRSTI: v = r; break;   case GBA_FIRSTI1: v = r + 1; break;
                        case GBA_FIRSTJ: case GBA_SECONDI: v = k; break;
                        case GBA_FIRSTJ1: case GBA_SECONDI1: v = k + 1; break;
                        case GBA_SECONDJ: v = c; break;  case GBA_SECONDJ1: v = c + 1; break;
                        default: v = 0;
                    }
                    if (s.zsize == 4) *(int32_t *)tk = (int32_t)v; else *(int64_t *)tk = v;
                } else if (flipxy) s.fmult(tk, b, a);
                else s.fmult(tk, a, b);
                if (t_in) s.fadd(t, t, tk); else { memcpy(t, tk, s.zsize); t_in = true; }
            }
            uint8_t traw[ORC_MAX_VALUE];                                     /* T(r,c) in the monoid's type */
            if (t_in) memcpy(traw, t, s.zsize);
            if (t_in && cast_t) { uint8_t tc[ORC_MAX_VALUE]; orc_cast(tc, C->type, t, s.ztype); memcpy(t, tc, zs); }
            /* Z = accum (C,T); C<M,replace> = Z */
            const void *cv; const bool c_in = get_entry(C, r, c, &cv, zs);
            bool m = true;
            if (M) {
                const void *mv; const size_t ms = orc_type_size(M->type);
                m = get_entry(M, r, c, &mv, ms);
                if (m && !Mask_struct) { bool nz = false; for (size_t q = 0; q < ms; q++) nz |= (((const uint8_t *)mv)[q] != 0); m = nz; }
            }
            if (Mask_comp) m = !m;
            uint8_t z[ORC_MAX_VALUE]; bool z_in;
            if (facc) {
                if (c_in && t_in) {
                    if (accum->type == C->type) facc(z, cv, t);
                    else {
                        uint8_t cu[ORC_MAX_VALUE], tu[ORC_MAX_VALUE], zu[ORC_MAX_VALUE];
                        orc_cast(cu, accum->type, cv, C->type);
                        orc_cast(tu, accum->type, traw, s.ztype);
                        facc(zu, cu, tu);
                        orc_cast(z, C->type, zu, accum->type);
                    }
                    z_in = true;
                }
                else if (c_in) { memcpy(z, cv, zs); z_in = true; }
                else if (t_in) { memcpy(z, t, zs); z_in = true; }
                else z_in = false;
            } else { z_in = t_in; if (t_in) memcpy(z, t, zs); }
            const int64_t o = r * ncols + c;
            if (m) { present[o] = z_in; if (z_in) memcpy((uint8_t *)value + o * zs, z, zs); }
            else if (C_replace) present[o] = 0;
            else { present[o] = c_in; if (c_in) memcpy((uint8_t *)value + o * zs, cv, zs); }
        }
    return 0;
}

/* ------------------------------------------------------- sort / zombies -- */
/* what GrB_wait does to a jumbled result with zombies (Source/GB_wait.c,
 * Source/GB_unjumble.c): drop zombies, sort every vector by index */
typedef struct { int64_t i; int64_t p; } ip_pair;
static int cmp_ip(const void *a, const void *b) { int64_t x = ((const ip_pair *)a)->i, y = ((const ip_pair *)b)->i; return (x > y) - (x < y); }
int orc_matrix_wait(gba_matrix *C) {
    if (C->sparsity != GBA_SPARSE) return 0;
    const size_t zs = orc_type_size(C->type);
    int64_t *Np = (int64_t *)calloc((size_t)C->nvec + 1, sizeof(int64_t));
    for (int64_t j = 0; j < C->nvec; j++) for (int64_t p = C->p[j]; p < C->p[j + 1]; p++) Np[j] += (C->i[p] >= 0);
    orc_cumsum(Np, C->nvec);
    const int64_t nnz = Np[C->nvec];
    int64_t *Ni = (int64_t *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(int64_t));
    uint8_t *Nx = (uint8_t *)malloc((size_t)(nnz > 0 ? nnz : 1) * (zs ? zs : 1));
    #pragma omp parallel for schedule(dynamic, 256)
    for (int64_t j = 0; j < C->nvec; j++) {
        const int64_t n = Np[j + 1] - Np[j];
        if (n == 0) continue;
        ip_pair *w = (ip_pair *)malloc((size_t)n * sizeof(ip_pair));
        int64_t q = 0;
        for (int64_t p = C->p[j]; p < C->p[j + 1]; p++) if (C->i[p] >= 0) { w[q].i = C->i[p]; w[q].p = p; q++; }
        qsort(w, (size_t)n, sizeof(ip_pair), cmp_ip);
        for (q = 0; q < n; q++) { Ni[Np[j] + q] = w[q].i; memcpy(Nx + (Np[j] + q) * zs, (uint8_t *)C->x + (C->iso ? 0 : w[q].p) * zs, zs); }
        free(w);
    }
    free(C->p); free(C->i); free(C->x);
    C->p = Np; C->i = Ni; C->x = Nx; C->nvals = nnz; C->nzombies = 0; C->jumbled = false; C->iso = false;
    return 0;
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
