/*
 * graphblas_amd.h -- C ABI of libgraphblas_amd.so, the MI355X (gfx950) back end
 * for the sparse matrix-multiply hot path of SuiteSparse:GraphBLAS.
 *
 * Everything here is plain C: pointers, sizes and enums.  No opaque reference
 * structs cross this boundary.  A matrix is described by a POD view
 * (gba_matrix) of the same arrays the reference keeps inside
 * GB_Matrix_opaque (Source/Shared/GB_matrix.h:213-234) and hands out through
 * GxB_Matrix_unpack_CSR/CSC/BitmapR/FullR (Include/GraphBLAS.h:9869-9881):
 * p, h, i, x, b with int64_t indices and int8_t bitmaps.
 *
 * Numeric codes are the reference's own so that a binding can pass them
 * through unchanged:
 *   gba_info      == GrB_Info            (Include/GraphBLAS.h:318-351)
 *   gba_type      == GB_Type_code        (Source/Shared/GB_opaque.h:25-44)
 *   gba_opcode    == GB_Opcode, binary   (Source/Shared/GB_opaque.h:209-284)
 *   sparsity      == GxB_HYPERSPARSE/SPARSE/BITMAP/FULL (Include/GraphBLAS.h:3822-3825)
 *   desc values   == GrB_Desc_Value      (Include/GraphBLAS.h:512-538)
 *
 * Memory: every array named by a gba_matrix lives in device (HBM) memory that
 * was obtained from gba_malloc (the allocator a host library would install
 * through GxB_init, Source/GxB_init.c:54-93), unless mem == GBA_MEM_HOST, in
 * which case the entry points stage it over PCIe first.  Output arrays are
 * allocated by the library with gba_malloc and owned by the caller afterwards
 * (the ownership rule of GxB_Matrix_unpack_*, Include/GraphBLAS.h:9298-9303).
 */
#ifndef GRAPHBLAS_AMD_H
#define GRAPHBLAS_AMD_H

#include <stdint.h>
#include <stddef.h>
#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- codes -- */

typedef int32_t gba_info;
enum {
    GBA_SUCCESS = 0,
    GBA_NO_VALUE = 1,               /* "method declined, try the next one" */
    GBA_UNINITIALIZED_OBJECT = -1,
    GBA_NULL_POINTER = -2,
    GBA_INVALID_VALUE = -3,
    GBA_INVALID_INDEX = -4,
    GBA_DOMAIN_MISMATCH = -5,
    GBA_DIMENSION_MISMATCH = -6,
    GBA_OUTPUT_NOT_EMPTY = -7,
    GBA_NOT_IMPLEMENTED = -8,       /* valid GraphBLAS, not built for device: caller keeps the CPU path */
    GBA_PANIC = -101,
    GBA_OUT_OF_MEMORY = -102,
    GBA_INSUFFICIENT_SPACE = -103,
    GBA_INVALID_OBJECT = -104,
    GBA_INDEX_OUT_OF_BOUNDS = -105,
    GBA_EMPTY_OBJECT = -106
};

typedef int32_t gba_type;
enum {
    GBA_BOOL = 1, GBA_INT8 = 2, GBA_UINT8 = 3, GBA_INT16 = 4, GBA_UINT16 = 5,
    GBA_INT32 = 6, GBA_UINT32 = 7, GBA_INT64 = 8, GBA_UINT64 = 9,
    GBA_FP32 = 10, GBA_FP64 = 11, GBA_FC32 = 12, GBA_FC64 = 13, GBA_UDT = 14
};

typedef int32_t gba_opcode;
enum {
    GBA_NOP = 0,
    GBA_FIRST = 72, GBA_SECOND = 73, GBA_ANY = 74, GBA_PAIR = 75,
    GBA_MIN = 76, GBA_MAX = 77, GBA_PLUS = 78, GBA_MINUS = 79, GBA_RMINUS = 80,
    GBA_TIMES = 81, GBA_DIV = 82, GBA_RDIV = 83, GBA_POW = 84,
    GBA_ISEQ = 85, GBA_ISNE = 86, GBA_ISGT = 87, GBA_ISLT = 88, GBA_ISGE = 89, GBA_ISLE = 90,
    GBA_LOR = 91, GBA_LAND = 92, GBA_LXOR = 93,
    GBA_BOR = 94, GBA_BAND = 95, GBA_BXOR = 96, GBA_BXNOR = 97,
    GBA_EQ = 102, GBA_NE = 103, GBA_GT = 104, GBA_LT = 105, GBA_GE = 106, GBA_LE = 107,
    GBA_FIRSTI = 115, GBA_FIRSTI1 = 116, GBA_FIRSTJ = 117, GBA_FIRSTJ1 = 118,
    GBA_SECONDI = 119, GBA_SECONDI1 = 120, GBA_SECONDJ = 121, GBA_SECONDJ1 = 122,
    GBA_USER_BINOP = 123
};

enum { GBA_HYPERSPARSE = 1, GBA_SPARSE = 2, GBA_BITMAP = 4, GBA_FULL = 8 };

enum { GBA_MEM_DEVICE = 0, GBA_MEM_HOST = 1,
       GBA_MEM_HOST_VISIBLE = 2   /* host memory the device can address: gba_host_malloc, hipHostMalloc */ };

/* descriptor values (GrB_Desc_Value) */
enum {
    GBA_DEFAULT = 0, GBA_REPLACE = 1, GBA_COMP = 2, GBA_TRAN = 3, GBA_STRUCTURE = 4,
    GBA_COMP_STRUCTURE = 6,
    GBA_AxB_GUSTAVSON = 7081, GBA_AxB_DOT = 7083, GBA_AxB_HASH = 7084, GBA_AxB_SAXPY = 7085
};

/* which kernel family handled a call; the analogue of the burble breadcrumbs
 * "saxpy3", "dot3", "dot4", "saxpy4", ... (Source/GB_AxB_meta.c:619-838) */
enum {
    GBA_METHOD_NONE = 0,
    GBA_METHOD_SAXPY3 = 1, GBA_METHOD_SAXBIT = 2, GBA_METHOD_SAXPY4 = 3, GBA_METHOD_SAXPY5 = 4,
    GBA_METHOD_DOT2 = 5, GBA_METHOD_DOT3 = 6, GBA_METHOD_DOT4 = 7,
    GBA_METHOD_ROWSCALE = 8, GBA_METHOD_COLSCALE = 9, GBA_METHOD_ISO_FULL = 10
};

/* ---------------------------------------------------------------- types -- */

/* POD view of one matrix or vector (a vector is an n-by-1 CSC matrix, as in
 * Source/Shared/GB_matrix.h:18-25).  Field meaning follows GB_Matrix_opaque:
 *   sparse:  p[nvec+1], i[nvals], x[nvals]         (nvec == vdim, h == NULL)
 *   hyper:   h[nvec] lists the non-empty vectors, p[nvec+1]
 *   bitmap:  b[vlen*vdim] (0/1), x[vlen*vdim], nvals = number of ones
 *   full:    x[vlen*vdim]
 * iso: x holds ONE value shared by all entries.
 * A zombie is an entry whose index i was replaced by GBA_FLIP(i). */
typedef struct gba_matrix {
    int64_t  vlen;       /* length of each vector  (nrows if is_csc, ncols if by row) */
    int64_t  vdim;       /* number of vectors      (ncols if is_csc, nrows if by row) */
    int64_t  nvec;       /* vectors present in p/h (== vdim unless hypersparse)       */
    int64_t  nvals;      /* entries held (including zombies)                          */
    int64_t *p;
    int64_t *h;
    int64_t *i;
    void    *x;
    int8_t  *b;
    int64_t  nzombies;
    gba_type type;
    int32_t  sparsity;   /* GBA_HYPERSPARSE | GBA_SPARSE | GBA_BITMAP | GBA_FULL */
    int32_t  mem;        /* GBA_MEM_DEVICE | GBA_MEM_HOST | GBA_MEM_HOST_VISIBLE */
    bool     is_csc;
    bool     iso;
    bool     jumbled;    /* indices within a vector may be out of order */
    float    hyper_switch;  /* of an OUTPUT of gba_mxm / gba_mxv / gba_vxm: when > 0, a sparse or hypersparse result is
                               re-conformed as the reference does at the end of GB_mxm (Source/GB_conform_hyper.c): to
                               hypersparse when vdim > 1 and its non-empty vectors are at most hyper_switch * vdim
                               (GB_convert_sparse_to_hyper_test.c), back to sparse when vdim <= 1 or they are more than
                               2 * hyper_switch * vdim (GB_convert_hyper_to_sparse_test.c).  The reference's default is
                               0.0625 (GxB_HYPER_DEFAULT).  0 (a zeroed struct) or negative: the result keeps the form it
                               came in with (sparse stays sparse, hypersparse stays hypersparse).  Ignored for inputs. */
} gba_matrix;

#define GBA_FLIP(i) (-(i) - 2)          /* Source/Template/GB_zombie.h */

/* built-in semiring: z = add-reduce over k of mult(x = A(i,k), y = B(k,j)).
 * gba_mxm / gba_mxv / gba_vxm take built-in operators only (real and complex); user-defined types and
 * operators (GBA_USER_BINOP) go through gba_mxm_udt, which compiles them with hipRTC. */
typedef struct gba_semiring {
    gba_opcode add;      /* monoid operator; its type is ztype            */
    gba_opcode mult;
    gba_type   xtype;    /* mult->xtype; ignored when A is pattern-only   */
    gba_type   ytype;
    gba_type   ztype;
} gba_semiring;

/* binary accumulator (optional): GBA_NOP means "no accum" (NULL in GrB_mxm) */
typedef struct gba_binop {
    gba_opcode op;
    gba_type   type;     /* x, y and z types of the accumulator */
} gba_binop;

/* decoded GrB_Descriptor (Source/Shared/GB_opaque.h:444-465) */
typedef struct gba_descriptor {
    int32_t out;         /* GBA_DEFAULT | GBA_REPLACE                               */
    int32_t mask;        /* GBA_DEFAULT | GBA_COMP | GBA_STRUCTURE | GBA_COMP_STRUCTURE */
    int32_t in0;         /* GBA_DEFAULT | GBA_TRAN                                  */
    int32_t in1;
    int32_t axb;         /* GBA_DEFAULT | GBA_AxB_GUSTAVSON | _DOT | _HASH | _SAXPY */
    int32_t do_sort;     /* nonzero: return C with sorted vectors (GxB_SORT)        */
} gba_descriptor;

/* what a call did, for tests and logs */
typedef struct gba_report {
    int32_t method;         /* GBA_METHOD_*                                     */
    int32_t swapped;        /* swap rule applied: computed (B'A')' instead      */
    int32_t mask_applied;   /* the kernel applied M itself                      */
    int32_t done_in_place;  /* C was updated in place (dot4 / saxpy4 / saxpy5)  */
    int32_t a_transposed;   /* an explicit transpose of A was materialised      */
    int32_t b_transposed;
    int64_t flops;          /* multiply-adds counted by the flop count, or -1   */
    double  kernel_ms;      /* device time of the whole method (HIP events on the library stream) */
    /* saxpy3 only: device time per phase, HIP events on the library stream.
     * [0] flop count + binning  [1] symbolic  [2] numeric binning + cumsum + allocation
     * [3] numeric wave bin  [4] numeric block bin 1  [5] numeric block bin 2  [6] numeric heavy-vector (panel) bin  [7] sort */
    double  phase_ms[8];
    int64_t bin_vectors[4]; /* saxpy3 numeric: vectors routed to wave / block1 / block2 / global bins */
    int64_t bin_cnz[4];     /* entries of C those vectors hold                                        */
    int64_t bin_bnz[4];     /* entries of B those vectors hold                                        */
    int64_t bin_madds[4];   /* multiply-adds those vectors take                                       */
    int64_t panels;         /* saxpy3 numeric, heavy vectors: LDS panels swept, and panels redone at   */
    int64_t panels_redone;  /* half width after a table overflow                                      */
    int64_t cnz;            /* entries produced (saxpy3, dot3) */
    int32_t typecast;       /* bit 0: A's values were cast to the multiplier's type, bit 1: B's, bit 2: T to C's type */
    int32_t index_bytes;    /* bytes per index the entry-tiled SpMV kernel streamed: 8, or 4 from the remembered copy; 0 otherwise */
} gba_report;

/* ------------------------------------------------------ library + memory -- */

/* Bind the calling thread's work to one GPU and create its stream.  Replaces
 * GB_cuda_init behind GrB_init(GxB_*_GPU) (Source/GB_init.c:151-162). */
gba_info gba_init(int device);
gba_info gba_finalize(void);
const char *gba_last_error(void);          /* message of the last failing call; GrB_error analogue (Source/GB_where.h:40-86) */
const char *gba_version(void);

/* Device allocator: what GxB_init(mode, malloc, calloc, realloc, free) would be
 * given in GPU mode (Source/GB_init.c:84-95). */
void    *gba_malloc(size_t nbytes);
void     gba_free(void *p);
/* gba_free keeps blocks for reuse by later gba_malloc calls (hipFree of a
 * multi-gigabyte block stalls for hundreds of ms); the reference's CUDA build
 * gets the same from its RMM pool (rmm_wrap/rmm_wrap.h:53-83).  The pool holds
 * at most `limit` idle bytes (default: 5/8 of the device memory, or
 * GBA_POOL_LIMIT_MB), and is emptied before an allocation is reported failed. */
gba_info gba_pool_set_limit(size_t nbytes);
gba_info gba_pool_release(void);           /* hipFree every idle block now */
gba_info gba_pool_stats(size_t *idle_bytes, size_t *live_bytes);
/* Host allocator whose blocks the device can address (pinned, mapped): the four functions GxB_init (mode, malloc,
 * calloc, realloc, free) takes (Source/GxB_init.c:54-93; GPU mode installs rmm_wrap's, Source/GB_init.c:84-95).
 * Arrays that come out of GxB_*_unpack_* are then views with mem = GBA_MEM_HOST_VISIBLE: the operands of
 * gba_mxv / gba_vxm are read in place (no copy at all), those of gba_mxm are copied at the pinned-memory rate,
 * and a result copied back into such a block needs no staging buffer.  Usable before gba_init; thread-safe.
 * gba_is_device_visible (p): 1 when p lies in memory the device can address (any of these blocks, hipHostMalloc'd
 * or registered memory, device memory), else 0. */
void    *gba_host_malloc(size_t nbytes);
void    *gba_host_calloc(size_t count, size_t size);
void    *gba_host_realloc(void *p, size_t nbytes);
void     gba_host_free(void *p);
int      gba_is_device_visible(const void *p);
gba_info gba_memcpy_h2d(void *dst, const void *src, size_t nbytes);
gba_info gba_memcpy_d2h(void *dst, const void *src, size_t nbytes);
gba_info gba_memset(void *dst, int byte, size_t nbytes);
gba_info gba_sync(void);
void    *gba_stream(void);                 /* hipStream_t used by every kernel of this library */
/* HIP-event stopwatch on that stream: device time between the two calls, in ms */
gba_info gba_timer_start(void);
gba_info gba_timer_stop(double *ms);

/* The library remembers auxiliaries of the PATTERN of the last few large sparse matrices it multiplied by (a 32-bit copy
 * of the indices, the tile tables of the entry-tiled SpMV kernels), as the reference keeps A->Y with its matrix
 * (Source/Shared/GB_matrix.h:240-306).  They are built when a pattern is seen for the second time and are keyed by
 * the addresses of p and i, nvals, nvec and vlen; values are never remembered.  Arrays from gba_malloc are forgotten
 * when they are freed.  A caller that REWRITES p or i of its own memory in place calls gba_matrix_forget (A) first.
 * GBA_PATTERN_CACHE=0 turns the memory off. */
gba_info gba_matrix_forget(const gba_matrix *A);

/* Free the arrays a call allocated inside *A (p, h, i, x, b) and clear them.
 * GB_phybix_free analogue (Source/GB_phybix_free.c). */
void     gba_matrix_free_content(gba_matrix *A);

/* ------------------------------------------------------------- front end -- */

/* C<M> = accum (C, A*B).  Same argument order and meaning as
 * GrB_mxm (Include/GraphBLAS.h:4640-4649, Source/GrB_mxm.c:18); M may be NULL,
 * accum NULL or op == GBA_NOP, desc NULL.  C is in/out: on success its arrays
 * may have been replaced (old ones freed with gba_free unless aliased with an
 * input) and its sparsity/jumbled/nzombies fields describe the result, exactly
 * as GB_mxm leaves its C (Source/GB_mxm.c:149-221).
 * Operands and C of another built-in real type than the semiring's are cast
 * (Source/GB_casting.h rules); hypersparse A, B, M and C are accepted (the
 * inner dimension may have any size: the vectors a hypersparse operand lists
 * are used as they are, Source/Shared/GB_hyper_hash_lookup.h:16; an outer
 * dimension of more than 2^31 vectors is taken when C, the mask and the
 * operand that shares it are all hypersparse); a C
 * that is hypersparse on entry is hypersparse on return, any other sparse
 * result stays sparse, unless C->hyper_switch asks for the reference's rule
 * (see gba_matrix).  A C that is iso-valued on entry is accepted (its value
 * is written out before the update).  A product that replaces C and is
 * iso-valued by the rule of Source/GB_AxB_iso.c comes back with iso = true
 * and one value in x.
 * An accumulator of another built-in real type than C works in its own type
 * (Test/GB_spec_accum.m:45-63, GB_add inside Source/GB_accum_mask.c): where C
 * and the product both have an entry, both are cast to it and the result to
 * C's type; an entry of the product alone is cast to C's type, an entry of C
 * alone is kept bit for bit (up to 2^32 - 2 entries each in C and the product).
 * A sparse mask far costlier than the product is left out of the multiply and
 * applied in the write-back (Source/GB_AxB_saxpy3_slice_balanced.c:309-403);
 * report->mask_applied tells which happened.
 * The built-in complex semirings -- (PLUS, TIMES, ANY) x (FIRST, SECOND, PAIR,
 * PLUS, MINUS, RMINUS, TIMES, DIV, RDIV) over GBA_FC32 / GBA_FC64,
 * Source/Factories/GB_AxB_factory.c:46-243 -- are taken with A, B and C of
 * any built-in type (operands are cast to the semiring's type, the product to
 * C's type: the real part, Source/GB_casting.h:211-446; a real product goes
 * into a complex C the same way); the accumulator has C's type; they run on the road of
 * gba_mxm_udt (operators compiled by hipRTC, one dot product per entry) and
 * share its limits.
 * Returns GBA_NOT_IMPLEMENTED for valid calls outside the device path
 * (user-defined ops -- see gba_mxm_udt --, a complex accumulator on a real C, ...):
 * the caller keeps the CPU library for those, like the GB_cuda_*_branch test
 * in Source/GB_AxB_dot.c:189-197. */
gba_info gba_mxm(gba_matrix *C, const gba_matrix *M, const gba_binop *accum,
                 const gba_semiring *semiring, const gba_matrix *A, const gba_matrix *B,
                 const gba_descriptor *desc, gba_report *report);

/* User-defined type and operators, given as C source exactly as to GxB_Type_new and GxB_BinaryOp_new
 * (Include/GraphBLAS.h, "GxB_Type_new (&type, sizeof (ctype), name, defn)" and "GxB_BinaryOp_new (&op, f,
 * ztype, xtype, ytype, name, defn)"; the reference's JIT compiles the same strings with the C compiler,
 * Source/GB_jitifyer.c).  hipRTC compiles them for the current device; code objects are cached per source.
 * One type serves x, y and z (the shape of Demo/Program/gauss_demo.c:300-360). */
typedef struct {
    const char *type_name;      /* "gauss" */
    const char *type_defn;      /* "typedef struct { int32_t real ; int32_t imag ; } gauss ;" */
    size_t      type_size;      /* sizeof (gauss) */
    const char *add_name;       /* "addgauss" */
    const char *add_defn;       /* "void addgauss (gauss *z, const gauss *x, const gauss *y) { ... }" */
    const char *mult_name;
    const char *mult_defn;
} gba_user_semiring;

/* C<M> = accum (C, A*B) over a user-defined semiring: GrB_mxm for matrices of type GBA_UDT
 * (Source/GB_mxm.c:24).  accum_name/accum_defn NULL = no accumulator.  C may be the same handle as A or
 * B (A = A*A), not a copy of it.  Taken: every format and orientation of A, B and M (hypersparse
 * included); every combination of accum, mask, complement and replace on a
 * sparse, bitmap or full C, with or without entries (at most 2^32 - 2 entries in C and in A*B when C holds
 * entries that the mask or the accumulator keeps).  Declined with GBA_NOT_IMPLEMENTED: an iso-valued or
 * zombie-holding C that holds entries. */
gba_info gba_mxm_udt(gba_matrix *C, const gba_matrix *M, const char *accum_name, const char *accum_defn,
                     const gba_user_semiring *semiring, const gba_matrix *A, const gba_matrix *B,
                     const gba_descriptor *desc, gba_report *report);

/* How many user-defined or complex semirings this process compiled with hipRTC, and how many it loaded as code
 * objects from the on-disk cache instead ($GBA_JIT_CACHE, else $HOME/.cache/graphblas_amd; GBA_JIT_CACHE empty or 0
 * turns the disk cache off).  The reference keeps its JIT kernels on disk the same way (Source/GB_jitifyer.c). */
gba_info gba_jit_stats(uint64_t *compiled, uint64_t *loaded_from_disk);

/* w'<mask'> = accum (w', u'*A) over a user-defined semiring: GrB_vxm for the type GBA_UDT (Source/GrB_vxm.c:63-72).
 * Computed as w = A'*u with the operands of the multiplier exchanged, z = mult (y, x): a user-defined multiplier
 * is not rewritten, the reference passes its operands in the other order as well
 * (Source/GB_flip_binop.c:355-375).  desc->in1 is the transpose setting of A, desc->in0 is ignored, as in GrB_vxm.
 * Otherwise as gba_mxm_udt. */
gba_info gba_vxm_udt(gba_matrix *w, const gba_matrix *mask, const char *accum_name, const char *accum_defn,
                     const gba_user_semiring *semiring, const gba_matrix *u, const gba_matrix *A,
                     const gba_descriptor *desc, gba_report *report);

/* w<mask> = accum (w, A*u)   -- GrB_mxv (Include/GraphBLAS.h:4662-4671, Source/GrB_mxv.c:18) */
gba_info gba_mxv(gba_matrix *w, const gba_matrix *mask, const gba_binop *accum,
                 const gba_semiring *semiring, const gba_matrix *A, const gba_matrix *u,
                 const gba_descriptor *desc, gba_report *report);

/* w'<mask'> = accum (w', u'*A) -- GrB_vxm (Include/GraphBLAS.h:4651-4660, Source/GrB_vxm.c:19) */
gba_info gba_vxm(gba_matrix *w, const gba_matrix *mask, const gba_binop *accum,
                 const gba_semiring *semiring, const gba_matrix *u, const gba_matrix *A,
                 const gba_descriptor *desc, gba_report *report);

/* Method selection only (no compute): the decision GB_AxB_meta +
 * GB_AxB_meta_adotb_control + GB_AxB_saxpy_sparsity + GB_AxB_saxpy would take
 * for this call (Source/GB_AxB_meta.c:41, GB_AxB_meta_adotb_control.c:12,
 * GB_AxB_saxpy_sparsity.c:21, GB_AxB_saxpy.c:16).  Only dimensions, formats,
 * nvals and flags of the views are read (host-side scalars, no device access
 * except the diagonal test). */
gba_info gba_mxm_select(const gba_matrix *C, const gba_matrix *M, const gba_binop *accum,
                        const gba_semiring *semiring, const gba_matrix *A, const gba_matrix *B,
                        const gba_descriptor *desc, gba_report *report);

/* --------------------------------------------------- per-method entry points --
 * The seam the reference itself drew for a device back end: one "can you take
 * it" + "do it" pair per method (GB_cuda_AxB_dot3_branch / GB_cuda_AxB_dot3,
 * Source/GB_cuda_gateway.h:137-156).  All operands are CSC-agnostic "vector"
 * matrices exactly as the CPU drivers see them after GB_AxB_meta folded the
 * CSR/CSC formats; flipxy has already been removed by flipping the multiplier
 * (Source/GB_AxB_meta.c:456-468).  C is an empty view that the callee fills
 * (C->type must be set to the semiring's ztype). */

/* C<#M> = A*B, C sparse: GB_AxB_saxpy3 (Source/GB_AxB_saxpy3.c:120).
 * A, B sparse (other formats are brought to sparse form by gba_mxm before it calls
 * this); M NULL | sparse (sorted) | bitmap | full.  The mask is always applied inside
 * the kernels (*mask_applied = true whenever M is given): the reference's policy of
 * dropping a costly sparse mask and applying it afterwards
 * (Source/GB_AxB_saxpy3_slice_balanced.c:377-393) is not reproduced, so GBA_NO_VALUE
 * is never returned, and axb_method (GUSTAVSON / HASH hints) does not change the
 * kernels -- every vector is routed by its own flop count.  The kernels keep indices in 32 bits; operands with
 * vlen >= 2^31 - 1 (Source/GB_AxB_saxpy3.c:120 has no limit) are multiplied over the ranks of the distinct indices
 * that A holds (fewer than 2^31 - 1 of them, in fewer than 2^31 entries) and the result is renamed back; such a
 * mask must be sparse. */
gba_info gba_AxB_saxpy3(gba_matrix *C, const gba_matrix *M, bool Mask_comp, bool Mask_struct,
                        const gba_matrix *A, const gba_matrix *B, const gba_semiring *semiring,
                        int32_t axb_method, int32_t do_sort, bool *mask_applied, gba_report *report);

/* Bflops = cumsum of the per-vector flop count, (B->nvec+1) int64 on device;
 * GB_AxB_saxpy3_flopcount (Source/GB_AxB_saxpy3_flopcount.c:81). */
gba_info gba_AxB_saxpy3_flopcount(int64_t *Mwork, int64_t *Bflops_dev, const gba_matrix *M,
                                  bool Mask_comp, const gba_matrix *A, const gba_matrix *B);

/* C<M> = A'*B, M sparse and not complemented, C gets M's pattern, absent
 * entries become zombies: GB_AxB_dot3 (Source/GB_AxB_dot3.c:37). */
gba_info gba_AxB_dot3(gba_matrix *C, const gba_matrix *M, bool Mask_struct,
                      const gba_matrix *A, const gba_matrix *B, const gba_semiring *semiring,
                      gba_report *report);

/* The four methods below and gba_AxB_saxpy5 take A sparse (saxpy5: bitmap/full) and B bitmap/full
 * (saxpy5: sparse) with at most 65535 vectors on the dense side; a larger problem is declined with
 * GBA_NO_VALUE, and gba_mxm then computes it through the sparse path (same result). */
/* C<#M> = A'*B, C bitmap: GB_AxB_dot2 (Source/GB_AxB_dot2.c:53). */
gba_info gba_AxB_dot2(gba_matrix *C, const gba_matrix *M, bool Mask_comp, bool Mask_struct,
                      const gba_matrix *A, const gba_matrix *B, const gba_semiring *semiring,
                      gba_report *report);

/* C += A'*B in place, C full: GB_AxB_dot4 (Source/GB_AxB_dot4.c:49). */
gba_info gba_AxB_dot4(gba_matrix *C, const gba_matrix *A, const gba_matrix *B,
                      const gba_semiring *semiring, gba_report *report);

/* C += A*B in place, C full, A sparse, B bitmap/full: GB_AxB_saxpy4 (Source/GB_AxB_saxpy4.c:48). */
gba_info gba_AxB_saxpy4(gba_matrix *C, const gba_matrix *A, const gba_matrix *B,
                        const gba_semiring *semiring, gba_report *report);

/* C += A*B in place, C full, A bitmap/full, B sparse: GB_AxB_saxpy5 (Source/GB_AxB_saxpy5.c:55). */
gba_info gba_AxB_saxpy5(gba_matrix *C, const gba_matrix *A, const gba_matrix *B,
                        const gba_semiring *semiring, gba_report *report);

/* C<#M> = A*B, C bitmap: GB_AxB_saxbit (Source/GB_AxB_saxbit.c:46). */
gba_info gba_AxB_saxbit(gba_matrix *C, const gba_matrix *M, bool Mask_comp, bool Mask_struct,
                        const gba_matrix *A, const gba_matrix *B, const gba_semiring *semiring,
                        gba_report *report);

/* -------------------------------------------------------------- utilities -- */

/* Exclusive prefix sum in place over count[0..n] (count[n] receives the total):
 * GB_cumsum (Source/GB_cumsum.c:23).  Device array. */
gba_info gba_cumsum(int64_t *count_dev, int64_t n);

/* The same prefix sum for callers that cannot allocate device memory: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference (its
 * GB_cumsum takes its workspace from the Werk stack).  Calls no hipMalloc / hipFree and does not
 * synchronise: the kernels are queued on the library stream (gba_stream) and the tile sums of the
 * scan live in a 525 312-byte device array inside the library, one copy per GPU, held until the
 * process ends.  n <= GBA_CUMSUM_NOALLOC_MAX, larger n is rejected with GBA_INVALID_VALUE.  Every call
 * on a GPU shares that GPU's array: calls that overlap on one GPU (two streams, two threads) race. */
#define GBA_CUMSUM_NOALLOC_MAX ((int64_t)65535 * 2048 - 1)
gba_info gba_cumsum_noalloc(int64_t *count_dev, int64_t n);

/* Kill zombies and sort every vector: what GrB_wait(C, GrB_MATERIALIZE) does to
 * a jumbled saxpy3 result or a dot3 result with zombies (Source/GB_wait.c,
 * Source/GB_unjumble.c). */
gba_info gba_matrix_wait(gba_matrix *C);

/* AT = A' (pattern+values), sparse in, sparse out: GB_transpose_cast
 * (Source/GB_transpose_cast.c:1-38), bucket method. */
gba_info gba_transpose(gba_matrix *AT, const gba_matrix *A);

/* Flop-balanced 1-D partition of the vectors of the left operand for nparts
 * devices: part[r]..part[r+1] is the vector range of rank r.  Uses the same
 * per-vector flop count as saxpy3 (prefix sum then GB_pslice-style split,
 * Source/GB_pslice.c:116).  part has nparts+1 entries on the host. */
gba_info gba_partition_by_flops(int64_t *part, int32_t nparts,
                                const gba_matrix *A, const gba_matrix *B);

#ifdef __cplusplus
}
#endif
#endif /* GRAPHBLAS_AMD_H */
