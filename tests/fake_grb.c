/* TEST DOUBLE -- not SuiteSparse:GraphBLAS and not derived from its sources.
 *
 * shim/grb_shim.c is written against the public interface of the CPU library (Include/GraphBLAS.h).  That library
 * cannot be built in this repository's environment, so the shim is EXECUTED against this stand-in: the few dozen
 * public calls the shim makes (object queries, GxB_*_unpack_* / GxB_*_pack_*, operator introspection), implemented
 * over trivial structs, plus GrB_mxm / GrB_mxv / GrB_vxm stubs that only record that a call fell through to "the CPU
 * library".  It pins nothing about numerical parity (the multiply itself runs in libgraphblas_amd.so and is checked
 * against the oracle by the test); it shows that the shim's ownership transfers, alias handling and fall-through
 * work when run.  Prototypes come from the reference's header, so a mismatch with the real interface does not
 * compile.  The fake_* functions at the end are the test's own handle constructors and accessors.
 */
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "GraphBLAS.h"

struct GB_Type_opaque { int32_t code ; size_t size ; const char *name, *defn ; } ;
struct GB_BinaryOp_opaque { const char *name ; int32_t xcode, ycode, zcode ; const char *c_name, *c_defn ; } ;
struct GB_Monoid_opaque { GrB_BinaryOp op ; } ;
struct GB_Semiring_opaque { GrB_Monoid add ; GrB_BinaryOp mult ; } ;
struct GB_Descriptor_opaque { int32_t out, mask, in0, in1, axb, sort ; } ;

/* a matrix: whatever arrays its format needs, or nothing while it is unpacked */
struct GB_Matrix_opaque
{
    GrB_Type type ;
    GrB_Index nrows, ncols ;
    int sparsity ;              /* GxB_HYPERSPARSE, GxB_SPARSE, GxB_BITMAP, GxB_FULL */
    bool by_row, iso, jumbled, held ;
    GrB_Index *p, *h, *i ; void *x ; int8_t *b ;
    GrB_Index p_size, h_size, i_size, x_size, b_size, nvals, nvec ;
    double hyper_switch ;       /* 0 in a fresh handle of this double (the library's own default is 0.0625) */
} ;
struct GB_Vector_opaque
{
    GrB_Type type ;
    GrB_Index n ;
    int sparsity ;
    bool iso, jumbled, held ;
    GrB_Index *i ; void *x ; int8_t *b ;
    GrB_Index i_size, x_size, b_size, nvals ;
} ;

/* ------------------------------------------------------------------ allocator -- */
static void *(*fk_malloc) (size_t) = malloc ;
static void *(*fk_calloc) (size_t, size_t) = calloc ;
static void *(*fk_realloc) (void *, size_t) = realloc ;
static void (*fk_free) (void *) = free ;

GrB_Info GxB_Global_Option_get_FUNCTION (GxB_Option_Field field, void **value)
{
    if (value == NULL) return (GrB_NULL_POINTER) ;
    switch ((int) field)
    {
        case GxB_MALLOC_FUNCTION : (*value) = (void *) fk_malloc ; break ;
        case GxB_CALLOC_FUNCTION : (*value) = (void *) fk_calloc ; break ;
        case GxB_REALLOC_FUNCTION : (*value) = (void *) fk_realloc ; break ;
        case GxB_FREE_FUNCTION : (*value) = (void *) fk_free ; break ;
        default : return (GrB_INVALID_VALUE) ;
    }
    return (GrB_SUCCESS) ;
}

/* ---------------------------------------------------------------- built-ins -- */
#define FK_TYPE(sym, CODE, ctype) static struct GB_Type_opaque fk_##sym = { CODE, sizeof (ctype), #ctype, "" } ; GrB_Type sym = &fk_##sym ;
FK_TYPE (GrB_BOOL, GrB_BOOL_CODE, bool)
FK_TYPE (GrB_INT32, GrB_INT32_CODE, int32_t)
FK_TYPE (GrB_INT64, GrB_INT64_CODE, int64_t)
FK_TYPE (GrB_FP32, GrB_FP32_CODE, float)
FK_TYPE (GrB_FP64, GrB_FP64_CODE, double)

#define FK_OP(sym, X, Y, Z) static struct GB_BinaryOp_opaque fk_##sym = { #sym, X, Y, Z, "", "" } ; GrB_BinaryOp sym = &fk_##sym ;
FK_OP (GrB_PLUS_FP64, GrB_FP64_CODE, GrB_FP64_CODE, GrB_FP64_CODE)
FK_OP (GrB_TIMES_FP64, GrB_FP64_CODE, GrB_FP64_CODE, GrB_FP64_CODE)
FK_OP (GrB_MIN_FP64, GrB_FP64_CODE, GrB_FP64_CODE, GrB_FP64_CODE)
FK_OP (GrB_PLUS_FP32, GrB_FP32_CODE, GrB_FP32_CODE, GrB_FP32_CODE)
FK_OP (GrB_TIMES_FP32, GrB_FP32_CODE, GrB_FP32_CODE, GrB_FP32_CODE)
FK_OP (GrB_PLUS_INT64, GrB_INT64_CODE, GrB_INT64_CODE, GrB_INT64_CODE)
FK_OP (GxB_PAIR_INT64, GrB_INT64_CODE, GrB_INT64_CODE, GrB_INT64_CODE)
FK_OP (GxB_ANY_INT64, GrB_INT64_CODE, GrB_INT64_CODE, GrB_INT64_CODE)
/* a positional operator has no input types: the query answers GrB_NO_VALUE (code -1 here) */
FK_OP (GxB_SECONDI_INT64, -1, -1, GrB_INT64_CODE)
/* an operator the shim has no code for: the call must fall through */
FK_OP (GxB_POW_FP64, GrB_FP64_CODE, GrB_FP64_CODE, GrB_FP64_CODE)

#define FK_SEMIRING(sym, ADD, MULT) static struct GB_Monoid_opaque fk_mon_##sym = { &fk_##ADD } ; \
    static struct GB_Semiring_opaque fk_##sym = { &fk_mon_##sym, &fk_##MULT } ; GrB_Semiring sym = &fk_##sym ;
FK_SEMIRING (GrB_PLUS_TIMES_SEMIRING_FP64, GrB_PLUS_FP64, GrB_TIMES_FP64)
FK_SEMIRING (GrB_PLUS_TIMES_SEMIRING_FP32, GrB_PLUS_FP32, GrB_TIMES_FP32)
FK_SEMIRING (GrB_MIN_PLUS_SEMIRING_FP64, GrB_MIN_FP64, GrB_PLUS_FP64)
FK_SEMIRING (GxB_PLUS_PAIR_INT64, GrB_PLUS_INT64, GxB_PAIR_INT64)
FK_SEMIRING (GxB_ANY_SECONDI_INT64, GxB_ANY_INT64, GxB_SECONDI_INT64)
FK_SEMIRING (GxB_PLUS_POW_FP64, GrB_PLUS_FP64, GxB_POW_FP64)

#define FK_DESC(sym, OUT, MASK, IN0, IN1) static struct GB_Descriptor_opaque fk_##sym = { OUT, MASK, IN0, IN1, 0, 0 } ; GrB_Descriptor sym = &fk_##sym ;
FK_DESC (GrB_DESC_T0, 0, 0, GrB_TRAN, 0)
FK_DESC (GrB_DESC_T1, 0, 0, 0, GrB_TRAN)
FK_DESC (GrB_DESC_ST1, 0, GrB_STRUCTURE, 0, GrB_TRAN)
FK_DESC (GrB_DESC_RSC, GrB_REPLACE, GrB_COMP + GrB_STRUCTURE, 0, 0)
FK_DESC (GrB_DESC_R, GrB_REPLACE, 0, 0, 0)

/* ------------------------------------------------------------------ queries -- */
GrB_Info GrB_Matrix_nrows (GrB_Index *nrows, const GrB_Matrix A) { (*nrows) = A->nrows ; return (GrB_SUCCESS) ; }
GrB_Info GrB_Matrix_ncols (GrB_Index *ncols, const GrB_Matrix A) { (*ncols) = A->ncols ; return (GrB_SUCCESS) ; }
GrB_Info GrB_Vector_size (GrB_Index *n, const GrB_Vector v) { (*n) = v->n ; return (GrB_SUCCESS) ; }
GrB_Info GxB_Matrix_type (GrB_Type *type, const GrB_Matrix A) { (*type) = A->type ; return (GrB_SUCCESS) ; }
GrB_Info GxB_Vector_type (GrB_Type *type, const GrB_Vector v) { (*type) = v->type ; return (GrB_SUCCESS) ; }
GrB_Info GxB_Type_size (size_t *size, const GrB_Type type) { (*size) = type->size ; return (GrB_SUCCESS) ; }

GrB_Info GrB_Matrix_get_INT32 (GrB_Matrix A, int32_t *value, GrB_Field field)
{
    if (field != GrB_EL_TYPE_CODE) return (GrB_INVALID_VALUE) ;
    (*value) = A->type->code ;
    return (GrB_SUCCESS) ;
}
GrB_Info GrB_Vector_get_INT32 (GrB_Vector v, int32_t *value, GrB_Field field)
{
    if (field != GrB_EL_TYPE_CODE) return (GrB_INVALID_VALUE) ;
    (*value) = v->type->code ;
    return (GrB_SUCCESS) ;
}
GrB_Info GxB_Matrix_Option_get (GrB_Matrix A, GxB_Option_Field field, ...)
{
    va_list ap ; va_start (ap, field) ;
    GrB_Info info = GrB_SUCCESS ;
    if (field == GxB_SPARSITY_STATUS) { int *s = va_arg (ap, int *) ; (*s) = A->sparsity ; }
    else if (field == GxB_FORMAT) { GxB_Format_Value *f = va_arg (ap, GxB_Format_Value *) ; (*f) = A->by_row ? GxB_BY_ROW : GxB_BY_COL ; }
    else if (field == GxB_HYPER_SWITCH) { double *h = va_arg (ap, double *) ; (*h) = A->hyper_switch ; }
    else info = GrB_INVALID_VALUE ;
    va_end (ap) ;
    return (info) ;
}
GrB_Info GxB_Vector_Option_get (GrB_Vector v, GxB_Option_Field field, ...)
{
    va_list ap ; va_start (ap, field) ;
    GrB_Info info = GrB_SUCCESS ;
    if (field == GxB_SPARSITY_STATUS) { int *s = va_arg (ap, int *) ; (*s) = v->sparsity ; }
    else info = GrB_INVALID_VALUE ;
    va_end (ap) ;
    return (info) ;
}

static const char *fk_op_string (GrB_BinaryOp op, GrB_Field field)
{
    if (field == GrB_NAME) return (op->name) ;
    if ((int) field == GxB_JIT_C_NAME) return (op->c_name) ;
    if ((int) field == GxB_JIT_C_DEFINITION) return (op->c_defn) ;
    return (NULL) ;
}
GrB_Info GrB_BinaryOp_get_String (GrB_BinaryOp op, char *value, GrB_Field field)
{
    const char *s = fk_op_string (op, field) ;
    if (s == NULL) return (GrB_INVALID_VALUE) ;
    strcpy (value, s) ;
    return (GrB_SUCCESS) ;
}
GrB_Info GrB_BinaryOp_get_SIZE (GrB_BinaryOp op, size_t *value, GrB_Field field)
{
    const char *s = fk_op_string (op, field) ;
    if (s == NULL) return (GrB_INVALID_VALUE) ;
    (*value) = strlen (s) + 1 ;
    return (GrB_SUCCESS) ;
}
GrB_Info GrB_BinaryOp_get_INT32 (GrB_BinaryOp op, int32_t *value, GrB_Field field)
{
    int32_t c ;
    if (field == GrB_INP0_TYPE_CODE) c = op->xcode ;
    else if (field == GrB_INP1_TYPE_CODE) c = op->ycode ;
    else if (field == GrB_OUTP_TYPE_CODE) c = op->zcode ;
    else return (GrB_INVALID_VALUE) ;
    if (c < 0) return (GrB_NO_VALUE) ;
    (*value) = c ;
    return (GrB_SUCCESS) ;
}
GrB_Info GrB_Type_get_String (GrB_Type t, char *value, GrB_Field field)
{
    const char *s = ((int) field == GxB_JIT_C_NAME) ? t->name : ((int) field == GxB_JIT_C_DEFINITION) ? t->defn : NULL ;
    if (s == NULL) return (GrB_INVALID_VALUE) ;
    strcpy (value, s) ;
    return (GrB_SUCCESS) ;
}
GrB_Info GrB_Type_get_SIZE (GrB_Type t, size_t *value, GrB_Field field)
{
    const char *s = ((int) field == GxB_JIT_C_NAME) ? t->name : ((int) field == GxB_JIT_C_DEFINITION) ? t->defn : NULL ;
    if (s == NULL) return (GrB_INVALID_VALUE) ;
    (*value) = strlen (s) + 1 ;
    return (GrB_SUCCESS) ;
}
GrB_Info GxB_Semiring_add (GrB_Monoid *add, GrB_Semiring semiring) { (*add) = semiring->add ; return (GrB_SUCCESS) ; }
GrB_Info GxB_Semiring_multiply (GrB_BinaryOp *multiply, GrB_Semiring semiring) { (*multiply) = semiring->mult ; return (GrB_SUCCESS) ; }
GrB_Info GxB_Monoid_operator (GrB_BinaryOp *op, GrB_Monoid monoid) { (*op) = monoid->op ; return (GrB_SUCCESS) ; }
GrB_Info GxB_Desc_get_INT32 (GrB_Descriptor desc, GrB_Desc_Field field, int32_t *value)
{
    if (desc == NULL) { (*value) = 0 ; return (GrB_SUCCESS) ; }
    switch ((int) field)
    {
        case GrB_OUTP : (*value) = desc->out ; break ;
        case GrB_MASK : (*value) = desc->mask ; break ;
        case GrB_INP0 : (*value) = desc->in0 ; break ;
        case GrB_INP1 : (*value) = desc->in1 ; break ;
        case GxB_AxB_METHOD : (*value) = desc->axb ; break ;
        case GxB_SORT : (*value) = desc->sort ; break ;
        default : return (GrB_INVALID_VALUE) ;
    }
    return (GrB_SUCCESS) ;
}

/* ------------------------------------------------------------ unpack / pack -- */
/* unpack: the arrays leave the object (which must hold that format); pack: they come back and the caller's
 * pointers are cleared, as the real functions do */
#define FK_TAKE(dst, size_dst, src, size_src) do { (*(dst)) = (src) ; (*(size_dst)) = (size_src) ; (src) = NULL ; } while (0)
#define FK_GIVE(dst, size_dst, src, size_src) do { (dst) = (*(src)) ; (size_dst) = (size_src) ; (*(src)) = NULL ; } while (0)

static GrB_Info fk_unpack_sparse (GrB_Matrix A, bool by_row, bool hyper, GrB_Index **Ap, GrB_Index **Ah, GrB_Index **Ai,
    void **Ax, GrB_Index *Ap_size, GrB_Index *Ah_size, GrB_Index *Ai_size, GrB_Index *Ax_size, bool *iso,
    GrB_Index *nvec, bool *jumbled)
{
    if (A == NULL || !A->held) return (GrB_INVALID_OBJECT) ;
    if (A->by_row != by_row || A->sparsity != (hyper ? GxB_HYPERSPARSE : GxB_SPARSE)) return (GrB_INVALID_VALUE) ;
    FK_TAKE (Ap, Ap_size, A->p, A->p_size) ;
    if (hyper) { FK_TAKE (Ah, Ah_size, A->h, A->h_size) ; (*nvec) = A->nvec ; }
    FK_TAKE (Ai, Ai_size, A->i, A->i_size) ;
    FK_TAKE (Ax, Ax_size, A->x, A->x_size) ;
    (*iso) = A->iso ;
    if (jumbled != NULL) (*jumbled) = A->jumbled ;
    A->held = false ;
    return (GrB_SUCCESS) ;
}
static GrB_Info fk_pack_sparse (GrB_Matrix A, bool by_row, bool hyper, GrB_Index **Ap, GrB_Index **Ah, GrB_Index **Ai,
    void **Ax, GrB_Index Ap_size, GrB_Index Ah_size, GrB_Index Ai_size, GrB_Index Ax_size, bool iso, GrB_Index nvec,
    bool jumbled)
{
    if (A == NULL || A->held) return (GrB_INVALID_OBJECT) ;
    if ((*Ap) == NULL || (*Ai) == NULL || (*Ax) == NULL || (hyper && (*Ah) == NULL)) return (GrB_NULL_POINTER) ;
    const GrB_Index nv = hyper ? nvec : (by_row ? A->nrows : A->ncols) ;
    if (Ap_size < (nv + 1) * sizeof (GrB_Index)) return (GrB_INVALID_VALUE) ;
    FK_GIVE (A->p, A->p_size, Ap, Ap_size) ;
    if (hyper) FK_GIVE (A->h, A->h_size, Ah, Ah_size) ;
    FK_GIVE (A->i, A->i_size, Ai, Ai_size) ;
    FK_GIVE (A->x, A->x_size, Ax, Ax_size) ;
    A->by_row = by_row ; A->sparsity = hyper ? GxB_HYPERSPARSE : GxB_SPARSE ; A->iso = iso ; A->jumbled = jumbled ;
    A->nvec = nv ; A->nvals = A->p [nv] ; A->held = true ;
    return (GrB_SUCCESS) ;
}

GrB_Info GxB_Matrix_unpack_CSR (GrB_Matrix A, GrB_Index **Ap, GrB_Index **Aj, void **Ax, GrB_Index *Ap_size,
    GrB_Index *Aj_size, GrB_Index *Ax_size, bool *iso, bool *jumbled, const GrB_Descriptor desc)
{ (void) desc ; return (fk_unpack_sparse (A, true, false, Ap, NULL, Aj, Ax, Ap_size, NULL, Aj_size, Ax_size, iso, NULL, jumbled)) ; }
GrB_Info GxB_Matrix_unpack_CSC (GrB_Matrix A, GrB_Index **Ap, GrB_Index **Ai, void **Ax, GrB_Index *Ap_size,
    GrB_Index *Ai_size, GrB_Index *Ax_size, bool *iso, bool *jumbled, const GrB_Descriptor desc)
{ (void) desc ; return (fk_unpack_sparse (A, false, false, Ap, NULL, Ai, Ax, Ap_size, NULL, Ai_size, Ax_size, iso, NULL, jumbled)) ; }
GrB_Info GxB_Matrix_unpack_HyperCSR (GrB_Matrix A, GrB_Index **Ap, GrB_Index **Ah, GrB_Index **Aj, void **Ax,
    GrB_Index *Ap_size, GrB_Index *Ah_size, GrB_Index *Aj_size, GrB_Index *Ax_size, bool *iso, GrB_Index *nvec,
    bool *jumbled, const GrB_Descriptor desc)
{ (void) desc ; return (fk_unpack_sparse (A, true, true, Ap, Ah, Aj, Ax, Ap_size, Ah_size, Aj_size, Ax_size, iso, nvec, jumbled)) ; }
GrB_Info GxB_Matrix_unpack_HyperCSC (GrB_Matrix A, GrB_Index **Ap, GrB_Index **Ah, GrB_Index **Ai, void **Ax,
    GrB_Index *Ap_size, GrB_Index *Ah_size, GrB_Index *Ai_size, GrB_Index *Ax_size, bool *iso, GrB_Index *nvec,
    bool *jumbled, const GrB_Descriptor desc)
{ (void) desc ; return (fk_unpack_sparse (A, false, true, Ap, Ah, Ai, Ax, Ap_size, Ah_size, Ai_size, Ax_size, iso, nvec, jumbled)) ; }
GrB_Info GxB_Matrix_pack_CSR (GrB_Matrix A, GrB_Index **Ap, GrB_Index **Aj, void **Ax, GrB_Index Ap_size,
    GrB_Index Aj_size, GrB_Index Ax_size, bool iso, bool jumbled, const GrB_Descriptor desc)
{ (void) desc ; return (fk_pack_sparse (A, true, false, Ap, NULL, Aj, Ax, Ap_size, 0, Aj_size, Ax_size, iso, 0, jumbled)) ; }
GrB_Info GxB_Matrix_pack_CSC (GrB_Matrix A, GrB_Index **Ap, GrB_Index **Ai, void **Ax, GrB_Index Ap_size,
    GrB_Index Ai_size, GrB_Index Ax_size, bool iso, bool jumbled, const GrB_Descriptor desc)
{ (void) desc ; return (fk_pack_sparse (A, false, false, Ap, NULL, Ai, Ax, Ap_size, 0, Ai_size, Ax_size, iso, 0, jumbled)) ; }
GrB_Info GxB_Matrix_pack_HyperCSR (GrB_Matrix A, GrB_Index **Ap, GrB_Index **Ah, GrB_Index **Aj, void **Ax,
    GrB_Index Ap_size, GrB_Index Ah_size, GrB_Index Aj_size, GrB_Index Ax_size, bool iso, GrB_Index nvec, bool jumbled,
    const GrB_Descriptor desc)
{ (void) desc ; return (fk_pack_sparse (A, true, true, Ap, Ah, Aj, Ax, Ap_size, Ah_size, Aj_size, Ax_size, iso, nvec, jumbled)) ; }
GrB_Info GxB_Matrix_pack_HyperCSC (GrB_Matrix A, GrB_Index **Ap, GrB_Index **Ah, GrB_Index **Ai, void **Ax,
    GrB_Index Ap_size, GrB_Index Ah_size, GrB_Index Ai_size, GrB_Index Ax_size, bool iso, GrB_Index nvec, bool jumbled,
    const GrB_Descriptor desc)
{ (void) desc ; return (fk_pack_sparse (A, false, true, Ap, Ah, Ai, Ax, Ap_size, Ah_size, Ai_size, Ax_size, iso, nvec, jumbled)) ; }

static GrB_Info fk_unpack_dense (GrB_Matrix A, bool by_row, bool bitmap, int8_t **Ab, void **Ax, GrB_Index *Ab_size,
    GrB_Index *Ax_size, bool *iso, GrB_Index *nvals)
{
    if (A == NULL || !A->held) return (GrB_INVALID_OBJECT) ;
    if (A->by_row != by_row || A->sparsity != (bitmap ? GxB_BITMAP : GxB_FULL)) return (GrB_INVALID_VALUE) ;
    if (bitmap) { FK_TAKE (Ab, Ab_size, A->b, A->b_size) ; (*nvals) = A->nvals ; }
    FK_TAKE (Ax, Ax_size, A->x, A->x_size) ;
    (*iso) = A->iso ;
    A->held = false ;
    return (GrB_SUCCESS) ;
}
static GrB_Info fk_pack_dense (GrB_Matrix A, bool by_row, bool bitmap, int8_t **Ab, void **Ax, GrB_Index Ab_size,
    GrB_Index Ax_size, bool iso, GrB_Index nvals)
{
    if (A == NULL || A->held) return (GrB_INVALID_OBJECT) ;
    if ((*Ax) == NULL || (bitmap && (*Ab) == NULL)) return (GrB_NULL_POINTER) ;
    if (bitmap) FK_GIVE (A->b, A->b_size, Ab, Ab_size) ;
    FK_GIVE (A->x, A->x_size, Ax, Ax_size) ;
    A->by_row = by_row ; A->sparsity = bitmap ? GxB_BITMAP : GxB_FULL ; A->iso = iso ; A->jumbled = false ;
    A->nvals = bitmap ? nvals : A->nrows * A->ncols ; A->held = true ;
    return (GrB_SUCCESS) ;
}
GrB_Info GxB_Matrix_unpack_BitmapR (GrB_Matrix A, int8_t **Ab, void **Ax, GrB_Index *Ab_size, GrB_Index *Ax_size,
    bool *iso, GrB_Index *nvals, const GrB_Descriptor desc)
{ (void) desc ; return (fk_unpack_dense (A, true, true, Ab, Ax, Ab_size, Ax_size, iso, nvals)) ; }
GrB_Info GxB_Matrix_unpack_BitmapC (GrB_Matrix A, int8_t **Ab, void **Ax, GrB_Index *Ab_size, GrB_Index *Ax_size,
    bool *iso, GrB_Index *nvals, const GrB_Descriptor desc)
{ (void) desc ; return (fk_unpack_dense (A, false, true, Ab, Ax, Ab_size, Ax_size, iso, nvals)) ; }
GrB_Info GxB_Matrix_unpack_FullR (GrB_Matrix A, void **Ax, GrB_Index *Ax_size, bool *iso, const GrB_Descriptor desc)
{ (void) desc ; return (fk_unpack_dense (A, true, false, NULL, Ax, NULL, Ax_size, iso, NULL)) ; }
GrB_Info GxB_Matrix_unpack_FullC (GrB_Matrix A, void **Ax, GrB_Index *Ax_size, bool *iso, const GrB_Descriptor desc)
{ (void) desc ; return (fk_unpack_dense (A, false, false, NULL, Ax, NULL, Ax_size, iso, NULL)) ; }
GrB_Info GxB_Matrix_pack_BitmapR (GrB_Matrix A, int8_t **Ab, void **Ax, GrB_Index Ab_size, GrB_Index Ax_size,
    bool iso, GrB_Index nvals, const GrB_Descriptor desc)
{ (void) desc ; return (fk_pack_dense (A, true, true, Ab, Ax, Ab_size, Ax_size, iso, nvals)) ; }
GrB_Info GxB_Matrix_pack_BitmapC (GrB_Matrix A, int8_t **Ab, void **Ax, GrB_Index Ab_size, GrB_Index Ax_size,
    bool iso, GrB_Index nvals, const GrB_Descriptor desc)
{ (void) desc ; return (fk_pack_dense (A, false, true, Ab, Ax, Ab_size, Ax_size, iso, nvals)) ; }
GrB_Info GxB_Matrix_pack_FullR (GrB_Matrix A, void **Ax, GrB_Index Ax_size, bool iso, const GrB_Descriptor desc)
{ (void) desc ; return (fk_pack_dense (A, true, false, NULL, Ax, 0, Ax_size, iso, 0)) ; }
GrB_Info GxB_Matrix_pack_FullC (GrB_Matrix A, void **Ax, GrB_Index Ax_size, bool iso, const GrB_Descriptor desc)
{ (void) desc ; return (fk_pack_dense (A, false, false, NULL, Ax, 0, Ax_size, iso, 0)) ; }

GrB_Info GxB_Vector_unpack_CSC (GrB_Vector v, GrB_Index **vi, void **vx, GrB_Index *vi_size, GrB_Index *vx_size,
    bool *iso, GrB_Index *nvals, bool *jumbled, const GrB_Descriptor desc)
{
    (void) desc ;
    if (v == NULL || !v->held) return (GrB_INVALID_OBJECT) ;
    if (v->sparsity != GxB_SPARSE) return (GrB_INVALID_VALUE) ;
    FK_TAKE (vi, vi_size, v->i, v->i_size) ;
    FK_TAKE (vx, vx_size, v->x, v->x_size) ;
    (*iso) = v->iso ; (*nvals) = v->nvals ;
    if (jumbled != NULL) (*jumbled) = v->jumbled ;
    v->held = false ;
    return (GrB_SUCCESS) ;
}
GrB_Info GxB_Vector_pack_CSC (GrB_Vector v, GrB_Index **vi, void **vx, GrB_Index vi_size, GrB_Index vx_size,
    bool iso, GrB_Index nvals, bool jumbled, const GrB_Descriptor desc)
{
    (void) desc ;
    if (v == NULL || v->held) return (GrB_INVALID_OBJECT) ;
    if ((*vi) == NULL || (*vx) == NULL) return (GrB_NULL_POINTER) ;
    if (vi_size < nvals * sizeof (GrB_Index)) return (GrB_INVALID_VALUE) ;
    FK_GIVE (v->i, v->i_size, vi, vi_size) ;
    FK_GIVE (v->x, v->x_size, vx, vx_size) ;
    v->sparsity = GxB_SPARSE ; v->iso = iso ; v->jumbled = jumbled ; v->nvals = nvals ; v->held = true ;
    return (GrB_SUCCESS) ;
}
GrB_Info GxB_Vector_unpack_Bitmap (GrB_Vector v, int8_t **vb, void **vx, GrB_Index *vb_size, GrB_Index *vx_size,
    bool *iso, GrB_Index *nvals, const GrB_Descriptor desc)
{
    (void) desc ;
    if (v == NULL || !v->held) return (GrB_INVALID_OBJECT) ;
    if (v->sparsity != GxB_BITMAP) return (GrB_INVALID_VALUE) ;
    FK_TAKE (vb, vb_size, v->b, v->b_size) ;
    FK_TAKE (vx, vx_size, v->x, v->x_size) ;
    (*iso) = v->iso ; (*nvals) = v->nvals ;
    v->held = false ;
    return (GrB_SUCCESS) ;
}
GrB_Info GxB_Vector_pack_Bitmap (GrB_Vector v, int8_t **vb, void **vx, GrB_Index vb_size, GrB_Index vx_size,
    bool iso, GrB_Index nvals, const GrB_Descriptor desc)
{
    (void) desc ;
    if (v == NULL || v->held) return (GrB_INVALID_OBJECT) ;
    if ((*vb) == NULL || (*vx) == NULL) return (GrB_NULL_POINTER) ;
    FK_GIVE (v->b, v->b_size, vb, vb_size) ;
    FK_GIVE (v->x, v->x_size, vx, vx_size) ;
    v->sparsity = GxB_BITMAP ; v->iso = iso ; v->jumbled = false ; v->nvals = nvals ; v->held = true ;
    return (GrB_SUCCESS) ;
}
GrB_Info GxB_Vector_unpack_Full (GrB_Vector v, void **vx, GrB_Index *vx_size, bool *iso, const GrB_Descriptor desc)
{
    (void) desc ;
    if (v == NULL || !v->held) return (GrB_INVALID_OBJECT) ;
    if (v->sparsity != GxB_FULL) return (GrB_INVALID_VALUE) ;
    FK_TAKE (vx, vx_size, v->x, v->x_size) ;
    (*iso) = v->iso ;
    v->held = false ;
    return (GrB_SUCCESS) ;
}
GrB_Info GxB_Vector_pack_Full (GrB_Vector v, void **vx, GrB_Index vx_size, bool iso, const GrB_Descriptor desc)
{
    (void) desc ;
    if (v == NULL || v->held) return (GrB_INVALID_OBJECT) ;
    if ((*vx) == NULL) return (GrB_NULL_POINTER) ;
    FK_GIVE (v->x, v->x_size, vx, vx_size) ;
    v->sparsity = GxB_FULL ; v->iso = iso ; v->jumbled = false ; v->nvals = v->n ; v->held = true ;
    return (GrB_SUCCESS) ;
}

/* ------------------------------------------------- "the CPU library's" multiply -- */
/* what the shim reaches through dlsym (RTLD_NEXT, ...): nothing is computed, the call is recorded */
static int fk_cpu_calls [3] = {0, 0, 0} ;       /* mxm, vxm, mxv */
static bool fk_cpu_saw_intact = true ;          /* every handle of the last fall-through call held its arrays */

GrB_Info GrB_mxm (GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_Semiring semiring,
    const GrB_Matrix A, const GrB_Matrix B, const GrB_Descriptor desc)
{
    (void) accum ; (void) semiring ; (void) desc ;
    fk_cpu_calls [0]++ ;
    fk_cpu_saw_intact = C->held && A->held && B->held && (Mask == NULL || Mask->held) ;
    return (GrB_SUCCESS) ;
}
GrB_Info GrB_vxm (GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Semiring semiring,
    const GrB_Vector u, const GrB_Matrix A, const GrB_Descriptor desc)
{
    (void) accum ; (void) semiring ; (void) desc ;
    fk_cpu_calls [1]++ ;
    fk_cpu_saw_intact = w->held && u->held && A->held && (mask == NULL || mask->held) ;
    return (GrB_SUCCESS) ;
}
GrB_Info GrB_mxv (GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Semiring semiring,
    const GrB_Matrix A, const GrB_Vector u, const GrB_Descriptor desc)
{
    (void) accum ; (void) semiring ; (void) desc ;
    fk_cpu_calls [2]++ ;
    fk_cpu_saw_intact = w->held && u->held && A->held && (mask == NULL || mask->held) ;
    return (GrB_SUCCESS) ;
}

/* ------------------------------------------------------- the test's own helpers -- */
int fake_cpu_calls (int which) { return (fk_cpu_calls [which]) ; }
int fake_cpu_saw_intact (void) { return (fk_cpu_saw_intact ? 1 : 0) ; }
void fake_set_allocator (void *m, void *c, void *r, void *f)
{
    fk_malloc = (void *(*) (size_t)) m ; fk_calloc = (void *(*) (size_t, size_t)) c ;
    fk_realloc = (void *(*) (void *, size_t)) r ; fk_free = (void (*) (void *)) f ;
}

static void *fk_copy (const void *src, size_t bytes, GrB_Index *size)
{
    const size_t n = bytes ? bytes : 8 ;
    void *p = fk_malloc (n) ;
    if (p != NULL && bytes && src != NULL) memcpy (p, src, bytes) ;
    (*size) = (GrB_Index) n ;
    return (p) ;
}

static GrB_Type fk_type_of (int code)
{
    switch (code)
    {
        case GrB_BOOL_CODE : return (GrB_BOOL) ;
        case GrB_INT32_CODE : return (GrB_INT32) ;
        case GrB_INT64_CODE : return (GrB_INT64) ;
        case GrB_FP32_CODE : return (GrB_FP32) ;
        default : return (GrB_FP64) ;
    }
}

/* sparse (hyper == 0) or hypersparse matrix from the caller's arrays (copied into the installed allocator) */
GrB_Matrix fake_matrix_sparse (int type_code, int64_t nrows, int64_t ncols, int by_row, int hyper, int64_t nvec,
    const int64_t *p, const int64_t *h, const int64_t *i, const void *x, int iso, int jumbled)
{
    GrB_Matrix A = calloc (1, sizeof (struct GB_Matrix_opaque)) ;
    A->type = fk_type_of (type_code) ; A->nrows = (GrB_Index) nrows ; A->ncols = (GrB_Index) ncols ;
    A->by_row = by_row ; A->sparsity = hyper ? GxB_HYPERSPARSE : GxB_SPARSE ; A->iso = iso ; A->jumbled = jumbled ;
    A->nvec = (GrB_Index) (hyper ? nvec : (by_row ? nrows : ncols)) ;
    A->nvals = (GrB_Index) p [A->nvec] ;
    A->p = fk_copy (p, (A->nvec + 1) * 8, &(A->p_size)) ;
    if (hyper) A->h = fk_copy (h, A->nvec * 8, &(A->h_size)) ;
    A->i = fk_copy (i, A->nvals * 8, &(A->i_size)) ;
    A->x = fk_copy (x, (iso ? 1 : A->nvals) * A->type->size, &(A->x_size)) ;
    A->held = true ;
    return (A) ;
}
/* bitmap (b != NULL) or full matrix */
GrB_Matrix fake_matrix_dense (int type_code, int64_t nrows, int64_t ncols, int by_row, const int8_t *b, const void *x,
    int64_t nvals)
{
    GrB_Matrix A = calloc (1, sizeof (struct GB_Matrix_opaque)) ;
    A->type = fk_type_of (type_code) ; A->nrows = (GrB_Index) nrows ; A->ncols = (GrB_Index) ncols ;
    A->by_row = by_row ; A->sparsity = b ? GxB_BITMAP : GxB_FULL ;
    const size_t n = (size_t) nrows * (size_t) ncols ;
    if (b) A->b = fk_copy (b, n, &(A->b_size)) ;
    A->x = fk_copy (x, n * A->type->size, &(A->x_size)) ;
    A->nvals = (GrB_Index) (b ? nvals : (int64_t) n) ;
    A->held = true ;
    return (A) ;
}
GrB_Vector fake_vector_sparse (int type_code, int64_t n, int64_t nvals, const int64_t *i, const void *x)
{
    GrB_Vector v = calloc (1, sizeof (struct GB_Vector_opaque)) ;
    v->type = fk_type_of (type_code) ; v->n = (GrB_Index) n ; v->sparsity = GxB_SPARSE ; v->nvals = (GrB_Index) nvals ;
    v->i = fk_copy (i, (size_t) nvals * 8, &(v->i_size)) ;
    v->x = fk_copy (x, (size_t) nvals * v->type->size, &(v->x_size)) ;
    v->held = true ;
    return (v) ;
}
GrB_Vector fake_vector_dense (int type_code, int64_t n, const int8_t *b, const void *x, int64_t nvals)
{
    GrB_Vector v = calloc (1, sizeof (struct GB_Vector_opaque)) ;
    v->type = fk_type_of (type_code) ; v->n = (GrB_Index) n ; v->sparsity = b ? GxB_BITMAP : GxB_FULL ;
    if (b) v->b = fk_copy (b, (size_t) n, &(v->b_size)) ;
    v->x = fk_copy (x, (size_t) n * v->type->size, &(v->x_size)) ;
    v->nvals = (GrB_Index) (b ? nvals : n) ;
    v->held = true ;
    return (v) ;
}
/* what a handle holds now: fields [sparsity, by_row, iso, jumbled, held, nvals, nvec, nrows, ncols], arrays p h i x b */
void fake_matrix_look (GrB_Matrix A, int64_t *fields, void **arrays)
{
    fields [0] = A->sparsity ; fields [1] = A->by_row ; fields [2] = A->iso ; fields [3] = A->jumbled ; fields [4] = A->held ;
    fields [5] = (int64_t) A->nvals ; fields [6] = (int64_t) A->nvec ; fields [7] = (int64_t) A->nrows ; fields [8] = (int64_t) A->ncols ;
    arrays [0] = A->p ; arrays [1] = A->h ; arrays [2] = A->i ; arrays [3] = A->x ; arrays [4] = A->b ;
}
void fake_vector_look (GrB_Vector v, int64_t *fields, void **arrays)
{
    fields [0] = v->sparsity ; fields [1] = 0 ; fields [2] = v->iso ; fields [3] = v->jumbled ; fields [4] = v->held ;
    fields [5] = (int64_t) v->nvals ; fields [6] = 1 ; fields [7] = (int64_t) v->n ; fields [8] = 1 ;
    arrays [0] = NULL ; arrays [1] = NULL ; arrays [2] = v->i ; arrays [3] = v->x ; arrays [4] = v->b ;
}
void fake_matrix_free (GrB_Matrix A)
{
    if (A == NULL) return ;
    fk_free (A->p) ; fk_free (A->h) ; fk_free (A->i) ; fk_free (A->x) ; fk_free (A->b) ; free (A) ;
}
void fake_vector_free (GrB_Vector v)
{
    if (v == NULL) return ;
    fk_free (v->i) ; fk_free (v->x) ; fk_free (v->b) ; free (v) ;
}

/* test helper: the hyper_switch of a handle (GxB_Matrix_Option_set (A, GxB_HYPER_SWITCH, h) in the real library) */
void fake_matrix_set_hyper_switch (GrB_Matrix A, double h) { A->hyper_switch = h ; }
