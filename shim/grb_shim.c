/* Drop-in definitions of GrB_mxm, GrB_mxv and GrB_vxm (Include/GraphBLAS.h:4640-4671) in front of an
 * unchanged SuiteSparse:GraphBLAS: a shared library that is linked or preloaded BEFORE libgraphblas.so.
 *
 *   1. the operands leave their opaque handles in O(1) through the public move interface,
 *      GxB_Matrix_unpack_CSR/CSC/HyperCSR/HyperCSC/BitmapR/BitmapC/FullR/FullC and
 *      GxB_Vector_unpack_CSC/Bitmap/Full (Include/GraphBLAS.h:9869-10158; O(1): 9298-9303);
 *   2. gba_mxm / gba_mxv / gba_vxm run the multiply on the MI355X (include/graphblas_amd.h);
 *   3. everything is packed back (GxB_Matrix_pack_*, GxB_Vector_pack_*, 9342-9760).
 * A semiring over one user-defined type goes to gba_mxm_udt with the strings the application gave to
 * GxB_Type_new / GxB_BinaryOp_new (shim_run_udt).
 * A call the device path does not complete (casts to or from user-defined types, an operator
 * without a built-in code or definition, and EVERY gba_info other than GBA_SUCCESS: not implemented, out of
 * device memory, a compile error of a user-defined operator, a dimension or domain mismatch) goes to the next
 * definition of the same symbol, dlsym (RTLD_NEXT, ...): the CPU library, with its arguments untouched -- C is
 * only replaced after the device call has succeeded -- which then computes the result or reports the error with
 * its own code and message.
 *
 * Aliased handles (Source/GB_mxm.c:48: any of C, Mask, A, B may be the same object; C = A*A, C<L> = L*L',
 * q<!pi> = q'*A, d = d min.+ A) are taken: every distinct handle is moved out once and the roles share its view.
 * The inputs are read from the host arrays, the output is built in device memory of its own, so an input that is
 * also the output is read as it was on entry, which is the semantics of Source/GB_AxB_meta.c:500-509.
 *
 * Memory: the handles hold host arrays of the allocator given to GxB_init.  A, B and the mask are handed
 * over as host views (mem = GBA_MEM_HOST; the library stages them over PCIe).  When the application gave
 * GxB_init the library's own host allocator (gba_host_malloc / _calloc / _realloc / _free: pinned memory the
 * device can address) the views are GBA_MEM_HOST_VISIBLE instead: GrB_mxv / GrB_vxm read their operands in place,
 * GrB_mxm copies at the pinned rate.  C travels to the device before the call (nothing when it has no entries) and
 * back after it, into arrays taken from the library's allocator (GxB_Global_Option_get_FUNCTION
 * (GxB_MALLOC_FUNCTION, ...)), which is what GxB_*_pack_* requires.
 *
 * Threads: between unpack and pack a handle is empty, so the whole window is one critical section (a mutex): calls
 * through the shim are serialised, and no other thread may use A, B, Mask or C during the call (the CPU library
 * allows concurrent readers of A; behind this shim it does not hold).
 *
 * Build:  gcc -O2 -fPIC -shared -I<reference>/Include -I<repo>/include shim/grb_shim.c \
 *             -o libgrb_shim.so -L<repo>/graphblas_amd -lgraphblas_amd -ldl
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <string.h>
#include <stdlib.h>
#include <pthread.h>
#include "GraphBLAS.h"
#include "graphblas_amd.h"

/* the three definitions must have exactly the header's types */
typedef GrB_Info (*shim_mxm_f) (GrB_Matrix, const GrB_Matrix, const GrB_BinaryOp, const GrB_Semiring,
    const GrB_Matrix, const GrB_Matrix, const GrB_Descriptor) ;
typedef GrB_Info (*shim_vxm_f) (GrB_Vector, const GrB_Vector, const GrB_BinaryOp, const GrB_Semiring,
    const GrB_Vector, const GrB_Matrix, const GrB_Descriptor) ;
typedef GrB_Info (*shim_mxv_f) (GrB_Vector, const GrB_Vector, const GrB_BinaryOp, const GrB_Semiring,
    const GrB_Matrix, const GrB_Vector, const GrB_Descriptor) ;
static const shim_mxm_f shim_check_mxm = GrB_mxm ;      /* a mismatch with the header is a compile error */
static const shim_vxm_f shim_check_vxm = GrB_vxm ;
static const shim_mxv_f shim_check_mxv = GrB_mxv ;

#define SHIM_OK(call) do { if ((call) != GrB_SUCCESS) return (false) ; } while (0)

/* ------------------------------------------------------------------ operators -- */

static const struct { const char *name ; gba_opcode code ; } shim_ops [ ] = {
    {"FIRST", GBA_FIRST}, {"SECOND", GBA_SECOND}, {"ANY", GBA_ANY}, {"PAIR", GBA_PAIR}, {"ONEB", GBA_PAIR},
    {"MIN", GBA_MIN}, {"MAX", GBA_MAX}, {"PLUS", GBA_PLUS}, {"MINUS", GBA_MINUS}, {"RMINUS", GBA_RMINUS},
    {"TIMES", GBA_TIMES}, {"DIV", GBA_DIV}, {"RDIV", GBA_RDIV}, {"ISEQ", GBA_ISEQ}, {"ISNE", GBA_ISNE},
    {"ISGT", GBA_ISGT}, {"ISLT", GBA_ISLT}, {"ISGE", GBA_ISGE}, {"ISLE", GBA_ISLE}, {"LOR", GBA_LOR},
    {"LAND", GBA_LAND}, {"LXOR", GBA_LXOR}, {"LXNOR", GBA_EQ}, {"BOR", GBA_BOR}, {"BAND", GBA_BAND},
    {"BXOR", GBA_BXOR}, {"BXNOR", GBA_BXNOR}, {"EQ", GBA_EQ}, {"NE", GBA_NE}, {"GT", GBA_GT}, {"LT", GBA_LT},
    {"GE", GBA_GE}, {"LE", GBA_LE}, {"FIRSTI", GBA_FIRSTI}, {"FIRSTI1", GBA_FIRSTI1}, {"FIRSTJ", GBA_FIRSTJ},
    {"FIRSTJ1", GBA_FIRSTJ1}, {"SECONDI", GBA_SECONDI}, {"SECONDI1", GBA_SECONDI1}, {"SECONDJ", GBA_SECONDJ},
    {"SECONDJ1", GBA_SECONDJ1} } ;

/* GrB_Type_Code of a built-in type -> gba_type (the codes agree for BOOL .. FP64; the complex types carry
 * GxB codes, Include/GraphBLAS.h:4093-4094) */
static bool shim_type (gba_type *t, int32_t code)
{
    if (code == GrB_UDT_CODE) { (*t) = GBA_UDT ; return (true) ; }          /* goes to gba_mxm_udt (shim_run_udt) */
    if (code == GxB_FC32_CODE) { (*t) = GBA_FC32 ; return (true) ; }
    if (code == GxB_FC64_CODE) { (*t) = GBA_FC64 ; return (true) ; }
    if (code < GrB_BOOL_CODE || code > GrB_FP64_CODE) return (false) ;
    (*t) = (gba_type) code ;
    return (true) ;
}

/* "GrB_PLUS_FP64", "GxB_FIRSTJ_INT64", "GrB_LOR" -> opcode; the types come from the operator itself */
static bool shim_binop (gba_opcode *code, gba_type *xtype, gba_type *ytype, gba_type *ztype, GrB_BinaryOp op)
{
    char name [GxB_MAX_NAME_LEN] ;
    SHIM_OK (GrB_BinaryOp_get_String (op, name, GrB_NAME)) ;
    const char *s = name ;
    if (strncmp (s, "GrB_", 4) == 0 || strncmp (s, "GxB_", 4) == 0) s += 4 ; else return (false) ;
    char base [GxB_MAX_NAME_LEN] ;
    strncpy (base, s, sizeof (base) - 1) ; base [sizeof (base) - 1] = '\0' ;
    char *us = strrchr (base, '_') ;
    if (us != NULL)
    {
        static const char *types [ ] = {"BOOL", "INT8", "UINT8", "INT16", "UINT16", "INT32", "UINT32", "INT64",
            "UINT64", "FP32", "FP64", "FC32", "FC64"} ;
        for (size_t k = 0 ; k < sizeof (types) / sizeof (types [0]) ; k++)
        {
            if (strcmp (us + 1, types [k]) == 0) { (*us) = '\0' ; break ; }
        }
    }
    bool found = false ;
    for (size_t k = 0 ; k < sizeof (shim_ops) / sizeof (shim_ops [0]) ; k++)
    {
        if (strcmp (base, shim_ops [k].name) == 0) { (*code) = shim_ops [k].code ; found = true ; break ; }
    }
    if (!found) return (false) ;
    int32_t xc, yc, zc ;
    SHIM_OK (GrB_BinaryOp_get_INT32 (op, &zc, GrB_OUTP_TYPE_CODE)) ;
    /* positional operators have no input types (GrB_NO_VALUE): their inputs are never read */
    if (GrB_BinaryOp_get_INT32 (op, &xc, GrB_INP0_TYPE_CODE) != GrB_SUCCESS) xc = zc ;
    if (GrB_BinaryOp_get_INT32 (op, &yc, GrB_INP1_TYPE_CODE) != GrB_SUCCESS) yc = zc ;
    return (shim_type (xtype, xc) && shim_type (ytype, yc) && shim_type (ztype, zc)) ;
}

static bool shim_semiring (gba_semiring *s, GrB_Semiring semiring)
{
    GrB_Monoid monoid ; GrB_BinaryOp add, mult ;
    SHIM_OK (GxB_Semiring_add (&monoid, semiring)) ;
    SHIM_OK (GxB_Monoid_operator (&add, monoid)) ;
    SHIM_OK (GxB_Semiring_multiply (&mult, semiring)) ;
    gba_type ax, ay, az ;
    if (!shim_binop (&(s->add), &ax, &ay, &az, add)) return (false) ;
    if (!shim_binop (&(s->mult), &(s->xtype), &(s->ytype), &(s->ztype), mult)) return (false) ;
    return (az == s->ztype) ;
}

static void shim_descriptor (gba_descriptor *d, GrB_Descriptor desc)
{
    memset (d, 0, sizeof (gba_descriptor)) ;
    if (desc == NULL) return ;
    int32_t v ;
    if (GxB_Desc_get_INT32 (desc, GrB_OUTP, &v) == GrB_SUCCESS) d->out = v ;
    if (GxB_Desc_get_INT32 (desc, GrB_MASK, &v) == GrB_SUCCESS) d->mask = v ;
    if (GxB_Desc_get_INT32 (desc, GrB_INP0, &v) == GrB_SUCCESS) d->in0 = v ;
    if (GxB_Desc_get_INT32 (desc, GrB_INP1, &v) == GrB_SUCCESS) d->in1 = v ;
    if (GxB_Desc_get_INT32 (desc, GxB_AxB_METHOD, &v) == GrB_SUCCESS) d->axb = v ;
    if (GxB_Desc_get_INT32 (desc, GxB_SORT, &v) == GrB_SUCCESS) d->do_sort = v ;
}

/* ------------------------------------------------------------------ operands -- */

/* one operand out of its handle: the view handed to the device library, and what packing it back needs */
typedef struct
{
    gba_matrix v ;
    GrB_Matrix A ;              /* NULL for a vector */
    GrB_Vector w ;
    GrB_Index p_size, h_size, i_size, x_size, b_size ;
    int64_t vecp [2] ;          /* the two "pointers" of a sparse vector (a vector is an n-by-1 CSC matrix) */
    bool unpacked ;
    size_t udt_size ;           /* bytes per value when v.type == GBA_UDT */
} shim_operand ;

/* do the handles hold arrays of gba_host_malloc (pinned memory the device can address)? */
static bool shim_host_visible (void)
{
    void *(*lib_malloc) (size_t) = NULL ;
    return (GxB_Global_Option_get_FUNCTION (GxB_MALLOC_FUNCTION, (void **) &lib_malloc) == GrB_SUCCESS
        && lib_malloc == gba_host_malloc) ;
}

static bool shim_unpack_matrix (shim_operand *o, GrB_Matrix A)
{
    memset (o, 0, sizeof (shim_operand)) ;
    o->A = A ;
    gba_matrix *v = &(o->v) ;
    GrB_Index nrows, ncols ; int32_t tcode, sparsity ; GxB_Format_Value fmt ; bool iso = false, jumbled = false ;
    SHIM_OK (GrB_Matrix_nrows (&nrows, A)) ;
    SHIM_OK (GrB_Matrix_ncols (&ncols, A)) ;
    SHIM_OK (GrB_Matrix_get_INT32 (A, &tcode, GrB_EL_TYPE_CODE)) ;
    if (!shim_type (&(v->type), tcode)) return (false) ;
    SHIM_OK (GxB_Matrix_Option_get (A, GxB_SPARSITY_STATUS, &sparsity)) ;
    SHIM_OK (GxB_Matrix_Option_get (A, GxB_FORMAT, &fmt)) ;
    const bool by_row = (fmt == GxB_BY_ROW) ;
    v->is_csc = !by_row ;
    v->vlen = (int64_t) (by_row ? ncols : nrows) ;
    v->vdim = (int64_t) (by_row ? nrows : ncols) ;
    v->nvec = v->vdim ;
    v->mem = GBA_MEM_HOST ;
    v->sparsity = sparsity ;
    GrB_Index nvals = 0, nvec = 0 ;
    GrB_Info info ;
    switch (sparsity)
    {
        case GxB_SPARSE :
            info = by_row
                ? GxB_Matrix_unpack_CSR (A, (GrB_Index **) &(v->p), (GrB_Index **) &(v->i), &(v->x), &(o->p_size), &(o->i_size), &(o->x_size), &iso, &jumbled, NULL)
                : GxB_Matrix_unpack_CSC (A, (GrB_Index **) &(v->p), (GrB_Index **) &(v->i), &(v->x), &(o->p_size), &(o->i_size), &(o->x_size), &iso, &jumbled, NULL) ;
            if (info != GrB_SUCCESS) return (false) ;
            v->nvals = v->p [v->vdim] ;
            break ;
        case GxB_HYPERSPARSE :
            info = by_row
                ? GxB_Matrix_unpack_HyperCSR (A, (GrB_Index **) &(v->p), (GrB_Index **) &(v->h), (GrB_Index **) &(v->i), &(v->x), &(o->p_size), &(o->h_size), &(o->i_size), &(o->x_size), &iso, &nvec, &jumbled, NULL)
                : GxB_Matrix_unpack_HyperCSC (A, (GrB_Index **) &(v->p), (GrB_Index **) &(v->h), (GrB_Index **) &(v->i), &(v->x), &(o->p_size), &(o->h_size), &(o->i_size), &(o->x_size), &iso, &nvec, &jumbled, NULL) ;
            if (info != GrB_SUCCESS) return (false) ;
            v->nvec = (int64_t) nvec ;
            v->nvals = v->p [nvec] ;
            break ;
        case GxB_BITMAP :
            info = by_row
                ? GxB_Matrix_unpack_BitmapR (A, &(v->b), &(v->x), &(o->b_size), &(o->x_size), &iso, &nvals, NULL)
                : GxB_Matrix_unpack_BitmapC (A, &(v->b), &(v->x), &(o->b_size), &(o->x_size), &iso, &nvals, NULL) ;
            if (info != GrB_SUCCESS) return (false) ;
            v->nvals = (int64_t) nvals ;
            break ;
        case GxB_FULL :
            info = by_row
                ? GxB_Matrix_unpack_FullR (A, &(v->x), &(o->x_size), &iso, NULL)
                : GxB_Matrix_unpack_FullC (A, &(v->x), &(o->x_size), &iso, NULL) ;
            if (info != GrB_SUCCESS) return (false) ;
            v->nvals = v->vlen * v->vdim ;
            break ;
        default :
            return (false) ;
    }
    v->iso = iso ; v->jumbled = jumbled ;
    if (shim_host_visible ()) v->mem = GBA_MEM_HOST_VISIBLE ;
    o->unpacked = true ;
    return (true) ;
}

static bool shim_unpack_vector (shim_operand *o, GrB_Vector w)
{
    memset (o, 0, sizeof (shim_operand)) ;
    o->w = w ;
    gba_matrix *v = &(o->v) ;
    GrB_Index n, nvals = 0 ; int32_t tcode, sparsity ; bool iso = false, jumbled = false ;
    SHIM_OK (GrB_Vector_size (&n, w)) ;
    SHIM_OK (GrB_Vector_get_INT32 (w, &tcode, GrB_EL_TYPE_CODE)) ;
    if (!shim_type (&(v->type), tcode)) return (false) ;
    SHIM_OK (GxB_Vector_Option_get (w, GxB_SPARSITY_STATUS, &sparsity)) ;
    v->is_csc = true ; v->vlen = (int64_t) n ; v->vdim = 1 ; v->nvec = 1 ; v->mem = GBA_MEM_HOST ;
    v->sparsity = sparsity ;
    switch (sparsity)
    {
        case GxB_SPARSE :
        case GxB_HYPERSPARSE :                       /* a hypersparse vector unpacks as a sparse one */
            SHIM_OK (GxB_Vector_unpack_CSC (w, (GrB_Index **) &(v->i), &(v->x), &(o->i_size), &(o->x_size), &iso, &nvals, &jumbled, NULL)) ;
            o->vecp [0] = 0 ; o->vecp [1] = (int64_t) nvals ;
            v->p = o->vecp ; v->sparsity = GBA_SPARSE ;
            break ;
        case GxB_BITMAP :
            SHIM_OK (GxB_Vector_unpack_Bitmap (w, &(v->b), &(v->x), &(o->b_size), &(o->x_size), &iso, &nvals, NULL)) ;
            break ;
        case GxB_FULL :
            SHIM_OK (GxB_Vector_unpack_Full (w, &(v->x), &(o->x_size), &iso, NULL)) ;
            nvals = n ;
            break ;
        default :
            return (false) ;
    }
    v->nvals = (int64_t) nvals ;
    v->iso = iso ; v->jumbled = jumbled ;
    /* (the two pointers of a sparse vector live in this struct, in ordinary memory: such a view is copied) */
    if (shim_host_visible () && v->p == NULL) v->mem = GBA_MEM_HOST_VISIBLE ;
    o->unpacked = true ;
    return (true) ;
}

/* put the host arrays named by o->v back into the handle they came from (or, for C, the new ones) */
static GrB_Info shim_pack (shim_operand *o)
{
    if (!o->unpacked) return (GrB_SUCCESS) ;
    gba_matrix *v = &(o->v) ;
    o->unpacked = false ;
    if (o->A != NULL)
    {
        const bool by_row = !v->is_csc ;
        switch (v->sparsity)
        {
            case GBA_SPARSE :
                return (by_row
                    ? GxB_Matrix_pack_CSR (o->A, (GrB_Index **) &(v->p), (GrB_Index **) &(v->i), &(v->x), o->p_size, o->i_size, o->x_size, v->iso, v->jumbled, NULL)
                    : GxB_Matrix_pack_CSC (o->A, (GrB_Index **) &(v->p), (GrB_Index **) &(v->i), &(v->x), o->p_size, o->i_size, o->x_size, v->iso, v->jumbled, NULL)) ;
            case GBA_HYPERSPARSE :
                return (by_row
                    ? GxB_Matrix_pack_HyperCSR (o->A, (GrB_Index **) &(v->p), (GrB_Index **) &(v->h), (GrB_Index **) &(v->i), &(v->x), o->p_size, o->h_size, o->i_size, o->x_size, v->iso, (GrB_Index) v->nvec, v->jumbled, NULL)
                    : GxB_Matrix_pack_HyperCSC (o->A, (GrB_Index **) &(v->p), (GrB_Index **) &(v->h), (GrB_Index **) &(v->i), &(v->x), o->p_size, o->h_size, o->i_size, o->x_size, v->iso, (GrB_Index) v->nvec, v->jumbled, NULL)) ;
            case GBA_BITMAP :
                return (by_row
                    ? GxB_Matrix_pack_BitmapR (o->A, &(v->b), &(v->x), o->b_size, o->x_size, v->iso, (GrB_Index) v->nvals, NULL)
                    : GxB_Matrix_pack_BitmapC (o->A, &(v->b), &(v->x), o->b_size, o->x_size, v->iso, (GrB_Index) v->nvals, NULL)) ;
            default :
                return (by_row
                    ? GxB_Matrix_pack_FullR (o->A, &(v->x), o->x_size, v->iso, NULL)
                    : GxB_Matrix_pack_FullC (o->A, &(v->x), o->x_size, v->iso, NULL)) ;
        }
    }
    switch (v->sparsity)
    {
        case GBA_SPARSE :
            return (GxB_Vector_pack_CSC (o->w, (GrB_Index **) &(v->i), &(v->x), o->i_size, o->x_size, v->iso, (GrB_Index) v->nvals, v->jumbled, NULL)) ;
        case GBA_BITMAP :
            return (GxB_Vector_pack_Bitmap (o->w, &(v->b), &(v->x), o->b_size, o->x_size, v->iso, (GrB_Index) v->nvals, NULL)) ;
        default :
            return (GxB_Vector_pack_Full (o->w, &(v->x), o->x_size, v->iso, NULL)) ;
    }
}

/* --------------------------------------------------------------- the output -- */

static size_t shim_type_size (gba_type t, size_t udt_size)
{
    switch (t)
    {
        case GBA_UDT : return (udt_size) ;
        case GBA_BOOL : case GBA_INT8 : case GBA_UINT8 : return (1) ;
        case GBA_INT16 : case GBA_UINT16 : return (2) ;
        case GBA_INT32 : case GBA_UINT32 : case GBA_FP32 : return (4) ;
        case GBA_FC64 : return (16) ;
        default : return (8) ;
    }
}

/* bytes of each array of a view */
static void shim_sizes (const gba_matrix *v, size_t udt_size, size_t *pb, size_t *hb, size_t *ib, size_t *xb, size_t *bb)
{
    const size_t dense = (size_t) v->vlen * (size_t) v->vdim, ts = shim_type_size (v->type, udt_size) ;
    const bool sparse = (v->sparsity == GBA_SPARSE || v->sparsity == GBA_HYPERSPARSE) ;
    (*pb) = sparse ? (size_t) (v->nvec + 1) * 8 : 0 ;
    (*hb) = (v->sparsity == GBA_HYPERSPARSE) ? (size_t) v->nvec * 8 : 0 ;
    (*ib) = sparse ? (size_t) v->nvals * 8 : 0 ;
    (*bb) = (v->sparsity == GBA_BITMAP) ? dense : 0 ;
    (*xb) = v->iso ? ts : (sparse ? (size_t) v->nvals : dense) * ts ;
}

/* device copy of the host view of C (the library wants its in/out operand on the device) */
static bool shim_upload (gba_matrix *dev, const gba_matrix *host, size_t udt_size)
{
    (*dev) = (*host) ;
    dev->mem = GBA_MEM_DEVICE ;
    dev->p = NULL ; dev->h = NULL ; dev->i = NULL ; dev->x = NULL ; dev->b = NULL ;
    size_t pb, hb, ib, xb, bb ;
    shim_sizes (host, udt_size, &pb, &hb, &ib, &xb, &bb) ;
    const void *src [5] = {host->p, host->h, host->i, host->x, host->b} ;
    void **dst [5] = {(void **) &(dev->p), (void **) &(dev->h), (void **) &(dev->i), &(dev->x), (void **) &(dev->b)} ;
    const size_t nb [5] = {pb, hb, ib, xb, bb} ;
    for (int k = 0 ; k < 5 ; k++)
    {
        if (src [k] == NULL || (nb [k] == 0 && k != 3)) continue ;
        (*dst [k]) = gba_malloc (nb [k] ? nb [k] : 1) ;
        if ((*dst [k]) == NULL || (nb [k] && gba_memcpy_h2d ((*dst [k]), src [k], nb [k]) != GBA_SUCCESS))
        {
            gba_matrix_free_content (dev) ;
            return (false) ;
        }
    }
    return (true) ;
}

/* the result arrays come back to host memory of the library's allocator and replace those of C */
static bool shim_download (shim_operand *c, gba_matrix *dev)
{
    void *(*lib_malloc) (size_t) = NULL ; void (*lib_free) (void *) = NULL ;
    if (GxB_Global_Option_get_FUNCTION (GxB_MALLOC_FUNCTION, (void **) &lib_malloc) != GrB_SUCCESS || lib_malloc == NULL) return (false) ;
    if (GxB_Global_Option_get_FUNCTION (GxB_FREE_FUNCTION, (void **) &lib_free) != GrB_SUCCESS || lib_free == NULL) return (false) ;
    if (dev->nzombies > 0 && gba_matrix_wait (dev) != GBA_SUCCESS) return (false) ;     /* a packed matrix has no zombies */
    gba_matrix h = (*dev) ;
    h.mem = GBA_MEM_HOST ; h.p = NULL ; h.h = NULL ; h.i = NULL ; h.x = NULL ; h.b = NULL ;
    size_t pb, hb, ib, xb, bb ;
    shim_sizes (dev, c->udt_size, &pb, &hb, &ib, &xb, &bb) ;
    const void *src [5] = {dev->p, dev->h, dev->i, dev->x, dev->b} ;
    void **dst [5] = {(void **) &(h.p), (void **) &(h.h), (void **) &(h.i), &(h.x), (void **) &(h.b)} ;
    const size_t need [5] = {pb, hb, ib, xb, bb} ;                          /* bytes to copy */
    size_t nb [5] ;                                                         /* bytes to allocate: pack wants non-NULL arrays */
    const bool hyper = (dev->sparsity == GBA_HYPERSPARSE) ;                 /* a C that came in hypersparse comes back so */
    const bool sparse = (dev->sparsity == GBA_SPARSE) || hyper ;
    const bool wanted [5] = {sparse, hyper, sparse, true, dev->sparsity == GBA_BITMAP} ;
    bool ok = true ;
    for (int k = 0 ; k < 5 ; k++)
    {
        nb [k] = 0 ;
        if (!wanted [k] || !ok) continue ;
        nb [k] = need [k] ? need [k] : 8 ;
        (*dst [k]) = lib_malloc (nb [k]) ;
        ok = ((*dst [k]) != NULL) && (need [k] == 0 || (src [k] != NULL && gba_memcpy_d2h ((*dst [k]), src [k], need [k]) == GBA_SUCCESS)) ;
    }
    if (!ok)
    {
        lib_free (h.p) ; lib_free (h.h) ; lib_free (h.i) ; lib_free (h.x) ; lib_free (h.b) ;
        return (false) ;
    }
    /* the old host arrays of C are not needed any more */
    gba_matrix *old = &(c->v) ;
    if (old->p != c->vecp) lib_free (old->p) ;
    lib_free (old->h) ; lib_free (old->i) ; lib_free (old->x) ; lib_free (old->b) ;
    const bool is_vector = (c->w != NULL) ;
    c->v = h ;
    c->p_size = nb [0] ; c->h_size = nb [1] ; c->i_size = nb [2] ; c->x_size = nb [3] ; c->b_size = nb [4] ;
    if (is_vector && h.sparsity == GBA_SPARSE)
    {
        int64_t nv = h.p [1] ;                                              /* Vector_pack_CSC takes nvals, not p */
        lib_free (h.p) ; c->v.p = c->vecp ; c->v.nvals = nv ;
    }
    return (true) ;
}

/* ------------------------------------------------------------------ the call -- */

/* The operands of one call.  Every DISTINCT handle is moved out of its object once (ops [0 .. nops)); the four
 * roles point into that list, so GrB_mxm (C, L, NULL, sr, L, L, desc) has two operands and four roles. */
typedef struct
{
    shim_operand ops [4] ;
    int nops ;
    shim_operand *c, *m, *a, *b ;
    bool has_mask ;
} shim_call ;

static shim_operand *shim_role_matrix (shim_call *k, GrB_Matrix A)
{
    for (int q = 0 ; q < k->nops ; q++) if (k->ops [q].A == A && A != NULL) return (&(k->ops [q])) ;
    if (!shim_unpack_matrix (&(k->ops [k->nops]), A)) return (NULL) ;       /* (nothing was moved when this fails) */
    return (&(k->ops [k->nops++])) ;
}

static shim_operand *shim_role_vector (shim_call *k, GrB_Vector w)
{
    for (int q = 0 ; q < k->nops ; q++) if (k->ops [q].w == w && w != NULL) return (&(k->ops [q])) ;
    if (!shim_unpack_vector (&(k->ops [k->nops]), w)) return (NULL) ;
    return (&(k->ops [k->nops++])) ;
}

/* every handle gets its arrays back; false when one could not be restored (the object is then empty) */
static bool shim_pack_all (shim_call *k)
{
    bool ok = true ;
    for (int q = 0 ; q < k->nops ; q++) ok = (shim_pack (&(k->ops [q])) == GrB_SUCCESS) && ok ;
    return (ok) ;
}

static pthread_mutex_t shim_lock = PTHREAD_MUTEX_INITIALIZER ;      /* a handle is empty between unpack and pack */

static bool shim_device_ready (void)
{
    static bool inited = false ;
    if (!inited)
    {
        const char *dev = getenv ("GBA_DEVICE") ;
        if (gba_init (dev ? atoi (dev) : 0) != GBA_SUCCESS) return (false) ;
        inited = true ;
    }
    return (true) ;
}

static bool shim_run_udt (int which, shim_call *k, const GrB_BinaryOp accum,
    const GrB_Semiring semiring, const GrB_Descriptor desc) ;

/* true: the result was computed on the device and C holds it.  false: C, A, B and the mask are as they came and the
   caller passes the call on to the CPU library -- also after a device-side failure of any kind (out of memory, a
   compile error, an unsupported case, a dimension mismatch): the CPU library then computes the result or reports the
   error with its own code and message. */
static bool shim_run (int which, shim_call *k, const GrB_BinaryOp accum,
    const GrB_Semiring semiring, const GrB_Descriptor desc)
{
    gba_semiring s ; gba_binop acc ; gba_descriptor d ;
    if (k->c->v.type == GBA_UDT) return (shim_run_udt (which, k, accum, semiring, desc)) ;
    if (semiring == NULL || !shim_semiring (&s, semiring)) return (false) ;
    if (k->a->v.type == GBA_UDT || k->b->v.type == GBA_UDT) return (false) ;     /* a built-in semiring on user-typed operands */
    acc.op = GBA_NOP ; acc.type = s.ztype ;
    if (accum != NULL)
    {
        gba_type ax, ay ;
        if (!shim_binop (&(acc.op), &ax, &ay, &(acc.type), accum)) return (false) ;
    }
    shim_descriptor (&d, desc) ;
    if (k->has_mask && k->m->v.type == GBA_UDT && !(d.mask == GBA_STRUCTURE || d.mask == GBA_COMP_STRUCTURE))
        return (false) ;                                /* a user-typed mask read by value: the CPU library's error */
    if (!shim_device_ready ()) return (false) ;
    gba_matrix cdev ;
    if (!shim_upload (&cdev, &(k->c->v), 0)) return (false) ;      /* (an iso-valued C travels with its one value) */
    if (k->c->A != NULL)
    {
        /* the result is re-conformed to sparse or hypersparse by C's own hyper_switch, as GB_mxm ends (GB_conform) */
        double hs = 0 ;
        if (GxB_Matrix_Option_get (k->c->A, GxB_HYPER_SWITCH, &hs) == GrB_SUCCESS && hs > 0) cdev.hyper_switch = (float) hs ;
    }
    /* the inputs stay host views (an input that is also the output is read as it was on entry); C is cdev */
    const gba_matrix *m = k->has_mask ? &(k->m->v) : NULL ;
    gba_info info ;
    if (which == 0) info = gba_mxm (&cdev, m, accum ? &acc : NULL, &s, &(k->a->v), &(k->b->v), &d, NULL) ;
    else if (which == 1) info = gba_vxm (&cdev, m, accum ? &acc : NULL, &s, &(k->a->v), &(k->b->v), &d, NULL) ;
    else info = gba_mxv (&cdev, m, accum ? &acc : NULL, &s, &(k->a->v), &(k->b->v), &d, NULL) ;
    const bool taken = (info == GBA_SUCCESS) && shim_download (k->c, &cdev) ;
    gba_matrix_free_content (&cdev) ;
    return (taken) ;
}

/* ---------------------------------------------------------- user-defined types -- */

/* The strings the application gave to GxB_Type_new and GxB_BinaryOp_new (Include/GraphBLAS.h:4056-4057,
 * GxB_JIT_C_NAME / GxB_JIT_C_DEFINITION) are what the CPU library's own JIT compiles; gba_mxm_udt compiles the
 * same strings for the device.  An object created without a definition cannot be compiled: the call stays on
 * the CPU. */
static char *shim_type_string (GrB_Type t, GrB_Field f)
{
    size_t n = 0 ;
    if (GrB_Type_get_SIZE (t, &n, f) != GrB_SUCCESS || n <= 1) return (NULL) ;
    char *str = malloc (n) ;
    if (str != NULL && GrB_Type_get_String (t, str, f) != GrB_SUCCESS) { free (str) ; str = NULL ; }
    return (str) ;
}

static char *shim_binop_string (GrB_BinaryOp op, GrB_Field f)
{
    size_t n = 0 ;
    if (GrB_BinaryOp_get_SIZE (op, &n, f) != GrB_SUCCESS || n <= 1) return (NULL) ;
    char *str = malloc (n) ;
    if (str != NULL && GrB_BinaryOp_get_String (op, str, f) != GrB_SUCCESS) { free (str) ; str = NULL ; }
    return (str) ;
}

static bool shim_operand_type (GrB_Type *t, const shim_operand *o)
{
    return ((o->A != NULL ? GxB_Matrix_type (t, o->A) : GxB_Vector_type (t, o->w)) == GrB_SUCCESS) ;
}

static bool shim_binop_is_over (GrB_BinaryOp op, int32_t code)
{
    int32_t xc, yc, zc ;
    return (GrB_BinaryOp_get_INT32 (op, &zc, GrB_OUTP_TYPE_CODE) == GrB_SUCCESS && zc == code
        && GrB_BinaryOp_get_INT32 (op, &xc, GrB_INP0_TYPE_CODE) == GrB_SUCCESS && xc == code
        && GrB_BinaryOp_get_INT32 (op, &yc, GrB_INP1_TYPE_CODE) == GrB_SUCCESS && yc == code) ;
}

/* C<M> = accum (C, A*B) (which == 0), w'<m'> = accum (w', u'*A) (which == 1: gba_vxm_udt exchanges the operands of
 * the multiplier) or w<m> = accum (w, A*u) (which == 2) over a semiring of one user-defined type.  Same contract as
 * shim_run. */
static bool shim_run_udt (int which, shim_call *k, const GrB_BinaryOp accum,
    const GrB_Semiring semiring, const GrB_Descriptor desc)
{
    if (semiring == NULL) return (false) ;
    if (k->c->v.type != GBA_UDT || k->a->v.type != GBA_UDT || k->b->v.type != GBA_UDT || k->c->v.iso) return (false) ;
    GrB_Type tc, ta, tb ;
    if (!shim_operand_type (&tc, k->c) || !shim_operand_type (&ta, k->a) || !shim_operand_type (&tb, k->b)
        || ta != tc || tb != tc) return (false) ;
    gba_descriptor d ;
    shim_descriptor (&d, desc) ;
    const bool structural = (d.mask == GBA_STRUCTURE || d.mask == GBA_COMP_STRUCTURE) ;
    if (k->has_mask && k->m->v.type == GBA_UDT && !structural) return (false) ;
    GrB_Monoid monoid ; GrB_BinaryOp add, mult ;
    SHIM_OK (GxB_Semiring_add (&monoid, semiring)) ;
    SHIM_OK (GxB_Monoid_operator (&add, monoid)) ;
    SHIM_OK (GxB_Semiring_multiply (&mult, semiring)) ;
    if (!shim_binop_is_over (add, GrB_UDT_CODE) || !shim_binop_is_over (mult, GrB_UDT_CODE)) return (false) ;
    if (accum != NULL && !shim_binop_is_over (accum, GrB_UDT_CODE)) return (false) ;
    size_t tsize = 0 ;
    if (GxB_Type_size (&tsize, tc) != GrB_SUCCESS || tsize == 0) return (false) ;
    char *str [8] = { shim_type_string (tc, GxB_JIT_C_NAME), shim_type_string (tc, GxB_JIT_C_DEFINITION),
        shim_binop_string (add, GxB_JIT_C_NAME), shim_binop_string (add, GxB_JIT_C_DEFINITION),
        shim_binop_string (mult, GxB_JIT_C_NAME), shim_binop_string (mult, GxB_JIT_C_DEFINITION),
        accum ? shim_binop_string (accum, GxB_JIT_C_NAME) : NULL, accum ? shim_binop_string (accum, GxB_JIT_C_DEFINITION) : NULL } ;
    bool taken = true ;
    for (int q = 0 ; q < (accum ? 8 : 6) ; q++) taken = taken && (str [q] != NULL) ;
    /* device copies, one per distinct operand (the compiled kernels want device views); C gets one of its own */
    gba_matrix dev [4], cdev ;
    memset (dev, 0, sizeof (dev)) ; memset (&cdev, 0, sizeof (cdev)) ;
    bool up [4] = {false, false, false, false} ;
    if (taken && !shim_device_ready ()) taken = false ;
    if (taken)
    {
        k->c->udt_size = tsize ;
        taken = shim_upload (&cdev, &(k->c->v), tsize) ;
        const shim_operand *roles [3] = { k->a, k->b, k->has_mask ? k->m : NULL } ;
        for (int r = 0 ; r < 3 && taken ; r++)
        {
            if (roles [r] == NULL) continue ;
            const int q = (int) (roles [r] - k->ops) ;
            if (!up [q]) { taken = shim_upload (&(dev [q]), &(roles [r]->v), tsize) ; up [q] = taken ; }
        }
    }
    if (taken)
    {
        gba_user_semiring us ;
        us.type_name = str [0] ; us.type_defn = str [1] ; us.type_size = tsize ;
        us.add_name = str [2] ; us.add_defn = str [3] ; us.mult_name = str [4] ; us.mult_defn = str [5] ;
        const gba_matrix *dm = k->has_mask ? &(dev [k->m - k->ops]) : NULL ;
        const gba_info info = (which == 1)
            ? gba_vxm_udt (&cdev, dm, str [6], str [7], &us, &(dev [k->a - k->ops]), &(dev [k->b - k->ops]), &d, NULL)
            : gba_mxm_udt (&cdev, dm, str [6], str [7], &us, &(dev [k->a - k->ops]), &(dev [k->b - k->ops]), &d, NULL) ;
        taken = (info == GBA_SUCCESS) && shim_download (k->c, &cdev) ;
    }
    gba_matrix_free_content (&cdev) ;
    for (int q = 0 ; q < 4 ; q++) if (up [q]) gba_matrix_free_content (&(dev [q])) ;
    for (int q = 0 ; q < 8 ; q++) free (str [q]) ;
    return (taken) ;
}

/* after the multiply: everything goes back into its handle.  A handle that cannot be restored is a lost object:
   the call then reports GrB_PANIC instead of going on with an empty operand. */
#define SHIM_FINISH(call_next)                                              \
    const bool packed = shim_pack_all (&k) ;                                \
    pthread_mutex_unlock (&shim_lock) ;                                     \
    if (!packed) return (GrB_PANIC) ;                                       \
    if (ok) return (GrB_SUCCESS) ;                                          \
    return (next ? (call_next) : GrB_PANIC) ;

GrB_Info GrB_mxm (GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_Semiring semiring,
    const GrB_Matrix A, const GrB_Matrix B, const GrB_Descriptor desc)
{
    static shim_mxm_f next = NULL ;
    if (next == NULL) next = (shim_mxm_f) dlsym (RTLD_NEXT, "GrB_mxm") ;
    (void) shim_check_mxm ;
    if (C == NULL || A == NULL || B == NULL) return (next ? next (C, Mask, accum, semiring, A, B, desc) : GrB_NULL_POINTER) ;
    shim_call k ; memset (&k, 0, sizeof (k)) ;
    pthread_mutex_lock (&shim_lock) ;
    k.has_mask = (Mask != NULL) ;
    /* C first: when it is also an input, that input reads the view made here */
    bool ok = ((k.c = shim_role_matrix (&k, C)) != NULL) && ((k.a = shim_role_matrix (&k, A)) != NULL)
        && ((k.b = shim_role_matrix (&k, B)) != NULL) && (!k.has_mask || (k.m = shim_role_matrix (&k, Mask)) != NULL) ;
    if (ok) ok = shim_run (0, &k, accum, semiring, desc) ;
    SHIM_FINISH (next (C, Mask, accum, semiring, A, B, desc))
}

GrB_Info GrB_vxm (GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Semiring semiring,
    const GrB_Vector u, const GrB_Matrix A, const GrB_Descriptor desc)
{
    static shim_vxm_f next = NULL ;
    if (next == NULL) next = (shim_vxm_f) dlsym (RTLD_NEXT, "GrB_vxm") ;
    (void) shim_check_vxm ;
    if (w == NULL || u == NULL || A == NULL) return (next ? next (w, mask, accum, semiring, u, A, desc) : GrB_NULL_POINTER) ;
    shim_call k ; memset (&k, 0, sizeof (k)) ;
    pthread_mutex_lock (&shim_lock) ;
    k.has_mask = (mask != NULL) ;
    bool ok = ((k.c = shim_role_vector (&k, w)) != NULL) && ((k.a = shim_role_vector (&k, u)) != NULL)
        && ((k.b = shim_role_matrix (&k, A)) != NULL) && (!k.has_mask || (k.m = shim_role_vector (&k, mask)) != NULL) ;
    if (ok) ok = shim_run (1, &k, accum, semiring, desc) ;
    SHIM_FINISH (next (w, mask, accum, semiring, u, A, desc))
}

GrB_Info GrB_mxv (GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Semiring semiring,
    const GrB_Matrix A, const GrB_Vector u, const GrB_Descriptor desc)
{
    static shim_mxv_f next = NULL ;
    if (next == NULL) next = (shim_mxv_f) dlsym (RTLD_NEXT, "GrB_mxv") ;
    (void) shim_check_mxv ;
    if (w == NULL || u == NULL || A == NULL) return (next ? next (w, mask, accum, semiring, A, u, desc) : GrB_NULL_POINTER) ;
    shim_call k ; memset (&k, 0, sizeof (k)) ;
    pthread_mutex_lock (&shim_lock) ;
    k.has_mask = (mask != NULL) ;
    bool ok = ((k.c = shim_role_vector (&k, w)) != NULL) && ((k.a = shim_role_matrix (&k, A)) != NULL)
        && ((k.b = shim_role_vector (&k, u)) != NULL) && (!k.has_mask || (k.m = shim_role_vector (&k, mask)) != NULL) ;
    if (ok) ok = shim_run (2, &k, accum, semiring, desc) ;
    SHIM_FINISH (next (w, mask, accum, semiring, A, u, desc))
}
