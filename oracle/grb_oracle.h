/* grb_oracle.h -- TEST INFRASTRUCTURE ONLY: CPU restatement of the reference's
 * sparse multiply path (see grb_oracle.c).  Not part of the product. */
#ifndef GRB_ORACLE_H
#define GRB_ORACLE_H
#include <stdint.h>
#include <stdbool.h>
#include <stddef.h>
#include "../include/graphblas_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef void (*orc_binfn)(void *z, const void *x, const void *y);

/* working semiring: built-in (function pointers looked up) or user-defined */
typedef struct orc_semiring {
    orc_binfn fadd, fmult;
    int add_op, mult_op;
    int positional;            /* canonical positional multiplier opcode or 0 */
    int xtype, ytype, ztype;
    size_t xsize, ysize, zsize;
    bool a_pattern, b_pattern;
} orc_semiring;

size_t orc_type_size(int code);
int  orc_semiring_builtin(orc_semiring *s, const gba_semiring *in, bool flipxy);
void orc_semiring_user(orc_semiring *s, orc_binfn fadd, orc_binfn fmult, size_t xsize, size_t ysize, size_t zsize);

void orc_free(void *p);
void orc_matrix_free_content(gba_matrix *A);
void orc_cumsum(int64_t *count, int64_t n);
int  orc_num_threads(void);

int orc_saxpy3_flopcount(int64_t *Mwork, int64_t *Bflops, const gba_matrix *M, bool Mask_comp,
                         const gba_matrix *A, const gba_matrix *B);
int orc_saxpy3(gba_matrix *C, const gba_matrix *M, bool Mask_comp, bool Mask_struct,
               const gba_matrix *A, const gba_matrix *B, const orc_semiring *s);
int orc_dot3(gba_matrix *C, const gba_matrix *M, bool Mask_struct, const gba_matrix *A, const gba_matrix *B,
             const orc_semiring *s);
int orc_dot2(gba_matrix *C, const gba_matrix *M, bool Mask_comp, bool Mask_struct, const gba_matrix *A,
             const gba_matrix *B, const orc_semiring *s);
int orc_dot4(gba_matrix *C, const gba_matrix *A, const gba_matrix *B, const orc_semiring *s);
int orc_saxpy4(gba_matrix *C, const gba_matrix *A, const gba_matrix *B, const orc_semiring *s);
int orc_saxpy5(gba_matrix *C, const gba_matrix *A, const gba_matrix *B, const orc_semiring *s);
int orc_saxbit(gba_matrix *C, const gba_matrix *M, bool Mask_comp, bool Mask_struct, const gba_matrix *A,
               const gba_matrix *B, const orc_semiring *s);
int orc_mxm_spec(int8_t *present, void *value, int64_t nrows, int64_t ncols, int64_t inner,
                 const gba_matrix *C, const gba_matrix *M, const gba_binop *accum, const gba_semiring *sr,
                 const gba_matrix *A, bool A_tran, const gba_matrix *B, bool B_tran, bool flipxy,
                 bool C_replace, bool Mask_comp, bool Mask_struct);
int orc_matrix_wait(gba_matrix *C);

/* replay of the GrB_mxm sequence of Demo/Program/gauss_demo.c:377-471 with the
 * oracle's methods; prints the matrices in the demo's own format */
int orc_gauss_replay(char *out, size_t out_size);
int orc_wildtype_replay(char *out, size_t out_size);

#ifdef __cplusplus
}
#endif
#endif
