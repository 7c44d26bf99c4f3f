/*
 * wildtype_replay.c -- TEST INFRASTRUCTURE ONLY.  Replays, with the oracle's
 * method restatements, the two GrB_mxm calls of the reference's known-answer
 * demo Demo/Program/wildtype_demo.c (a 192-byte user-defined type: a 4-by-4
 * block of doubles and 64 characters; semiring (wildadd, wildmult), lines
 * 112-190 of that file):
 *   C = A*B                         saxpy3          (wildtype_demo.c:366)
 *   C<M,struct,replace> = C*C'      dot3            (wildtype_demo.c:381, GrB_DESC_RST1)
 * The inputs are built as the program builds them (wildtype_demo.c:256-336);
 * the expected entries are in Demo/Output/wildtype_demo.out:142-330.  All demo
 * matrices are held by row, so both calls reach the method drivers with the
 * operands swapped and the multiplier flipped (Source/GB_AxB_meta.c:277-296,
 * 428-439), as in gauss_replay.c.
 */
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include "grb_oracle.h"

typedef struct { double stuff[4][4]; char whatstuff[64]; } wildtype;

static void wildadd(void *z_, const void *x_, const void *y_) {
    const wildtype *x = x_, *y = y_;
    wildtype z; memset(&z, 0, sizeof(z));
    memcpy(z.whatstuff, ((const wildtype *)z_)->whatstuff, 64);       /* the operator leaves the tail of z's text alone */
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) z.stuff[i][j] = x->stuff[i][j] + y->stuff[i][j];
    const char *psrc = "this was added"; char *pdst = z.whatstuff;
    while ((*pdst++ = *psrc++)) ;
    *(wildtype *)z_ = z;
}
static void wildmult(void *z_, const void *x_, const void *y_) {
    const wildtype *x = x_, *y = y_;
    wildtype z; memset(&z, 0, sizeof(z));
    memcpy(z.whatstuff, ((const wildtype *)z_)->whatstuff, 64);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
        z.stuff[i][j] = 0;
        for (int k = 0; k < 4; k++) z.stuff[i][j] += x->stuff[i][k] * y->stuff[k][j];
    }
    const char *psrc = "this was multiplied"; char *pdst = z.whatstuff;
    while ((*pdst++ = *psrc++)) ;
    *(wildtype *)z_ = z;
}
static void wildmult_flipped(void *z, const void *x, const void *y) { wildmult(z, y, x); }

#define N 10

/* a 10x10 matrix held by row from (row, column, value) triples sorted by row then column */
static void make_rows(gba_matrix *A, int nz, const int *r, const int *c, const void *x, size_t size, int type) {
    memset(A, 0, sizeof(*A));
    A->vlen = N; A->vdim = N; A->nvec = N; A->nvals = nz; A->type = type; A->sparsity = GBA_SPARSE; A->mem = GBA_MEM_HOST; A->is_csc = false;
    A->p = calloc(N + 1, sizeof(int64_t)); A->i = malloc((nz ? nz : 1) * sizeof(int64_t)); A->x = malloc((nz ? nz : 1) * size);
    for (int k = 0; k < nz; k++) { A->p[r[k] + 1]++; A->i[k] = c[k]; }
    for (int j = 0; j < N; j++) A->p[j + 1] += A->p[j];
    memcpy(A->x, x, nz * size);
}
/* sort every vector by index (saxpy3 hands its result over jumbled); the matrices here hold a few entries */
static void sort_vectors(gba_matrix *A) {
    wildtype *x = (wildtype *)A->x;
    for (int j = 0; j < N; j++)
        for (int64_t p = A->p[j] + 1; p < A->p[j + 1]; p++)
            for (int64_t q = p; q > A->p[j] && A->i[q - 1] > A->i[q]; q--) {
                const int64_t ti = A->i[q]; A->i[q] = A->i[q - 1]; A->i[q - 1] = ti;
                const wildtype tx = x[q]; x[q] = x[q - 1]; x[q - 1] = tx;
            }
    A->jumbled = false;
}
static size_t print_matrix(char *out, size_t cap, size_t pos, const char *name, const gba_matrix *A) {
    pos += (size_t)snprintf(out + pos, pos < cap ? cap - pos : 0, "== %s\n", name);
    for (int j = 0; j < N; j++) for (int64_t p = A->p[j]; p < A->p[j + 1]; p++) {
        if (A->i[p] < 0) continue;                                   /* a zombie of dot3 */
        const wildtype *w = (const wildtype *)A->x + p;
        pos += (size_t)snprintf(out + pos, pos < cap ? cap - pos : 0, "(%d,%d) [%.63s]", j, (int)A->i[p], w->whatstuff);
        for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++)
            pos += (size_t)snprintf(out + pos, pos < cap ? cap - pos : 0, " %.17g", w->stuff[a][b]);
        pos += (size_t)snprintf(out + pos, pos < cap ? cap - pos : 0, "\n");
    }
    return pos;
}

int orc_wildtype_replay(char *out, size_t cap) {
    size_t pos = 0;
    orc_semiring s;
    orc_semiring_user(&s, wildadd, wildmult_flipped, sizeof(wildtype), sizeof(wildtype), sizeof(wildtype));
    /* A(2,7), A(3,7), A(2,4): wildtype_demo.c:258-283 */
    wildtype s1; memset(&s1, 0, sizeof(s1));
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) s1.stuff[i][j] = 100 * i + j;
    wildtype a27 = s1; strcpy(a27.whatstuff, "this is A(2,7)");
    s1.stuff[2][3] = 909;
    wildtype a37 = s1; strcpy(a37.whatstuff, "this is A(3,7)");
    s1.stuff[3][3] = 42;
    wildtype a24 = s1; strcpy(a24.whatstuff, "this is A(2,4)");
    /* B(7,2), B(7,5), B(4,2): wildtype_demo.c:314-336 */
    wildtype s2; memset(&s2, 0, sizeof(s2));
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) s2.stuff[i][j] = (double)(j - i) + 0.5;
    wildtype b72 = s2; strcpy(b72.whatstuff, "this is B(7,2)");
    s2.stuff[0][0] = -1;
    wildtype b75 = s2; strcpy(b75.whatstuff, "here is B(7,5)");
    s2.stuff[0][3] = 77;
    wildtype b42 = s2; strcpy(b42.whatstuff, "finally, B(4,2)");

    gba_matrix A, B, C, C2, M;
    { const int r[] = {2, 2, 3}, c[] = {4, 7, 7}; wildtype x[3]; x[0] = a24; x[1] = a27; x[2] = a37; make_rows(&A, 3, r, c, x, sizeof(wildtype), GBA_UDT); }
    { const int r[] = {4, 7, 7}, c[] = {2, 2, 5}; wildtype x[3]; x[0] = b42; x[1] = b72; x[2] = b75; make_rows(&B, 3, r, c, x, sizeof(wildtype), GBA_UDT); }
    pos = print_matrix(out, cap, pos, "input A", &A);
    pos = print_matrix(out, cap, pos, "input B", &B);
    /* C = A*B: internally C_I = A_I (user B) * B_I (user A) with the flipped multiplier */
    if (orc_saxpy3(&C, NULL, false, false, &B, &A, &s)) return -1;
    sort_vectors(&C);
    pos = print_matrix(out, cap, pos, "C = A*B", &C);
    /* C<M,struct,replace> = C*C' (GrB_DESC_RST1): dot3 over the rows of C */
    { const int r[] = {2, 2, 3, 3}, c[] = {2, 3, 2, 3}; const uint8_t x[4] = {1, 1, 1, 1}; make_rows(&M, 4, r, c, x, 1, GBA_BOOL); }
    if (orc_dot3(&C2, &M, true, &C, &C, &s)) return -2;
    pos = print_matrix(out, cap, pos, "C<M,struct,replace> = C*C'", &C2);
    orc_matrix_free_content(&A); orc_matrix_free_content(&B); orc_matrix_free_content(&C);
    orc_matrix_free_content(&C2); orc_matrix_free_content(&M);
    return (pos < cap) ? 0 : -8;
}
