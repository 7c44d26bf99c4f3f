/*
 * gauss_replay.c -- TEST INFRASTRUCTURE ONLY.  Replays, with the oracle's
 * method restatements, the seven GrB_mxm calls of the reference's known-answer
 * demo Demo/Program/gauss_demo.c:377-471 (user-defined Gaussian-integer type,
 * semiring (addgauss, multgauss)).  The expected matrices and the method each
 * call takes are recorded in Demo/Output/gauss_demo.out:223-405:
 *   A = A*A              bitmap saxpy (saxbit)
 *   C<D> = A*A'          masked dot product (dot3)
 *   C = D*A              colscale
 *   C = A*D              rowscale
 *   C += A*A'            dot4, C full in place
 *   C += B*A             saxpy4
 *   C += A*B             saxpy5
 * All demo matrices are held by row, so every call reaches the method drivers
 * with the operands swapped and the multiplier flipped (swap rule,
 * Source/GB_AxB_meta.c:277-296, 428-439); the replay does the same: the
 * internal vectors are the user's rows and fmult_flipped(z,x,y) = mult(y,x).
 */
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include "grb_oracle.h"

typedef struct { int32_t real, imag; } gauss;

static void addgauss(void *z_, const void *x_, const void *y_) {
    const gauss x = *(const gauss *)x_, y = *(const gauss *)y_;
    gauss z; z.real = x.real + y.real; z.imag = x.imag + y.imag; *(gauss *)z_ = z;
}
static void multgauss(void *z_, const void *x_, const void *y_) {
    const gauss x = *(const gauss *)x_, y = *(const gauss *)y_;
    gauss z; z.real = x.real * y.real - x.imag * y.imag; z.imag = x.real * y.imag + x.imag * y.real; *(gauss *)z_ = z;
}
static void multgauss_flipped(void *z, const void *x, const void *y) { multgauss(z, y, x); }

#define N 4

/* a 4x4 matrix held by row: vector j = row j, index i = column */
static void make_bitmap(gba_matrix *A, const gauss *x, const int8_t *b, int64_t nvals) {
    memset(A, 0, sizeof(*A));
    A->vlen = N; A->vdim = N; A->nvec = N; A->nvals = nvals; A->type = GBA_UDT; A->sparsity = b ? GBA_BITMAP : GBA_FULL;
    A->mem = GBA_MEM_HOST; A->is_csc = false;
    A->x = malloc(N * N * sizeof(gauss)); memcpy(A->x, x, N * N * sizeof(gauss));
    if (b) { A->b = malloc(N * N); memcpy(A->b, b, N * N); }
}
static void make_sparse(gba_matrix *A, const gauss *x, const int8_t *b) {
    memset(A, 0, sizeof(*A));
    A->vlen = N; A->vdim = N; A->nvec = N; A->type = GBA_UDT; A->sparsity = GBA_SPARSE; A->mem = GBA_MEM_HOST; A->is_csc = false;
    A->p = malloc((N + 1) * sizeof(int64_t)); A->i = malloc(N * N * sizeof(int64_t)); A->x = malloc(N * N * sizeof(gauss));
    int64_t nz = 0;
    for (int j = 0; j < N; j++) {
        A->p[j] = nz;
        for (int i = 0; i < N; i++) if (!b || b[j * N + i]) { A->i[nz] = i; ((gauss *)A->x)[nz] = x[j * N + i]; nz++; }
    }
    A->p[N] = nz; A->nvals = nz;
}
/* row-major dense image (value, present) of a by-row matrix of any format */
static void to_dense(const gba_matrix *A, gauss *x, int8_t *b) {
    memset(b, 0, N * N); memset(x, 0, N * N * sizeof(gauss));
    if (A->sparsity == GBA_SPARSE) {
        for (int j = 0; j < N; j++) for (int64_t p = A->p[j]; p < A->p[j + 1]; p++) {
            if (A->i[p] < 0) continue;
            b[j * N + A->i[p]] = 1; x[j * N + A->i[p]] = ((gauss *)A->x)[p];
        }
    } else {
        for (int q = 0; q < N * N; q++) if (A->sparsity == GBA_FULL || A->b[q]) { b[q] = 1; x[q] = ((gauss *)A->x)[q]; }
    }
}
static size_t print_gauss(char *out, size_t cap, size_t pos, const char *name, const gauss *x, const int8_t *b) {
    pos += (size_t)snprintf(out + pos, pos < cap ? cap - pos : 0, "%s\n", name);
    for (int r = 0; r < N; r++) {
        pos += (size_t)snprintf(out + pos, pos < cap ? cap - pos : 0, "row %2d: ", r);
        for (int c = 0; c < N; c++) {
            if (b[r * N + c]) pos += (size_t)snprintf(out + pos, pos < cap ? cap - pos : 0, " (%4d,%4d)", x[r * N + c].real, x[r * N + c].imag);
            else pos += (size_t)snprintf(out + pos, pos < cap ? cap - pos : 0, "      .     ");
        }
        pos += (size_t)snprintf(out + pos, pos < cap ? cap - pos : 0, "\n");
    }
    return pos;
}

int orc_gauss_replay(char *out, size_t cap) {
    size_t pos = 0;
    orc_semiring s;
    orc_semiring_user(&s, addgauss, multgauss_flipped, sizeof(gauss), sizeof(gauss), sizeof(gauss));
    gauss ax[N * N], cx[N * N], dx[N * N]; int8_t ab[N * N], cb[N * N], db[N * N];
    /* A(i,j) = (i+1, 2-j), A(0,0) missing (gauss_demo.c:377-393) */
    for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) { ax[i * N + j].real = i + 1; ax[i * N + j].imag = 2 - j; ab[i * N + j] = !(i == 0 && j == 0); }
    gba_matrix A, C, D, T;
    /* A = A*A: both bitmap -> saxbit; internal C = A_I * B_I with A_I = user's second operand */
    make_bitmap(&A, ax, ab, 15);
    if (orc_saxbit(&T, NULL, false, false, &A, &A, &s)) return -1;
    to_dense(&T, ax, ab); orc_matrix_free_content(&T); orc_matrix_free_content(&A);
    pos = print_gauss(out, cap, pos, "A = A^2", ax, ab);
    /* C<D> = A*A' (GrB_DESC_T1), A and D sparse: dot3 */
    make_sparse(&A, ax, ab);
    for (int q = 0; q < N * N; q++) { db[q] = (q / N == q % N); dx[q].real = 1; dx[q].imag = 0; }
    make_sparse(&D, dx, db); D.type = GBA_BOOL;     /* structural use only: values are all true */
    { uint8_t *bx = calloc(N, 1); memset(bx, 1, N); free(D.x); D.x = bx; }
    if (orc_dot3(&T, &D, false, &A, &A, &s)) return -2;
    to_dense(&T, cx, cb); orc_matrix_free_content(&T); orc_matrix_free_content(&D);
    pos = print_gauss(out, cap, pos, "C = diag(AA')", cx, cb);
    /* D = diag(A); C = D*A: colscale, internally C_I = A_I(user A) * B_I(user D) */
    for (int q = 0; q < N * N; q++) { db[q] = (q / N == q % N); dx[q] = ax[q]; }
    make_sparse(&D, dx, db);
    if (orc_saxpy3(&T, NULL, false, false, &A, &D, &s)) return -3;
    to_dense(&T, cx, cb); orc_matrix_free_content(&T);
    pos = print_gauss(out, cap, pos, "C = D*A", cx, cb);
    /* C = A*D: rowscale, internally C_I = A_I(user D) * B_I(user A) */
    if (orc_saxpy3(&T, NULL, false, false, &D, &A, &s)) return -4;
    to_dense(&T, cx, cb); orc_matrix_free_content(&T); orc_matrix_free_content(&D);
    pos = print_gauss(out, cap, pos, "C = A*D", cx, cb);
    /* C = (1,-2) full; C += A*A' (T1): dot4 in place */
    for (int q = 0; q < N * N; q++) { cx[q].real = 1; cx[q].imag = -2; }
    make_bitmap(&C, cx, NULL, N * N);
    if (orc_dot4(&C, &A, &A, &s)) return -5;
    to_dense(&C, cx, cb);
    pos = print_gauss(out, cap, pos, "C += A*A'", cx, cb);
    /* B = (1,-2) full; C += B*A: saxpy4 with A_I = user A (sparse), B_I = user B (full) */
    gauss bx[N * N]; for (int q = 0; q < N * N; q++) { bx[q].real = 1; bx[q].imag = -2; }
    gba_matrix B; make_bitmap(&B, bx, NULL, N * N);
    if (orc_saxpy4(&C, &A, &B, &s)) return -6;
    to_dense(&C, cx, cb);
    pos = print_gauss(out, cap, pos, "C += B*A", cx, cb);
    /* C += A*B: saxpy5 with A_I = user B (full), B_I = user A (sparse) */
    if (orc_saxpy5(&C, &B, &A, &s)) return -7;
    to_dense(&C, cx, cb);
    pos = print_gauss(out, cap, pos, "C += A*B", cx, cb);
    orc_matrix_free_content(&A); orc_matrix_free_content(&B); orc_matrix_free_content(&C);
    return (pos < cap) ? 0 : -8;
}
