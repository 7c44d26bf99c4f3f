// User-defined types and operators: C<M> = accum (C, A*B) over a semiring given as C source, the way
// GxB_Type_new / GxB_BinaryOp_new hand their definitions to the reference's JIT
// (Include/GraphBLAS.h: GxB_Type_new, GxB_BinaryOp_new; Source/GB_jitifyer.c compiles them with the C
// compiler and dlopens the result).  Here hipRTC compiles them for the device that is current and the
// code object is kept per source text.
//
// The pattern of T = A*B does not depend on the types, so it comes from the existing pattern-only
// kernels (saxpy3 with ANY_PAIR).  The values are then one dot product per entry of T -- the rows of A
// against the columns of B, both sorted, merged by one thread in increasing k, with the user's multiply
// and add called in the order the CPU's dot kernels use (Source/Template/GB_AxB_dot_cij.c).  No atomics
// are needed, so any type size works.
#include "gba_internal.hpp"
#include <sys/stat.h>
#include <unistd.h>
#include <cstdio>
#include <cstring>
#include <hip/hiprtc.h>
#include <map>
#include <mutex>
#include <string>

namespace gba {

struct JitKernels { hipModule_t module = nullptr; hipFunction_t dot = nullptr, accum = nullptr, merge = nullptr; };

static std::mutex g_jit_mu;
static std::map<std::string, JitKernels> g_jit_cache;

#define GBA_RTC(call) do { hiprtcResult r_ = (call); if (r_ != HIPRTC_SUCCESS) \
    GBA_FAIL(GBA_PANIC, "%s -> %s", #call, hiprtcGetErrorString(r_)); } while (0)

// gba_finalize: the compiled modules go with the device context they were loaded into
void jit_unload_all() {
    std::lock_guard<std::mutex> lock(g_jit_mu);
    for (auto &e : g_jit_cache) hipModuleUnload(e.second.module);
    g_jit_cache.clear();
}

// ---- code objects kept on disk ----------------------------------------------------------------------------------
// Folder: $GBA_JIT_CACHE, else $HOME/.cache/graphblas_amd; GBA_JIT_CACHE set to an empty string or to 0 turns the disk
// cache off.  File: <128-bit FNV-1a of the identity>.co = magic, length and text of the identity, length and bytes of
// the code object.  A file whose identity text differs (a hash collision, another version's layout) is not used.  Any
// failure to read or write only means that the kernel is compiled.
static unsigned long long g_jit_compiles = 0, g_jit_disk_hits = 0;
static const char JIT_FILE_MAGIC[8] = {'g', 'b', 'a', 'j', 'i', 't', '0', '1'};

static std::string jit_disk_folder() {
    const char *e = getenv("GBA_JIT_CACHE");
    if (e) return (e[0] == '\0' || std::string(e) == "0") ? std::string() : std::string(e);
    const char *home = getenv("HOME");
    return (home && home[0]) ? std::string(home) + "/.cache/graphblas_amd" : std::string();
}

static std::string jit_disk_path(const std::string &identity) {
    const std::string folder = jit_disk_folder();
    if (folder.empty()) return std::string();
    unsigned long long h[2] = {0xcbf29ce484222325ull, 0x84222325cbf29ce4ull};
    for (unsigned char ch : identity) {
        h[0] = (h[0] ^ ch) * 0x100000001b3ull;
        h[1] = (h[1] ^ (unsigned char)(ch + 0x5b)) * 0x100000001b3ull;
    }
    char name[40];
    snprintf(name, sizeof(name), "%016llx%016llx.co", h[0], h[1]);
    return folder + "/" + name;
}

static bool jit_disk_load(const std::string &identity, std::string *code) {
    const std::string path = jit_disk_path(identity);
    if (path.empty()) return false;
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    bool ok = false;
    char magic[8]; unsigned long long n = 0;
    if (fread(magic, 1, 8, f) == 8 && memcmp(magic, JIT_FILE_MAGIC, 8) == 0 && fread(&n, 8, 1, f) == 1 && n == identity.size()) {
        std::string text(n, '\0');
        if (fread(&text[0], 1, n, f) == n && text == identity && fread(&n, 8, 1, f) == 1 && n > 0 && n < (1ull << 30)) {
            code->assign(n, '\0');
            ok = fread(&(*code)[0], 1, n, f) == n;
        }
    }
    fclose(f);
    return ok;
}

static void jit_disk_store(const std::string &identity, const std::string &code) {
    const std::string path = jit_disk_path(identity);
    if (path.empty()) return;
    const std::string folder = jit_disk_folder();
    for (size_t at = 1; at <= folder.size(); at++)                       // mkdir -p
        if (at == folder.size() || folder[at] == '/') (void)mkdir(folder.substr(0, at).c_str(), 0755);
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return;
    unsigned long long n1 = identity.size(), n2 = code.size();
    const bool ok = fwrite(JIT_FILE_MAGIC, 1, 8, f) == 8 && fwrite(&n1, 8, 1, f) == 1 && fwrite(identity.data(), 1, n1, f) == n1
                 && fwrite(&n2, 8, 1, f) == 1 && fwrite(code.data(), 1, n2, f) == n2;
    if (fclose(f) != 0 || !ok || rename(tmp.c_str(), path.c_str()) != 0) (void)remove(tmp.c_str());   // whole file or none
}

static std::string device_function(const char *defn) {
    // "void f (T *z, const T *x, const T *y) { ... }"  ->  a __device__ function
    return std::string("__device__ inline ") + defn + "\n";
}

static gba_info jit_build(const gba_user_semiring *s, const char *accum_name, const char *accum_defn, bool flip_mult, JitKernels *out) {
    std::string src;
    src += "typedef signed char int8_t; typedef unsigned char uint8_t; typedef short int16_t; typedef unsigned short uint16_t;\n"
           "typedef int int32_t; typedef unsigned int uint32_t; typedef long long int64_t; typedef unsigned long long uint64_t;\n";
    src += std::string(s->type_defn) + "\n";
    src += device_function(s->add_defn);
    if (std::string(s->mult_defn) != s->add_defn) src += device_function(s->mult_defn);
    const bool own_accum = accum_defn && std::string(accum_defn) != s->add_defn && std::string(accum_defn) != s->mult_defn;
    if (own_accum) src += device_function(accum_defn);
    const std::string T = s->type_name, ADD = s->add_name, ACC = accum_name ? accum_name : s->add_name;
    std::string MULT = s->mult_name;
    if (flip_mult) {
        // z = f (y, x): a user-defined multiplier cannot be rewritten, so its operands are handed over in the other
        // order (Source/GB_flip_binop.c:355-375 leaves flipxy set for such an operator)
        MULT = std::string("gba_flipped_") + s->mult_name;
        src += "__device__ inline void " + MULT + " (" + T + " *z, const " + T + " *x, const " + T + " *y) { " + s->mult_name + " (z, y, x) ; }\n";
    }
    src +=
        "extern \"C\" __global__ void gba_udt_dot (int64_t nvec, int64_t tnz, const int64_t *Tp, const int64_t *Ti, " + T + " *Tx,\n"
        "    const int64_t *Ap, const int64_t *Ai, const " + T + " *Ax, int A_iso,\n"
        "    const int64_t *Bp, const int64_t *Bi, const " + T + " *Bx, int B_iso)\n"
        "{\n"
        "    const int64_t p = (int64_t) blockIdx.x * blockDim.x + threadIdx.x ;\n"
        "    if (p >= tnz) return ;\n"
        "    int64_t lo = 0, hi = nvec - 1 ;                 // vector j of T that holds entry p\n"
        "    while (lo < hi) { const int64_t mid = (lo + hi + 1) >> 1 ; if (Tp [mid] <= p) lo = mid ; else hi = mid - 1 ; }\n"
        "    const int64_t j = lo, i = Ti [p] ;\n"
        "    int64_t pa = Ap [i], pa1 = Ap [i+1], pb = Bp [j], pb1 = Bp [j+1] ;\n"
        "    " + T + " z ; int first = 1 ;\n"
        // a list more than eight times longer than the other is searched, not walked (the rule of
        // Source/Template/GB_AxB_dot_cij.c:555-678); the terms still arrive in increasing k
        "    const int64_t alen = pa1 - pa, blen = pb1 - pb ;\n"
        "    if (alen > 8 * blen)\n"
        "    {\n"
        "        for ( ; pb < pb1 && pa < pa1 ; pb++)\n"
        "        {\n"
        "            const int64_t kb = Bi [pb] ;\n"
        "            int64_t hi = pa1 ;\n"
        "            while (pa < hi) { const int64_t mid = (pa + hi) >> 1 ; if (Ai [mid] < kb) pa = mid + 1 ; else hi = mid ; }\n"
        "            if (pa < pa1 && Ai [pa] == kb)\n"
        "            {\n"
        "                " + T + " t ;\n"
        "                " + MULT + " (&t, &Ax [A_iso ? 0 : pa], &Bx [B_iso ? 0 : pb]) ;\n"
        "                if (first) { z = t ; first = 0 ; } else " + ADD + " (&z, &z, &t) ;\n"
        "                pa++ ;\n"
        "            }\n"
        "        }\n"
        "    }\n"
        "    else if (blen > 8 * alen)\n"
        "    {\n"
        "        for ( ; pa < pa1 && pb < pb1 ; pa++)\n"
        "        {\n"
        "            const int64_t ka = Ai [pa] ;\n"
        "            int64_t hi = pb1 ;\n"
        "            while (pb < hi) { const int64_t mid = (pb + hi) >> 1 ; if (Bi [mid] < ka) pb = mid + 1 ; else hi = mid ; }\n"
        "            if (pb < pb1 && Bi [pb] == ka)\n"
        "            {\n"
        "                " + T + " t ;\n"
        "                " + MULT + " (&t, &Ax [A_iso ? 0 : pa], &Bx [B_iso ? 0 : pb]) ;\n"
        "                if (first) { z = t ; first = 0 ; } else " + ADD + " (&z, &z, &t) ;\n"
        "                pb++ ;\n"
        "            }\n"
        "        }\n"
        "    }\n"
        "    else while (pa < pa1 && pb < pb1)\n"
        "    {\n"
        "        const int64_t ka = Ai [pa], kb = Bi [pb] ;\n"
        "        if (ka < kb) pa++ ; else if (kb < ka) pb++ ;\n"
        "        else\n"
        "        {\n"
        "            " + T + " t ;\n"
        "            " + MULT + " (&t, &Ax [A_iso ? 0 : pa], &Bx [B_iso ? 0 : pb]) ;\n"
        "            if (first) { z = t ; first = 0 ; } else " + ADD + " (&z, &z, &t) ;\n"
        "            pa++ ; pb++ ;\n"
        "        }\n"
        "    }\n"
        "    Tx [p] = z ;\n"
        "}\n"
        "extern \"C\" __global__ void gba_udt_accum (int64_t nvec, int64_t tnz, const int64_t *Tp, const int64_t *Ti, const " + T + " *Tx,\n"
        "    " + T + " *Cx, int8_t *Cb, int64_t stride_i, int64_t stride_j)\n"
        "{\n"
        "    const int64_t p = (int64_t) blockIdx.x * blockDim.x + threadIdx.x ;\n"
        "    if (p >= tnz) return ;\n"
        "    int64_t lo = 0, hi = nvec - 1 ;\n"
        "    while (lo < hi) { const int64_t mid = (lo + hi + 1) >> 1 ; if (Tp [mid] <= p) lo = mid ; else hi = mid - 1 ; }\n"
        "    const int64_t pos = Ti [p] * stride_i + lo * stride_j ;\n"
        "    if (!Cb || Cb [pos]) " + ACC + " (&Cx [pos], &Cx [pos], &Tx [p]) ;\n"
        "    else { Cx [pos] = Tx [p] ; Cb [pos] = 1 ; }\n"
        "}\n"
        // values of Z = accum (C,T) from the positions the write-back left behind (writeback_positions)
        "extern \"C\" __global__ void gba_udt_merge (int64_t n, const int64_t *Zx, const int8_t *Zb, const " + T + " *Cx,\n"
        "    const " + T + " *Tx, " + T + " *out)\n"
        "{\n"
        "    const int64_t p = (int64_t) blockIdx.x * blockDim.x + threadIdx.x ;\n"
        "    if (p >= n || (Zb && !Zb [p])) return ;\n"
        "    const uint64_t v = (uint64_t) Zx [p], pc = v & 0xffffffffull, pt = v >> 32 ;\n"
        "    if (pc && pt) " + ACC + " (&out [p], &Cx [pc - 1], &Tx [pt - 1]) ;\n"
        "    else if (pt) out [p] = Tx [pt - 1] ;\n"
        "    else if (pc) out [p] = Cx [pc - 1] ;\n"
        "}\n";

    std::lock_guard<std::mutex> lock(g_jit_mu);
    const std::string key = std::to_string(ctx().device) + "\n" + src;      // modules belong to the device they were loaded on
    auto it = g_jit_cache.find(key);
    if (it != g_jit_cache.end()) { *out = it->second; return GBA_SUCCESS; }

    hipDeviceProp_t prop;
    GBA_HIP(hipGetDeviceProperties(&prop, ctx().device));
    const std::string arch = std::string("--offload-arch=") + prop.gcnArchName;
    // what the code object depends on: the target, the options and the source.  Kept on disk under that identity, so a
    // later process loads the code object instead of compiling (the reference keeps its compiled kernels the same
    // way, Source/GB_jitifyer.c: a cache folder, one file per kernel, checked against its source when loaded).
    const std::string identity = arch + " -O3 -ffp-contract=off\n" + src;
    std::string code;
    const bool from_disk = jit_disk_load(identity, &code);
    if (!from_disk) {
        hiprtcProgram prog;
        GBA_RTC(hiprtcCreateProgram(&prog, src.c_str(), "gba_udt.hip", 0, nullptr, nullptr));
        const char *opts[] = {arch.c_str(), "-O3", "-ffp-contract=off"};
        const hiprtcResult cr = hiprtcCompileProgram(prog, 3, opts);
        if (cr != HIPRTC_SUCCESS) {
            size_t n = 0; hiprtcGetProgramLogSize(prog, &n);
            std::string log(n, '\0'); if (n) hiprtcGetProgramLog(prog, &log[0]);
            hiprtcDestroyProgram(&prog);
            GBA_FAIL(GBA_INVALID_VALUE, "the user-defined semiring does not compile: %.600s", log.c_str());
        }
        size_t csize = 0;
        GBA_RTC(hiprtcGetCodeSize(prog, &csize));
        code.assign(csize, '\0');
        GBA_RTC(hiprtcGetCode(prog, &code[0]));
        hiprtcDestroyProgram(&prog);
        jit_disk_store(identity, code);
    }
    g_jit_compiles += from_disk ? 0 : 1;
    g_jit_disk_hits += from_disk ? 1 : 0;
    JitKernels k;
    GBA_HIP(hipModuleLoadData(&k.module, code.data()));
    GBA_HIP(hipModuleGetFunction(&k.dot, k.module, "gba_udt_dot"));
    GBA_HIP(hipModuleGetFunction(&k.accum, k.module, "gba_udt_accum"));
    GBA_HIP(hipModuleGetFunction(&k.merge, k.module, "gba_udt_merge"));
    g_jit_cache[key] = k;
    *out = k;
    return GBA_SUCCESS;
}

// the operand as a sparse matrix whose vectors are the rows (want_rows) or the columns of op(X)
static gba_info sparse_lists(const gba_matrix *X, bool tran, bool want_rows, gba_matrix *out, std::vector<void *> &owned) {
    gba_matrix S = *X;
    if (X->nzombies) GBA_FAIL(GBA_INVALID_OBJECT, "operand has zombies");
    if (is_hyper(X)) GBA_OK(hyper_to_sparse(X, &S, owned));            // a pointer per vector; i and x are shared
    else if (is_dense_fmt(X)) GBA_OK(to_sparse(X, &S, owned));
    // vectors of S are rows of X when X is held by row; op(X) = X' swaps rows and columns
    const bool vectors_are_rows = (!S.is_csc) != tran;
    if (vectors_are_rows == want_rows) {
        if (S.jumbled) GBA_FAIL(GBA_NOT_IMPLEMENTED, "jumbled operand (wait on it first)");
        *out = S;
        return GBA_SUCCESS;
    }
    gba_matrix Tm;
    GBA_OK(transpose(&Tm, &S));
    owned.push_back(Tm.p); owned.push_back(Tm.i); owned.push_back(Tm.x);
    *out = Tm;
    return GBA_SUCCESS;
}

__global__ void __launch_bounds__(256) positions_kernel(int64_t *x, int64_t n, int shift) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) x[p] = (p + 1) << shift;
}

// C<M> = accum (C,T), any combination that the typed write-back takes (sparse or bitmap/full C with entries,
// any mask, replace or not, with or without accum), for values of s bytes: the pattern work is done by
// writeback_positions on INT64 stand-ins, the compiled merge kernel then fetches or combines the values.
// T is sparse, in C's orientation, and is consumed.
static gba_info udt_writeback(gba_matrix *C, const gba_matrix *M, bool has_accum, bool C_replace, bool Mask_comp,
                              bool Mask_struct, gba_matrix *T, const JitKernels &jit, size_t esize) {
    hipStream_t st = ctx().stream;
    struct Guard { gba_matrix *T; gba_matrix Cpos{}, Tpos{}; void *out = nullptr;
                   ~Guard() { gba_matrix_free_content(&Cpos); gba_matrix_free_content(&Tpos); gba_matrix_free_content(T); gba_free(out); } } g{T};
    if (C->iso || C->nzombies || is_hyper(C)) GBA_FAIL(GBA_NOT_IMPLEMENTED, "gba_mxm_udt: write-back into an iso-valued, hypersparse or zombie-holding C");
    const bool dense = is_dense_fmt(C);
    const int64_t nslots = C->vlen * C->vdim;
    const int64_t cnz = dense ? nslots : C->nvals, tnz = T->nvals;
    const int64_t most = ((int64_t)1 << 32) - 2;
    if (cnz > most || tnz > most) GBA_FAIL(GBA_NOT_IMPLEMENTED, "gba_mxm_udt: write-back of more than 2^32 - 2 entries");
    auto dup = [&](const void *src, size_t bytes, void **dst) -> gba_info {
        *dst = gba_malloc(bytes ? bytes : 1);
        if (!*dst) GBA_FAIL(GBA_OUT_OF_MEMORY, "gba_mxm_udt: write-back copy of %zu bytes", bytes);
        if (src && bytes) GBA_HIP(hipMemcpyAsync(*dst, src, bytes, hipMemcpyDeviceToDevice, st));
        return GBA_SUCCESS;
    };
    auto positions = [&](int64_t n, int shift, void **dst) -> gba_info {
        *dst = gba_malloc((size_t)(n > 0 ? n : 1) * 8);
        if (!*dst) GBA_FAIL(GBA_OUT_OF_MEMORY, "gba_mxm_udt: positions of %ld entries", (long)n);
        if (n > 0) hipLaunchKernelGGL(positions_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, reinterpret_cast<int64_t *>(*dst), n, shift);
        return GBA_SUCCESS;
    };
    gba_matrix &Cpos = g.Cpos, &Tpos = g.Tpos;
    Cpos = *C; Cpos.type = GBA_INT64; Cpos.p = Cpos.i = Cpos.h = nullptr; Cpos.x = nullptr; Cpos.b = nullptr;
    if (is_sparse(C)) {
        GBA_OK(dup(C->p, (size_t)(C->nvec + 1) * 8, (void **)&Cpos.p));
        GBA_OK(dup(C->i, (size_t)cnz * 8, (void **)&Cpos.i));
    } else if (is_bitmap(C)) {
        GBA_OK(dup(C->b, (size_t)nslots, (void **)&Cpos.b));
    }
    GBA_OK(positions(cnz, 0, &Cpos.x));
    Tpos = *T; Tpos.type = GBA_INT64; Tpos.iso = false; Tpos.x = nullptr;
    T->p = T->i = T->h = nullptr; T->b = nullptr;            // the pattern of T now belongs to Tpos; T keeps its values
    GBA_OK(positions(tnz, 32, &Tpos.x));
    const void *Tx = T->x, *Cx = C->x;
    GBA_OK(writeback_positions(&Cpos, M, has_accum, C_replace, Mask_comp, Mask_struct, &Tpos));
    const bool zdense = is_dense_fmt(&Cpos);
    int64_t zn = zdense ? nslots : Cpos.nvals;
    g.out = gba_malloc((size_t)(zn > 0 ? zn : 1) * esize);
    if (!g.out) GBA_FAIL(GBA_OUT_OF_MEMORY, "gba_mxm_udt: values of the result");
    if (zn > 0) {
        const int8_t *Zb = is_bitmap(&Cpos) ? Cpos.b : nullptr;
        void *args[] = {&zn, &Cpos.x, (void *)&Zb, (void *)&Cx, (void *)&Tx, &g.out};
        const hipError_t e = hipModuleLaunchKernel(jit.merge, (unsigned)ceil_div(zn, 256), 1, 1, 256, 1, 1, 0, st, args, nullptr);
        if (e != hipSuccess) GBA_FAIL(GBA_PANIC, "launch of the compiled merge kernel: %s", hipGetErrorString(e));
    }
    GBA_HIP(hipStreamSynchronize(st));
    gba_matrix_free_content(C);                              // the old arrays were read above
    gba_free(Cpos.x);
    *C = Cpos;
    C->x = g.out; C->type = GBA_UDT; C->mem = GBA_MEM_DEVICE;
    g.out = nullptr; Cpos = gba_matrix{};                   // handed over
    return GBA_SUCCESS;
}

// set while a caller needs the product with sorted vectors (it goes on as the operand of another compiled dot
// kernel): the pattern is sorted before the values are computed, so only indices move
static thread_local bool g_udt_sorted_pattern = false;     // per host thread: concurrent calls do not see each other's switch

static gba_matrix pattern_view(const gba_matrix &S) {
    gba_matrix v = S;
    v.type = GBA_BOOL; v.iso = true;        // ANY_PAIR reads no values
    return v;
}

static gba_info mxm_udt_body(gba_matrix *C, const gba_matrix *M, const char *accum_name, const char *accum_defn,
                 const gba_user_semiring *s, const gba_matrix *A, const gba_matrix *B, const gba_descriptor *desc,
                 gba_report *report, bool flip_mult) {
    GBA_OK(require_init());
    if (!C || !s || !A || !B) GBA_FAIL(GBA_NULL_POINTER, "gba_mxm_udt: null argument");
    if (!s->type_name || !s->type_defn || !s->add_name || !s->add_defn || !s->mult_name || !s->mult_defn || s->type_size == 0)
        GBA_FAIL(GBA_NULL_POINTER, "gba_mxm_udt: the semiring needs type and operator names and definitions");
    if (A->type != GBA_UDT || B->type != GBA_UDT || C->type != GBA_UDT)
        GBA_FAIL(GBA_DOMAIN_MISMATCH, "gba_mxm_udt: A, B and C must have the user-defined type");
    if (C->mem != GBA_MEM_DEVICE || A->mem != GBA_MEM_DEVICE || B->mem != GBA_MEM_DEVICE || (M && M->mem != GBA_MEM_DEVICE))
        GBA_FAIL(GBA_NOT_IMPLEMENTED, "gba_mxm_udt: device views only");
    const bool A_tran = desc && desc->in0 == GBA_TRAN, B_tran = desc && desc->in1 == GBA_TRAN;
    const bool C_replace = desc && desc->out == GBA_REPLACE;
    const bool Mask_comp = desc && (desc->mask == GBA_COMP || desc->mask == GBA_COMP_STRUCTURE);
    const bool Mask_struct = desc && (desc->mask == GBA_STRUCTURE || desc->mask == GBA_COMP_STRUCTURE);
    auto rows = [](const gba_matrix *X) { return X->is_csc ? X->vlen : X->vdim; };
    auto cols = [](const gba_matrix *X) { return X->is_csc ? X->vdim : X->vlen; };
    const int64_t m = A_tran ? cols(A) : rows(A), ka = A_tran ? rows(A) : cols(A);
    const int64_t kb = B_tran ? cols(B) : rows(B), n = B_tran ? rows(B) : cols(B);
    if (ka != kb || rows(C) != m || cols(C) != n) GBA_FAIL(GBA_DIMENSION_MISMATCH, "gba_mxm_udt: %ld-by-%ld times %ld-by-%ld into %ld-by-%ld",
                                                          (long)m, (long)ka, (long)kb, (long)n, (long)rows(C), (long)cols(C));
    if (M && (rows(M) != m || cols(M) != n)) GBA_FAIL(GBA_DIMENSION_MISMATCH, "gba_mxm_udt: mask dimensions");
    if (!M && Mask_comp) GBA_FAIL(GBA_NOT_IMPLEMENTED, "gba_mxm_udt: complemented empty mask");
    hipStream_t st = ctx().stream;
    struct SizeScope { size_t old; SizeScope(size_t v) : old(ctx().udt_size) { ctx().udt_size = v; } ~SizeScope() { ctx().udt_size = old; } }
        scope(s->type_size);
    gba_report rep{};
    KernelTimer timer(report != nullptr);

    JitKernels jit;
    GBA_OK(jit_build(s, accum_name, accum_defn, flip_mult, &jit));

    struct Owned { std::vector<void *> v; ~Owned() { for (void *p : v) gba_free(p); } } own;
    // hypersparse mask and C: the sparse form of the same matrix (a pointer per vector); a C that came in
    // hypersparse goes back hypersparse
    gba_matrix Msp;
    if (M && is_hyper(M)) { GBA_OK(hyper_to_sparse(M, &Msp, own.v)); M = &Msp; }
    const bool C_was_hyper = is_hyper(C);
    if (C_was_hyper) {
        std::vector<void *> mine;
        gba_matrix Csp;
        GBA_OK(hyper_to_sparse(C, &Csp, mine));             // C owns its arrays: the new p replaces p and h
        gba_free(C->p); gba_free(C->h);
        *C = Csp;
    }
    gba_matrix Arows, Acols, Bcols, Mcol;
    GBA_OK(sparse_lists(A, A_tran, true, &Arows, own.v));
    GBA_OK(sparse_lists(A, A_tran, false, &Acols, own.v));
    GBA_OK(sparse_lists(B, B_tran, false, &Bcols, own.v));
    const gba_matrix *Mc = nullptr;
    if (M) {
        Mcol = *M;
        if (!M->is_csc) {                                  // the pattern pass works by column
            gba_matrix Ms = *M;
            if (!is_sparse(M)) GBA_OK(to_sparse(M, &Ms, own.v));       // a bitmap/full mask held by row: its entries as lists first
            GBA_OK(transpose(&Mcol, &Ms)); own.v.push_back(Mcol.p); own.v.push_back(Mcol.i); own.v.push_back(Mcol.x);
        }
        Mcol.is_csc = true;
        Mc = &Mcol;
    }

    // pattern of T = op(A)*op(B), by column
    gba_semiring anypair{GBA_ANY, GBA_PAIR, GBA_BOOL, GBA_BOOL, GBA_BOOL};
    Canon c;
    GBA_OK(canonicalize(&anypair, false, GBA_BOOL, GBA_BOOL, &c));
    gba_matrix Ap_ = pattern_view(Acols), Bp_ = pattern_view(Bcols), T{};
    Ap_.is_csc = Bp_.is_csc = true;
    bool applied = false;
    GBA_OK(saxpy3(&T, Mc, Mask_comp, Mask_struct, &Ap_, &Bp_, c, 0, g_udt_sorted_pattern ? 1 : 0, &applied, nullptr));
    const int64_t tnz = T.nvals;
    gba_free(T.x); T.x = nullptr;                          // the boolean values are not wanted
    void *Tx = gba_malloc((size_t)(tnz > 0 ? tnz : 1) * s->type_size);
    if (!Tx) { gba_matrix_free_content(&T); GBA_FAIL(GBA_OUT_OF_MEMORY, "gba_mxm_udt: values of T"); }
    T.x = Tx; T.type = GBA_UDT; T.iso = false; T.is_csc = true;
    if (tnz > 0) {
        int64_t nvec = T.nvec; int a_iso = Arows.iso, b_iso = Bcols.iso;
        void *args[] = {&nvec, (void *)&tnz, &T.p, &T.i, &T.x, &Arows.p, &Arows.i, &Arows.x, &a_iso, &Bcols.p, &Bcols.i, &Bcols.x, &b_iso};
        const hipError_t e = hipModuleLaunchKernel(jit.dot, (unsigned)ceil_div(tnz, 256), 1, 1, 256, 1, 1, 0, st, args, nullptr);
        if (e != hipSuccess) { gba_matrix_free_content(&T); GBA_FAIL(GBA_PANIC, "launch of the compiled dot kernel: %s", hipGetErrorString(e)); }
    }
    GBA_HIP(hipStreamSynchronize(st));
    rep.method = GBA_METHOD_DOT3; rep.mask_applied = (M != nullptr); rep.cnz = tnz;

    // C<M> = accum (C,T)
    const bool has_accum = accum_name != nullptr;
    const bool c_empty = is_sparse(C) && C->nvals == 0;
    gba_info info = GBA_SUCCESS;
    if (!has_accum && (c_empty || C_replace || !M)) {
        // no mask: C = T.  masked: T already holds only permitted entries, and C is empty or replaced
        if (!C->is_csc) {
            gba_matrix Tr;
            info = transpose(&Tr, &T);
            gba_matrix_free_content(&T);
            if (info != GBA_SUCCESS) return info;
            T = Tr;
        }
        const bool keep_csc = C->is_csc;
        // the old arrays of C belong to the caller's handle; when C is also an input they were read above
        gba_matrix_free_content(C);
        *C = T;
        C->is_csc = keep_csc; C->type = GBA_UDT; C->mem = GBA_MEM_DEVICE;
    } else if (has_accum && !M && is_dense_fmt(C) && !C->iso) {
        if (tnz > 0) {
            int64_t nvec = T.nvec;
            // T is by column: vector j = column, index i = row; C(i,j) sits at i*stride_i + j*stride_j
            int64_t stride_i = C->is_csc ? 1 : C->vlen, stride_j = C->is_csc ? C->vlen : 1;
            int8_t *Cb = is_bitmap(C) ? C->b : nullptr;
            void *args[] = {&nvec, (void *)&tnz, &T.p, &T.i, &T.x, &C->x, &Cb, &stride_i, &stride_j};
            const hipError_t e = hipModuleLaunchKernel(jit.accum, (unsigned)ceil_div(tnz, 256), 1, 1, 256, 1, 1, 0, st, args, nullptr);
            if (e != hipSuccess) info = GBA_PANIC;
            GBA_HIP(hipStreamSynchronize(st));
            if (is_bitmap(C)) { int64_t nv = 0; if (dev_count_ones(C->b, C->vlen * C->vdim, &nv) == GBA_SUCCESS) C->nvals = nv; }
        }
        gba_matrix_free_content(&T);
        rep.done_in_place = 1;
    } else {
        // the general case: C holds entries that the mask or the accumulator must keep or combine
        if (!C->is_csc) {
            gba_matrix Tr;
            info = transpose(&Tr, &T);
            gba_matrix_free_content(&T);
            if (info != GBA_SUCCESS) return info;
            T = Tr;
        }
        T.is_csc = C->is_csc;
        info = udt_writeback(C, M, has_accum, C_replace, Mask_comp, Mask_struct, &T, jit, s->type_size);
        if (info != GBA_SUCCESS) return info;
    }
    if (report) { rep.kernel_ms = timer.stop_ms(); *report = rep; }
    if (info == GBA_SUCCESS && C_was_hyper && is_sparse(C)) info = sparse_to_hyper(C);
    return info;
}

gba_info mxm_udt(gba_matrix *C, const gba_matrix *M, const char *accum_name, const char *accum_defn,
                 const gba_user_semiring *s, const gba_matrix *A, const gba_matrix *B, const gba_descriptor *desc,
                 gba_report *report, bool flip_mult) {
    // a call that fails or is declined after a hypersparse C was expanded hands C back hypersparse, as it came
    const bool C_was_hyper = C && is_hyper(C);
    const gba_info info = mxm_udt_body(C, M, accum_name, accum_defn, s, A, B, desc, report, flip_mult);
    if (info != GBA_SUCCESS && C_was_hyper && is_sparse(C) && C->mem == GBA_MEM_DEVICE) {
        const std::string keep = last_error();
        if (sparse_to_hyper(C) == GBA_SUCCESS) set_error("%s", keep.c_str());
    }
    return info;
}

// ------------------------------------------------------------------ complex --
// The built-in complex semirings -- (PLUS, TIMES, ANY) x (FIRST, SECOND, PAIR, PLUS, MINUS, RMINUS, TIMES,
// DIV, RDIV) over GxB_FC32 and GxB_FC64, Source/Factories/GB_AxB_factory.c:46-243 with GB_COMPLEX and
// GB_AxB_type_factory.c:100-154 -- take the road of the user-defined types: a 16-byte value has no atomic
// update, and the per-entry dot product needs none.  The operators are written out as source for hipRTC:
// x*y, x+y, x-y as the C99 operators of Source/Shared/GB_complex.h:237-246 compile them (products and
// sums in the type's own precision, no fused multiply-add, the infinity recovery of C99 Annex G.5.1 when
// both parts come out NaN), and the library's own division, GB_FC64_div / GB_FC32_div
// (Source/GB_math.h:301-389: Smith's three cases; single complex divides in double and rounds).
static std::string complex_type_defn(bool dbl) {
    return dbl ? "typedef struct { double re ; double im ; } gba_fc64 ;" : "typedef struct { float re ; float im ; } gba_fc32 ;";
}

static bool complex_op_source(int op, bool dbl, std::string *name, std::string *defn) {
    const std::string T = dbl ? "gba_fc64" : "gba_fc32", R = dbl ? "double" : "float";
    const char *tag = nullptr; std::string body;
    switch (op) {
        case GBA_FIRST: case GBA_ANY: tag = "first"; body = "const " + T + " a = *x ; *z = a ;"; break;
        case GBA_SECOND: tag = "second"; body = "const " + T + " b = *y ; *z = b ;"; break;
        case GBA_PAIR:   tag = "pair";   body = "z->re = 1 ; z->im = 0 ;"; break;
        case GBA_PLUS:   tag = "plus";   body = "const " + T + " a = *x, b = *y ; z->re = a.re + b.re ; z->im = a.im + b.im ;"; break;
        case GBA_MINUS:  tag = "minus";  body = "const " + T + " a = *x, b = *y ; z->re = a.re - b.re ; z->im = a.im - b.im ;"; break;
        case GBA_RMINUS: tag = "rminus"; body = "const " + T + " a = *x, b = *y ; z->re = b.re - a.re ; z->im = b.im - a.im ;"; break;
        case GBA_TIMES:  tag = "times";
            body = R + " a = x->re, b = x->im, c = y->re, d = y->im ;\n"
                   "    const " + R + " ac = a * c, bd = b * d, ad = a * d, bc = b * c ;\n"
                   "    " + R + " re = ac - bd, im = ad + bc ;\n"
                   "    if (__builtin_isnan (re) && __builtin_isnan (im))\n"
                   "    {\n"
                   "        int recalc = 0 ;\n"
                   "        if (__builtin_isinf (a) || __builtin_isinf (b))\n"
                   "        {\n"
                   "            a = __builtin_copysign (__builtin_isinf (a) ? 1.0 : 0.0, a) ; b = __builtin_copysign (__builtin_isinf (b) ? 1.0 : 0.0, b) ;\n"
                   "            if (__builtin_isnan (c)) c = __builtin_copysign (0.0, c) ;\n"
                   "            if (__builtin_isnan (d)) d = __builtin_copysign (0.0, d) ;\n"
                   "            recalc = 1 ;\n"
                   "        }\n"
                   "        if (__builtin_isinf (c) || __builtin_isinf (d))\n"
                   "        {\n"
                   "            c = __builtin_copysign (__builtin_isinf (c) ? 1.0 : 0.0, c) ; d = __builtin_copysign (__builtin_isinf (d) ? 1.0 : 0.0, d) ;\n"
                   "            if (__builtin_isnan (a)) a = __builtin_copysign (0.0, a) ;\n"
                   "            if (__builtin_isnan (b)) b = __builtin_copysign (0.0, b) ;\n"
                   "            recalc = 1 ;\n"
                   "        }\n"
                   "        if (!recalc && (__builtin_isinf (ac) || __builtin_isinf (bd) || __builtin_isinf (ad) || __builtin_isinf (bc)))\n"
                   "        {\n"
                   "            if (__builtin_isnan (a)) a = __builtin_copysign (0.0, a) ;\n"
                   "            if (__builtin_isnan (b)) b = __builtin_copysign (0.0, b) ;\n"
                   "            if (__builtin_isnan (c)) c = __builtin_copysign (0.0, c) ;\n"
                   "            if (__builtin_isnan (d)) d = __builtin_copysign (0.0, d) ;\n"
                   "            recalc = 1 ;\n"
                   "        }\n"
                   "        if (recalc)\n"
                   "        {\n"
                   "            re = (" + R + ") __builtin_huge_val () * (a * c - b * d) ;\n"
                   "            im = (" + R + ") __builtin_huge_val () * (a * d + b * c) ;\n"
                   "        }\n"
                   "    }\n"
                   "    z->re = re ; z->im = im ;";
            break;
        case GBA_DIV: case GBA_RDIV: {
            tag = (op == GBA_DIV) ? "div" : "rdiv";
            const std::string num = (op == GBA_DIV) ? "x" : "y", den = (op == GBA_DIV) ? "y" : "x";
            body = "const double xr = " + num + "->re, xi = " + num + "->im, yr = " + den + "->re, yi = " + den + "->im ;\n"
                   "    double zr, zi ;\n"
                   "    if (yi == 0) { zr = xr / yr ; zi = xi / yr ; }\n"
                   "    else if (yr == 0) { zr = xi / yi ; zi = -xr / yi ; }\n"
                   "    else if (__builtin_isinf (yi) && __builtin_isinf (yr))\n"
                   "    {\n"
                   "        const double r = (__builtin_signbit (yr) == __builtin_signbit (yi)) ? 1 : -1, d = yr + r * yi ;\n"
                   "        zr = (xr + xi * r) / d ; zi = (xi - xr * r) / d ;\n"
                   "    }\n"
                   "    else if (__builtin_fabs (yr) >= __builtin_fabs (yi))\n"
                   "    {\n"
                   "        const double r = yi / yr, d = yr + r * yi ;\n"
                   "        zr = (xr + xi * r) / d ; zi = (xi - xr * r) / d ;\n"
                   "    }\n"
                   "    else\n"
                   "    {\n"
                   "        const double r = yr / yi, d = r * yr + yi ;\n"
                   "        zr = (xr * r + xi) / d ; zi = (xi * r - xr) / d ;\n"
                   "    }\n"
                   "    z->re = (" + R + ") zr ; z->im = (" + R + ") zi ;";
            break;
        }
        default: return false;
    }
    *name = T + "_" + tag;
    *defn = "void " + *name + " (" + T + " *z, const " + T + " *x, const " + T + " *y)\n{\n    " + body + "\n}";
    return true;
}

gba_info mxm_complex(gba_matrix *C, const gba_matrix *M, const gba_binop *accum, const gba_semiring *s,
                     const gba_matrix *A, bool A_tran, const gba_matrix *B, bool B_tran, bool flipxy,
                     bool C_replace, bool Mask_comp, bool Mask_struct, gba_report *report) {
    GBA_OK(require_init());
    if (!C || !A || !B || !s) GBA_FAIL(GBA_NULL_POINTER, "C, A, B or the semiring is NULL");
    const int zt = s->ztype;
    const bool dbl = (zt == GBA_FC64);
    int mult = s->mult, add = s->add;
    const bool flipxy_in = flipxy;
    mult = flip_binop(mult, &flipxy);
    if (flipxy) GBA_FAIL(GBA_NOT_IMPLEMENTED, "multiplier %d cannot be flipped", mult);
    if (!(add == GBA_PLUS || add == GBA_TIMES || add == GBA_ANY))
        GBA_FAIL(GBA_DOMAIN_MISMATCH, "monoid %d has no complex form (PLUS, TIMES and ANY do)", add);
    // the operands the multiplier reads, the product and C all have the semiring's complex type: casts between
    // complex and real values are left to the CPU library
    const bool a_read = !(mult == GBA_SECOND || mult == GBA_PAIR), b_read = !(mult == GBA_FIRST || mult == GBA_PAIR);
    if (mult != GBA_PAIR && ((a_read && s->xtype != zt) || (b_read && s->ytype != zt)))
        GBA_FAIL(GBA_NOT_IMPLEMENTED, "complex monoid with a multiplier over types %d, %d", s->xtype, s->ytype);
    // A and B of another built-in type are cast to the semiring's (real part = the value, Source/GB_casting.h).  A C
    // of another built-in type: the product is computed into an empty matrix of the semiring's type, cast to C's
    // type (the real part, Source/GB_casting.h:211-446) and assigned under the mask and the accumulator
    if (C->type != zt) {
        if (!is_real_type(C->type) && !is_complex_type(C->type)) GBA_FAIL(GBA_DOMAIN_MISMATCH, "C of type %d under a complex semiring", C->type);
        if (C->mem != GBA_MEM_DEVICE) GBA_FAIL(GBA_NOT_IMPLEMENTED, "C must be a device view (mem == GBA_MEM_DEVICE)");
        if (accum && accum->op != GBA_NOP && accum->type != C->type) GBA_FAIL(GBA_NOT_IMPLEMENTED, "accumulator of type %d on a C of type %d", accum->type, C->type);
        const int64_t nvec = C->vdim;
        gba_matrix Tt{};
        Tt.vlen = C->vlen; Tt.vdim = Tt.nvec = nvec; Tt.type = zt; Tt.sparsity = GBA_SPARSE; Tt.mem = GBA_MEM_DEVICE; Tt.is_csc = C->is_csc;
        Tt.p = reinterpret_cast<int64_t *>(gba_malloc((size_t)(nvec + 1) * 8));
        if (!Tt.p) GBA_FAIL(GBA_OUT_OF_MEMORY, "pointers of the product");
        struct Guard { gba_matrix *m; std::vector<void *> owned; ~Guard() { gba_matrix_free_content(m); for (void *p : owned) gba_free(p); } } g{&Tt, {}};
        GBA_HIP(hipMemsetAsync(Tt.p, 0, (size_t)(nvec + 1) * 8, ctx().stream));
        gba_report rep{};
        {
            struct Sorted { bool old; Sorted() : old(g_udt_sorted_pattern) { g_udt_sorted_pattern = true; } ~Sorted() { g_udt_sorted_pattern = old; } } sorted;
            GBA_OK(mxm_complex(&Tt, nullptr, nullptr, s, A, A_tran, B, B_tran, flipxy_in, false, false, false, &rep));
        }
        gba_matrix Tc;
        GBA_OK(cast_operand(&Tt, C->type, &Tc, g.owned));
        GBA_HIP(hipStreamSynchronize(ctx().stream));
        const gba_info info = mxm_assign_cast(C, M, accum, C_replace, Mask_comp, Mask_struct, &Tc, nullptr);
        if (report) { rep.typecast |= 4; *report = rep; }
        return info;
    }
    for (const gba_matrix *X : {A, B})
        if (X->type != zt && !is_real_type(X->type) && !is_complex_type(X->type))
            GBA_FAIL(GBA_DOMAIN_MISMATCH, "operand of type %d under a complex semiring", X->type);
    if (C->mem != GBA_MEM_DEVICE) GBA_FAIL(GBA_NOT_IMPLEMENTED, "C must be a device view (mem == GBA_MEM_DEVICE)");
    if (M && !Mask_struct && !is_real_type(M->type)) GBA_FAIL(GBA_NOT_IMPLEMENTED, "valued mask of type %d", M->type);
    std::string add_name, add_defn, mult_name, mult_defn, acc_name, acc_defn;
    if (!complex_op_source(add, dbl, &add_name, &add_defn) || !complex_op_source(mult, dbl, &mult_name, &mult_defn))
        GBA_FAIL(GBA_DOMAIN_MISMATCH, "operator %d / %d has no complex form", add, mult);
    const bool has_accum = accum && accum->op != GBA_NOP;
    if (has_accum) {
        if (accum->type != zt) GBA_FAIL(GBA_NOT_IMPLEMENTED, "accumulator of type %d on a complex C (typecast)", accum->type);
        if (!complex_op_source(accum->op, dbl, &acc_name, &acc_defn)) GBA_FAIL(GBA_DOMAIN_MISMATCH, "accumulator %d has no complex form", accum->op);
    }
    const std::string tdefn = complex_type_defn(dbl);
    gba_user_semiring us{};
    us.type_name = dbl ? "gba_fc64" : "gba_fc32"; us.type_defn = tdefn.c_str(); us.type_size = type_size(zt);
    us.add_name = add_name.c_str(); us.add_defn = add_defn.c_str();
    us.mult_name = mult_name.c_str(); us.mult_defn = mult_defn.c_str();

    struct Owned { std::vector<void *> v; ~Owned() { for (void *p : v) gba_free(p); } } own;
    gba_matrix Av, Bv, Mv, Cv = *C;
    if (A->mem != GBA_MEM_DEVICE) GBA_OK(stage_to_device(A, &Av, own.v)); else Av = *A;
    if (B->mem != GBA_MEM_DEVICE) GBA_OK(stage_to_device(B, &Bv, own.v)); else Bv = *B;
    if (M) { if (M->mem != GBA_MEM_DEVICE) GBA_OK(stage_to_device(M, &Mv, own.v)); else Mv = *M; }
    gba_matrix Ac, Bc;
    if (Av.type != zt) { GBA_OK(cast_operand(&Av, zt, &Ac, own.v)); Av = Ac; }
    if (Bv.type != zt) { GBA_OK(cast_operand(&Bv, zt, &Bc, own.v)); Bv = Bc; }
    const int typecast = (A->type != zt ? 1 : 0) | (B->type != zt ? 2 : 0);
    Av.type = Bv.type = Cv.type = GBA_UDT;
    gba_descriptor d{};
    d.out = C_replace ? GBA_REPLACE : GBA_DEFAULT;
    d.mask = Mask_comp ? (Mask_struct ? GBA_COMP_STRUCTURE : GBA_COMP) : (Mask_struct ? GBA_STRUCTURE : GBA_DEFAULT);
    d.in0 = A_tran ? GBA_TRAN : GBA_DEFAULT; d.in1 = B_tran ? GBA_TRAN : GBA_DEFAULT;
    const gba_info info = mxm_udt(&Cv, M ? &Mv : nullptr, has_accum ? acc_name.c_str() : nullptr, has_accum ? acc_defn.c_str() : nullptr,
                                  &us, &Av, &Bv, &d, report, false);
    Cv.type = zt;
    *C = Cv;                    // on failure mxm_udt leaves the view as it was
    if (report && info == GBA_SUCCESS) report->typecast = typecast;
    return info;
}

}  // namespace gba

extern "C" gba_info gba_mxm_udt(gba_matrix *C, const gba_matrix *M, const char *accum_name, const char *accum_defn,
                                const gba_user_semiring *semiring, const gba_matrix *A, const gba_matrix *B,
                                const gba_descriptor *desc, gba_report *report) {
    return gba::mxm_udt(C, M, accum_name, accum_defn, semiring, A, B, desc, report, false);
}

extern "C" gba_info gba_jit_stats(uint64_t *compiled, uint64_t *loaded_from_disk) {
    std::lock_guard<std::mutex> lock(gba::g_jit_mu);
    if (compiled) *compiled = gba::g_jit_compiles;
    if (loaded_from_disk) *loaded_from_disk = gba::g_jit_disk_hits;
    return GBA_SUCCESS;
}

extern "C" gba_info gba_vxm_udt(gba_matrix *w, const gba_matrix *mask, const char *accum_name, const char *accum_defn,
                                const gba_user_semiring *semiring, const gba_matrix *u, const gba_matrix *A,
                                const gba_descriptor *desc, gba_report *report) {
    // w' = u'*A is w = A'*u with the multiplier's operands exchanged (Source/GrB_vxm.c:63-72); GrB_vxm reads the
    // transpose setting of A from INP1 and ignores INP0
    gba_descriptor d{};
    if (desc) d = *desc;
    d.in0 = (d.in1 == GBA_TRAN) ? GBA_DEFAULT : GBA_TRAN;
    d.in1 = GBA_DEFAULT;
    return gba::mxm_udt(w, mask, accum_name, accum_defn, semiring, A, u, &d, report, true);
}
