// Micro-benchmark (gfx950): what one CU sustains for the LDS operations of the panel kernels, 16 waves per CU,
// pseudo-random addresses: ds_or_b32, ds_add_f64, ds_add_f32, ds_read_b64, ds_add_u32 with and without return, and for random 256-byte
// chunk reads of a large array by depth of the per-wave ring.  Prints wave-instructions per microsecond per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int OP>
__global__ void __launch_bounds__(1024) lds_kernel(int iters, uint32_t span, double *sink) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *w32 = reinterpret_cast<uint32_t *>(smem);
    double *w64 = reinterpret_cast<double *>(smem);
    uint2 *wp = reinterpret_cast<uint2 *>(smem);
    float *wf = reinterpret_cast<float *>(smem);
    for (uint32_t q = threadIdx.x; q < span * 2; q += 1024) w32[q] = 0;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    double acc = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            x = x * 1664525u + 1013904223u;
            const uint32_t a = (x >> 8) % span;
            if (OP == 0) atomicOr(&w32[2 * a], 1u << (x & 31));
            if (OP == 1) atomicAdd(&w64[a], 1.0);
            if (OP == 2) { const uint2 r = wp[a]; acc += r.x + r.y; }
            if (OP == 3) acc += atomicAdd(&w32[a], 1u);
            if (OP == 4) w64[a] = 1.0;
            if (OP == 5) atomicAdd(&w64[(a & ~63u) + (threadIdx.x & 63)], 1.0);      // conflict-free fp64 adds
            if (OP == 6) atomicOr(&w32[2 * ((a & ~63u) + (threadIdx.x & 63))], 1u << (x & 31));
            if (OP == 7) atomicAdd(&wf[a], 1.0f);
            if (OP == 8) atomicAdd(&wf[(a & ~63u) + (threadIdx.x & 63)], 1.0f);
            if (OP == 9) atomicAdd(&w32[a], 1u);
        }
    }
    if (acc == 123.456) sink[0] = acc;
}

// every wave reads random 64-entry chunks (256 B of int32, optionally + 512 B of fp64), D chunks in flight
template <int D, bool VALS>
__global__ void __launch_bounds__(1024) chunk_kernel(const int32_t *__restrict__ idx, const double *__restrict__ val, uint64_t nchunks,
                                                     int iters, double *sink) {
    const int lane = threadIdx.x & 63;
    uint32_t x = (threadIdx.x >> 6) * 2654435761u + blockIdx.x * 40503u + 777u;     // wave-uniform
    int32_t ri[D]; double rv[D];
    double acc = 0;
    auto issue = [&](int u) {
        x = x * 1664525u + 1013904223u;
        const uint64_t c = ((uint64_t)x * nchunks) >> 32;
        ri[u] = idx[c * 64 + lane];
        rv[u] = VALS ? val[c * 64 + lane] : 0.0;
    };
#pragma unroll
    for (int u = 0; u < D; u++) issue(u);
    for (int it = 0; it < iters; it += D) {
#pragma unroll
        for (int u = 0; u < D; u++) { acc += ri[u] + rv[u]; issue(u); }
    }
    if (acc == 123.456) sink[0] = acc;
}

template <class K, class... Args>
static double time_kernel(K k, dim3 grid, dim3 block, size_t lds, Args... args) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, grid, block, lds, 0, args...);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k, grid, block, lds, 0, args...);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    double *sink; CK(hipMalloc(&sink, 8));
    const int ncu = 256, iters = 2000;
    const char *names[] = {"ds_or_b32 random", "ds_add_f64 random", "ds_read_b64 random", "ds_add_rtn_u32 random", "ds_write_b64 random",
                           "ds_add_f64 conflict-free", "ds_or_b32 conflict-free", "ds_add_f32 random", "ds_add_f32 conflict-free",
                           "ds_add_u32 random"};
    for (uint32_t span : {6144u, 1024u}) {
        double ms[10];
        ms[0] = time_kernel(lds_kernel<0>, dim3(ncu), dim3(1024), 65536, iters, span, sink);
        ms[1] = time_kernel(lds_kernel<1>, dim3(ncu), dim3(1024), 65536, iters, span, sink);
        ms[2] = time_kernel(lds_kernel<2>, dim3(ncu), dim3(1024), 65536, iters, span, sink);
        ms[3] = time_kernel(lds_kernel<3>, dim3(ncu), dim3(1024), 65536, iters, span, sink);
        ms[4] = time_kernel(lds_kernel<4>, dim3(ncu), dim3(1024), 65536, iters, span, sink);
        ms[5] = time_kernel(lds_kernel<5>, dim3(ncu), dim3(1024), 65536, iters, span, sink);
        ms[6] = time_kernel(lds_kernel<6>, dim3(ncu), dim3(1024), 65536, iters, span, sink);
        ms[7] = time_kernel(lds_kernel<7>, dim3(ncu), dim3(1024), 65536, iters, span, sink);
        ms[8] = time_kernel(lds_kernel<8>, dim3(ncu), dim3(1024), 65536, iters, span, sink);
        ms[9] = time_kernel(lds_kernel<9>, dim3(ncu), dim3(1024), 65536, iters, span, sink);
        for (int o = 0; o < 10; o++) {
            const double winst = 16.0 * iters * 8;                       // wave-instructions per CU
            printf("span %5u %-26s %8.3f ms  %7.1f wave-instr/us/CU  %6.1f ns per wave-instr\n", span, names[o], ms[o],
                   winst / (ms[o] * 1e3), ms[o] * 1e6 / winst);
        }
    }
    const uint64_t n = 256ull << 20;                                      // 256 Mi entries: 1 GiB of int32 + 2 GiB of fp64
    int32_t *idx; double *val;
    CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&val, n * 8));
    CK(hipMemset(idx, 0, n * 4)); CK(hipMemset(val, 0, n * 8));
    for (uint64_t span : {n / 64, (uint64_t)(4 << 20) / 64, (uint64_t)(64 << 20) / 64}) {
        const int it = 4096;
        const double bytes_i = 256.0 * ncu * 16 * it, bytes_v = 768.0 * ncu * 16 * it;
        double t;
        t = time_kernel(chunk_kernel<2, false>, dim3(ncu), dim3(1024), 0, idx, val, span, it, sink);
        printf("chunks over %6.0f Mi entries, idx only, D=2:  %7.3f ms %7.1f GB/s chip, %5.1f GB/s per CU\n", span * 64 / 1048576.0, t, bytes_i / t / 1e6, bytes_i / t / 1e6 / ncu);
        t = time_kernel(chunk_kernel<4, false>, dim3(ncu), dim3(1024), 0, idx, val, span, it, sink);
        printf("chunks over %6.0f Mi entries, idx only, D=4:  %7.3f ms %7.1f GB/s chip, %5.1f GB/s per CU\n", span * 64 / 1048576.0, t, bytes_i / t / 1e6, bytes_i / t / 1e6 / ncu);
        t = time_kernel(chunk_kernel<8, false>, dim3(ncu), dim3(1024), 0, idx, val, span, it, sink);
        printf("chunks over %6.0f Mi entries, idx only, D=8:  %7.3f ms %7.1f GB/s chip, %5.1f GB/s per CU\n", span * 64 / 1048576.0, t, bytes_i / t / 1e6, bytes_i / t / 1e6 / ncu);
        t = time_kernel(chunk_kernel<16, false>, dim3(ncu), dim3(1024), 0, idx, val, span, it, sink);
        printf("chunks over %6.0f Mi entries, idx only, D=16: %7.3f ms %7.1f GB/s chip, %5.1f GB/s per CU\n", span * 64 / 1048576.0, t, bytes_i / t / 1e6, bytes_i / t / 1e6 / ncu);
        t = time_kernel(chunk_kernel<4, true>, dim3(ncu), dim3(1024), 0, idx, val, span, it, sink);
        printf("chunks over %6.0f Mi entries, idx+val,  D=4:  %7.3f ms %7.1f GB/s chip, %5.1f GB/s per CU\n", span * 64 / 1048576.0, t, bytes_v / t / 1e6, bytes_v / t / 1e6 / ncu);
        t = time_kernel(chunk_kernel<8, true>, dim3(ncu), dim3(1024), 0, idx, val, span, it, sink);
        printf("chunks over %6.0f Mi entries, idx+val,  D=8:  %7.3f ms %7.1f GB/s chip, %5.1f GB/s per CU\n", span * 64 / 1048576.0, t, bytes_v / t / 1e6, bytes_v / t / 1e6 / ncu);
    }
    return 0;
}
