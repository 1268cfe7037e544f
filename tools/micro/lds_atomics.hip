// LDS atomic throughput on gfx950: ds_add_f32 / ds_add_u32 / ds_add_u64 / ds_pk_add_f16, no return, random rows of a 16 K-row table,
// one 1,024-thread workgroup per CU.   hipcc --offload-arch=gfx950 -O3 lds_atomics.hip -o lds_atomics && ./lds_atomics
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(1024) void k(const uint32_t* __restrict__ rows, float* out, int iters) {
    extern __shared__ float acc[];
    for (uint32_t i = threadIdx.x; i < 32768; i += 1024) acc[i] = 0.f;
    __syncthreads();
    const uint32_t* r = rows + (size_t)blockIdx.x * 1024 * 64 + threadIdx.x;
    uint32_t rw[64];
    #pragma unroll
    for (int j = 0; j < 64; j++) rw[j] = r[j * 1024];
    for (int it = 0; it < iters; it++) {
        #pragma unroll
        for (int j = 0; j < 64; j++) {
            const uint32_t a = (rw[j] + it * 977u) & 16383u;
            if (MODE == 0) {
                __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float*)(acc + 2 * a), 1.0f, 0, 0, false);
                __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float*)(acc + 2 * a + 1), 2.0f, 0, 0, false);
            } else if (MODE == 1) {
                atomicAdd((uint32_t*)acc + 2 * a, 1u);
                atomicAdd((uint32_t*)acc + 2 * a + 1, 2u);
            } else if (MODE == 2) {
                atomicAdd((unsigned long long*)acc + (a & 8191u), 3ull);
            } else if (MODE == 3) {
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                h2 v = {(_Float16)1.0f, (_Float16)2.0f};
                __builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) h2*)((uint32_t*)acc + a), v);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc[5];
}
int main() {
    const int nb = 256, iters = 16;
    uint32_t* rows; float* out;
    hipMalloc(&rows, (size_t)nb * 1024 * 64 * 4); hipMalloc(&out, nb * 4);
    uint32_t* h = (uint32_t*)malloc((size_t)nb * 1024 * 64 * 4);
    uint32_t s = 12345; for (size_t i = 0; i < (size_t)nb * 1024 * 64; i++) { s = s * 1664525u + 1013904223u; h[i] = (s >> 8) & 16383u; }
    hipMemcpy(rows, h, (size_t)nb * 1024 * 64 * 4, hipMemcpyHostToDevice);
    const char* names[4] = {"ds_add_f32 x2 per row", "ds_add_u32 x2 per row", "ds_add_u64 x1 per row", "ds_pk_add_f16 x1 per row"};
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 4; mode++) {
        void (*fn)(const uint32_t*, float*, int) = mode == 0 ? k<0> : mode == 1 ? k<1> : mode == 2 ? k<2> : k<3>;
        hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        hipLaunchKernelGGL(fn, dim3(nb), dim3(1024), 131072, 0, rows, out, 1);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(fn, dim3(nb), dim3(1024), 131072, 0, rows, out, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double rows_done = (double)nb * 1024 * 64 * iters;
        printf("%-26s %8.3f ms  %7.2f G rows/s chip  %6.2f clk per 64-lane row-instruction per CU (2.4 GHz): %s\n", names[mode], ms, rows_done / ms / 1e6,
               ms * 1e-3 * 2.4e9 / (1024.0 * 64 * iters / 64), hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
