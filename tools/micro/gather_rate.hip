// Scattered 4-byte / 8-byte gather rate on gfx950, the ceiling the hash grid's corner reads live under (roofline of k_render_frame_multi).
// Every lane reads table[a] at a pseudo-random element a; what is varied: the table size (L1-, L2-, Infinity-Cache-, HBM-resident), how many lanes of
// one wave-instruction share a 64-byte line (1 = every lane its own line, the fine hashed levels of ray-ordered training points; 4 / 16 = the 4x4-pixel
// tiles of a rendered frame on mid / coarse levels), the load width, and the waves per CU (12 = the frame kernel's occupancy, 32 = the chip's maximum).
//   hipcc --offload-arch=gfx950 -O3 gather_rate.hip -o gather_rate && ./gather_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// WIDTH = dwords per lane-load (1 or 2); SHARE = lanes per 64-byte line (1, 4, 16, 64)
template <int WIDTH, int SHARE>
__global__ __launch_bounds__(256) void k_gather(const uint32_t* __restrict__ table, uint32_t line_mask, uint32_t iters, uint32_t* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    // the line is chosen by the lane group, the element inside the line by the lane
    uint32_t s = (wave * 64u + lane / SHARE) * 0x9E3779B1u + 0x7F4A7C15u;
    const uint32_t within = (WIDTH == 1) ? (lane & 15u) : ((lane & 7u) * 2u);          // dword inside the 16-dword line
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        uint32_t v[8][WIDTH];
        #pragma unroll
        for (int j = 0; j < 8; j++) {
            s = s * 1664525u + 1013904223u;
            const uint32_t line = (s >> 8) & line_mask;
            const uint32_t* p = table + (size_t)line * 16u + within;
            if (WIDTH == 1) v[j][0] = *p;
            else { const uint2 q = *reinterpret_cast<const uint2*>(p); v[j][0] = q.x; v[j][WIDTH - 1] = q.y; }
        }
        #pragma unroll
        for (int j = 0; j < 8; j++) {
            #pragma unroll
            for (int w = 0; w < WIDTH; w++) acc ^= v[j][w];
        }
    }
    if (acc == 0x12345678u) out[0] = acc;                                              // never true for the table's contents; keeps the loads
}

template <int WIDTH, int SHARE>
static void run(const uint32_t* table, size_t table_bytes, int waves_per_cu, uint32_t* out) {
    const uint32_t lines = (uint32_t)(table_bytes / 64);
    const uint32_t mask = lines - 1;
    const int wgs = 256 * waves_per_cu / 4;
    const uint32_t iters = 96;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_gather<WIDTH, SHARE>), dim3(wgs), dim3(256), 0, 0, table, mask, 8u, out);   // warm the caches
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((k_gather<WIDTH, SHARE>), dim3(wgs), dim3(256), 0, 0, table, mask, iters, out);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double lane_loads = (double)wgs * 256 * iters * 8;
    const double rate = lane_loads / (best * 1e-3);
    printf("table %8.2f MiB  %d B/lane  %2d lanes/line  %2d waves/CU : %7.3f ms  %7.1f G lane-loads/s  %5.2f lanes/clk/CU (2.4 GHz)  %6.2f TB/s useful  %6.2f TB/s of 64-B lines\n",
           table_bytes / 1048576.0, 4 * WIDTH, SHARE, waves_per_cu, best, rate / 1e9, rate / 256 / 2.4e9, rate * 4 * WIDTH / 1e12, rate / SHARE * 64 / 1e12);
    CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
}

int main() {
    const size_t max_bytes = (size_t)1 << 30;
    uint32_t *table, *out;
    CHECK(hipMalloc(&table, max_bytes)); CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(table, 1, max_bytes)); CHECK(hipMemset(out, 0, 64));
    const size_t sizes[] = {(size_t)16 << 10, (size_t)2 << 20, (size_t)32 << 20, (size_t)1 << 30};
    for (int wpc : {12, 32}) {
        for (size_t sz : sizes) {
            run<1, 1>(table, sz, wpc, out);
            run<1, 4>(table, sz, wpc, out);
            run<1, 16>(table, sz, wpc, out);
            run<1, 64>(table, sz, wpc, out);
            run<2, 1>(table, sz, wpc, out);
            run<2, 4>(table, sz, wpc, out);
        }
    }
    CHECK(hipFree(table)); CHECK(hipFree(out));
    return 0;
}
