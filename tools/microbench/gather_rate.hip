// Microbenchmark: how fast does a CU gather random 128-byte records from a cache-resident table,
// (a) each lane reading its own record in 16-byte pieces (what a per-lane BVH walk does), versus
// (b) eight consecutive lanes reading one record together (one 128-byte line per eight lanes)?
//   hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip && ./gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t rng(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

// (a) per-lane: PIECES x 16 bytes of one random record per lane and step
template <int PIECES>
__global__ void __launch_bounds__(256) k_per_lane(const float4 *__restrict__ table, uint32_t n_records, int steps, float *out) {
    uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    uint32_t rec = rng(s) % n_records;
    for (int i = 0; i < steps; i++) {
        const float4 *q = table + (size_t)rec * 8;
        float4 v[PIECES];
#pragma unroll
        for (int p = 0; p < PIECES; p++) v[p] = q[p];
        float t = 0.f;
#pragma unroll
        for (int p = 0; p < PIECES; p++) t += v[p].x + v[p].w;
        acc += t;
        // next record depends on the data (a dependent chain, like a tree walk)
        rec = (rng(s) + (uint32_t)(__float_as_uint(t) & 1u)) % n_records;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// (b) cooperative: the wave's 64 records of a step are read by 8 instructions, each covering 8 records with
// 8 lanes per record (one full 128-byte line per 8 lanes), staged through LDS, then every lane reads its own
// record back from LDS.
__global__ void __launch_bounds__(256) k_coop(const float4 *__restrict__ table, uint32_t n_records, int steps, float *out) {
    __shared__ float4 stage[4][64 * 8 + 8];
    uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    uint32_t rec = rng(s) % n_records;
    for (int i = 0; i < steps; i++) {
#pragma unroll
        for (int g = 0; g < 8; g++) {
            // record of lane (8g + lane/8), piece lane%8
            uint32_t r = __shfl(rec, 8 * g + (lane >> 3));
            float4 v = table[(size_t)r * 8 + (lane & 7)];
            stage[wave][(8 * g + (lane >> 3)) * 8 + (lane & 7)] = v;
        }
        float t = 0.f;
#pragma unroll
        for (int p = 0; p < 7; p++) {
            float4 v = stage[wave][lane * 8 + p];
            t += v.x + v.w;
        }
        acc += t;
        rec = (rng(s) + (uint32_t)(__float_as_uint(t) & 1u)) % n_records;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int blocks = cus * 4, steps = 2000;
    float *out;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    for (size_t mb : {1, 4, 15, 64, 512, 4096}) {
        size_t n_records = mb * 1024 * 1024 / 128;
        float4 *table;
        CHECK(hipMalloc(&table, n_records * 128));
        CHECK(hipMemset(table, 0, n_records * 128));
        auto run = [&](const char *name, auto launch, double pieces) {
            launch();
            CHECK(hipDeviceSynchronize());
            hipEvent_t a, b;
            CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
            CHECK(hipEventRecord(a));
            launch();
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, a, b));
            double recs = (double)blocks * 256 * steps;
            printf("table %5zu MB  %-22s %8.3f ms  %7.2f G records/s  %6.2f TB/s of pieces  %5.2f lane-pieces/clk/CU\n", mb, name, ms,
                   recs / ms / 1e6, recs * pieces * 16 / ms / 1e9, recs * pieces / (ms * 1e-3) / 2.4e9 / cus);
        };
        run("per-lane 7 pieces", [&] { hipLaunchKernelGGL(k_per_lane<7>, dim3(blocks), dim3(256), 0, 0, table, (uint32_t)n_records, steps, out); }, 7);
        run("per-lane 4 pieces", [&] { hipLaunchKernelGGL(k_per_lane<4>, dim3(blocks), dim3(256), 0, 0, table, (uint32_t)n_records, steps, out); }, 4);
        run("per-lane 2 pieces", [&] { hipLaunchKernelGGL(k_per_lane<2>, dim3(blocks), dim3(256), 0, 0, table, (uint32_t)n_records, steps, out); }, 2);
        run("per-lane 1 piece", [&] { hipLaunchKernelGGL(k_per_lane<1>, dim3(blocks), dim3(256), 0, 0, table, (uint32_t)n_records, steps, out); }, 1);
        run("cooperative 8 lanes", [&] { hipLaunchKernelGGL(k_coop, dim3(blocks), dim3(256), 0, 0, table, (uint32_t)n_records, steps, out); }, 8);
        CHECK(hipFree(table));
    }
    return 0;
}
