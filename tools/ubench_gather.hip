// ubench_gather.hip -- what does an 8-byte gather cost on gfx950, as a function of how the
// 64 lane addresses of one wave-instruction spread over cache lines?  Standalone:
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_gather.hip -o gpurun_out/ubench_gather && ./gpurun_out/ubench_gather
// Every wave issues ITER x 8 independent gathers from a table small enough to live in L2
// (or L1), so the rate is set by the address/L1 pipeline, not by HBM.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r)); exit(1); } } while (0)

template <typename T>
__global__ __launch_bounds__(256) void gather_kernel(const T *__restrict__ x, const unsigned *__restrict__ idx,
                                                     T *__restrict__ out, int iters, unsigned mask) {
    // idx holds, per (iteration-slot j, lane), a table index; 8 slots are used per iteration
    const int lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    unsigned rot = wave * 977u;
    T acc = 0;
    unsigned my[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) my[j] = idx[j * 64 + lane];
    for (int it = 0; it < iters; ++it) {
        T v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned off = ((my[j] + rot) & mask) * (unsigned)sizeof(T);
            v[j] = *reinterpret_cast<const T *>(reinterpret_cast<const char *>(x) + off);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
        rot += 4099u * 16u;  // move the whole pattern, keep its shape (multiple of 16 elements)
    }
    if (acc == T(12345.678)) out[0] = acc;
}

template <typename T>
double run(const char *name, const std::vector<unsigned> &pattern, size_t table_elems, int waves_per_cu) {
    T *x; unsigned *idx; T *out;
    CHECK(hipMalloc(&x, table_elems * sizeof(T)));
    CHECK(hipMemset(x, 0, table_elems * sizeof(T)));
    CHECK(hipMalloc(&idx, pattern.size() * 4));
    CHECK(hipMemcpy(idx, pattern.data(), pattern.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&out, 64));
    const int iters = 2000;
    const int blocks = 256 * waves_per_cu / 4;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    gather_kernel<T><<<blocks, 256>>>(x, idx, out, 50, (unsigned)table_elems - 1);
    CHECK(hipEventRecord(a));
    gather_kernel<T><<<blocks, 256>>>(x, idx, out, iters, (unsigned)table_elems - 1);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double instr_per_cu = (double)iters * 8 * waves_per_cu;  // wave-instructions per CU
    const double cyc = ms * 1e-3 * 2.4e9 / instr_per_cu;           // at a nominal 2.4 GHz
    printf("%-44s %2zu B  table %7.0f KiB  %2d waves/CU: %6.1f cyc/wave-instr  (%.2f lanes/clk/CU)\n", name,
           sizeof(T), table_elems * sizeof(T) / 1024.0, waves_per_cu, cyc, 64.0 / cyc);
    hipFree(x); hipFree(idx); hipFree(out);
    return cyc;
}

int main(int argc, char **argv) {
    srand(1);
    auto make = [](auto f) { std::vector<unsigned> p(8 * 64); for (int j = 0; j < 8; ++j) for (int l = 0; l < 64; ++l) p[j * 64 + l] = f(j, l); return p; };
    if (argc > 1 && argv[1][0] == 'b') {
        // "big": 64 different random lines per wave-instruction from tables that leave L2 (4 MiB per XCD)
        // and then the 256 MiB Infinity Cache -- the access pattern of the uniform half of the power-law
        // matrix's columns (BASELINE config 5: x is 64 MB of fp32)
        auto rnd = make([](int, int) { return ((unsigned)rand() << 16) ^ (unsigned)rand(); });
        for (int wpc : {16, 32}) {
            printf("---- %d waves per CU, 64 random lines per wave-instruction\n", wpc);
            for (size_t mib : {1, 4, 16, 64, 256, 1024}) {
                char name[64];
                snprintf(name, sizeof name, "float: random in %zu MiB", mib);
                run<float>(name, rnd, mib << 18, wpc);
            }
            run<double>("double: random in 64 MiB", rnd, (size_t)64 << 17, wpc);
        }
        return 0;
    }
    const size_t L2 = 1u << 18;   // 2 MiB of doubles: L2 resident
    const size_t L1 = 1u << 11;   // 16 KiB of doubles: L1 resident
    for (int wpc : {8, 16, 32}) {
        printf("---- %d waves per CU\n", wpc);
        run<double>("contiguous (lane i -> elem i)", make([](int j, int l) { return (unsigned)(j * 64 + l); }), L2, wpc);
        run<double>("stride 2 elems (16 B)", make([](int j, int l) { return (unsigned)(j * 128 + 2 * l); }), L2, wpc);
        run<double>("stride 8 elems (64 B: one 64B line each)", make([](int j, int l) { return (unsigned)(j * 512 + 8 * l); }), L2, wpc);
        run<double>("stride 16 elems (128 B: one 128B line each)", make([](int j, int l) { return (unsigned)(j * 1024 + 16 * l); }), L2, wpc);
        run<double>("stride 16, table in L1", make([](int j, int l) { return (unsigned)((j * 1024 + 16 * l) & 2047); }), L1, wpc);
        run<double>("random in 2 MiB", make([](int, int) { return (unsigned)rand(); }), L2, wpc);
        run<double>("random in 16 KiB (L1)", make([](int, int) { return (unsigned)rand(); }), L1, wpc);
        run<double>("4 lanes per 64B line (pairs of pairs)", make([](int j, int l) { return (unsigned)(j * 1024 + (l / 4) * 64 + (l % 4)); }), L2, wpc);
        run<double>("8 lanes per 64B line, lines far apart", make([](int j, int l) { return (unsigned)(j * 8192 + (l / 8) * 1024 + (l % 8)); }), L2, wpc);
        run<double>("stencil-like: 28 clusters x 2.3 lanes", make([](int j, int l) { int e = 2 * l + (j & 1); int row = e / 28, k = e % 28; return (unsigned)(k * 4000 + row + j * 3); }), L2, wpc);
        run<float>("float: contiguous", make([](int j, int l) { return (unsigned)(j * 64 + l); }), L2, wpc);
        run<float>("float: random in 1 MiB", make([](int, int) { return (unsigned)rand(); }), L2, wpc);
    }
    return 0;
}
