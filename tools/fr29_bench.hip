// Latency and throughput of one 29-bit-limb Montgomery product (fr29.hpp fs_mul / fq29.hpp fq_mul) on gfx950:
// a dependent chain of CHAIN products per lane, with 1 .. 8 waves per SIMD and 1 .. 4 independent chains per lane.
//   hipcc -O3 --offload-arch=gfx950 -I../halo-accumulation_amd/csrc fr29_bench.hip -o fr29_bench && ./fr29_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "fr29.hpp"
using namespace halo;

constexpr int CHAIN = 256;
template <int ILP>
__global__ __launch_bounds__(256) void k_chain(const uint64_t *in, uint64_t *out) {
    uint32_t t = blockIdx.x * 256 + threadIdx.x;
    Fs<4> x[ILP];
    Fs<4> step = fs_load(in + 4 * (t & 255));
#pragma unroll
    for (int c = 0; c < ILP; c++) x[c] = fs_load(in + 4 * ((t + 17 * c) & 255));
#pragma unroll 1
    for (int i = 0; i < CHAIN; i++) {
#pragma unroll
        for (int c = 0; c < ILP; c++) x[c] = fs_widen<4>(fs_mul(x[c], step));
    }
    Fs<4> r = x[0];
#pragma unroll
    for (int c = 1; c < ILP; c++) r = fs_widen<4>(fs_tighten(fs_add(r, x[c])));
    fs_store(out + 4 * (size_t)t, r);
}

// the loop of k_powers: product, conversion to canonical words, coalesced store; MODE 0 = all, 1 = no store (fold into a
// register), 2 = no conversion either (product only)
template <int MODE>
__global__ __launch_bounds__(256) void k_powlike(const uint64_t *in, uint64_t *out, int E) {
    uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    uint32_t e = wave * (64u * (uint32_t)E) + lane;
    Fs<4> cur = fs_load(in + 4 * (lane & 255));
    Fs<4> step = fs_load(in + 4 * ((lane + 7) & 255));
    Fe sink = fe_zero();
#pragma unroll 1
    for (int k = 0; k < E; k++, e += 64) {
        if (MODE == 0) fs_store(out + 4 * (size_t)e, cur);
        if (MODE == 1) { Fe w = fs_to_fe(cur); for (int i = 0; i < 8; i++) sink.v[i] ^= w.v[i]; }
        cur = fs_widen<4>(fs_mul(cur, step));
    }
    if (MODE != 0) { for (int i = 0; i < 8; i++) sink.v[i] ^= cur.v[i]; if (sink.v[0] == 0x12345) fe_store(out + 4 * (size_t)(blockIdx.x * 256 + threadIdx.x), sink); }
}
template <int MODE>
static void run_pow(int E, size_t n, double ghz, const uint64_t *d_in, uint64_t *d_out) {
    int blocks = (int)(n / (64 * (size_t)E) / 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k_powlike<MODE>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, E);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int w = 0; w < 10; w++) hipLaunchKernelGGL(k_powlike<MODE>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, E);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("k_powers-like mode %d (0 all, 1 no store, 2 product only)  E %2d  n 2^20: %7.2f us per launch = %5.0f cycles per element-step per wave/SIMD\n", MODE, E,
           ms * 100.0, ms * 1e-4 * ghz * 1e9 / ((double)n / 64 / 1024));
}

template <int ILP>
static void run(int waves_per_simd, int cus, double ghz, const uint64_t *d_in, uint64_t *d_out) {
    int blocks = cus * waves_per_simd;  // 4 waves per block = 1 wave per SIMD per block-per-CU
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_chain<ILP>, dim3(blocks), dim3(256), 0, 0, d_in, d_out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k_chain<ILP>, dim3(blocks), dim3(256), 0, 0, d_in, d_out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    double cyc = ms * 1e-3 * ghz * 1e9;
    double per_wave_product = cyc / (CHAIN * ILP * waves_per_simd);  // SIMD cycles per wave-product (throughput view)
    printf("waves/SIMD %d  chains/lane %d : %8.3f ms  latency of one step of the loop %7.0f cycles  => %6.0f SIMD-cycles per wave-product\n",
           waves_per_simd, ILP, ms, cyc / CHAIN, per_wave_product);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    double ghz = prop.clockRate / 1e6;
    int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clock %.2f GHz; chain of %d products per lane\n", prop.name, cus, ghz, CHAIN);
    uint64_t *d_in, *d_out;
    hipMalloc(&d_in, 256 * 32);
    hipMalloc(&d_out, (size_t)cus * 8 * 256 * 32);
    uint64_t h[1024];
    for (int i = 0; i < 1024; i++) h[i] = 0x9E3779B97F4A7C15ull * (i + 1) >> 2;
    hipMemcpy(d_in, h, sizeof h, hipMemcpyHostToDevice);
    for (int w : {1, 2, 3, 4, 6, 8}) run<1>(w, cus, ghz, d_in, d_out);
    for (int w : {1, 2, 4}) run<2>(w, cus, ghz, d_in, d_out);
    for (int w : {1, 2}) run<4>(w, cus, ghz, d_in, d_out);
    uint64_t *d_big;
    hipMalloc(&d_big, (size_t)32 << 20);
    for (int E : {4, 16}) { run_pow<0>(E, 1 << 20, ghz, d_in, d_big); run_pow<1>(E, 1 << 20, ghz, d_in, d_big); run_pow<2>(E, 1 << 20, ghz, d_in, d_big); }
    return 0;
}
