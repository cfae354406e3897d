// Instruction-throughput microbenchmark for gfx950: which multiplier should a 255-bit
// Montgomery product be built from?  Prints wave-instructions per cycle per SIMD (assuming
// the clock reported by the device) for each candidate instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;

#define KERNEL(NAME, DECL, BODY, FOLD)                                              \
    __global__ __launch_bounds__(256) void NAME(uint64_t *out, uint32_t seed) {     \
        uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;                 \
        DECL;                                                                       \
        for (int i = 0; i < ITERS; i++) { BODY; }                                   \
        out[blockIdx.x * 256 + threadIdx.x] = FOLD;                                 \
    }

#define ACC64 uint64_t x0 = a, x1 = b, x2 = a + 1, x3 = b + 1, x4 = a + 2, x5 = b + 2, x6 = a + 3, x7 = b + 3
#define ACC32 uint32_t x0 = a, x1 = b, x2 = a + 1, x3 = b + 1, x4 = a + 2, x5 = b + 2, x6 = a + 3, x7 = b + 3
#define ACCF double x0 = a, x1 = b, x2 = a + 1, x3 = b + 1, x4 = a + 2, x5 = b + 2, x6 = a + 3, x7 = b + 3; double fa = (double)a * 1e-9, fb = (double)b
#define REP8(OP) OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7)
#define FOLD8 (uint64_t)(x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7)
#define FOLDF (uint64_t)(x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7)

#define OP_MAD64(x) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b) : "vcc");
#define OP_MULLO(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(a));
#define OP_MULHI(x) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(a));
#define OP_MAD24(x) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
#define OP_MULHI24(x) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x) : "v"(a));
#define OP_ADD32(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(a));
#define OP_ADDCO(x) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x) : "v"(a) : "vcc");
#define OP_LSHLADD64(x) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x) : "v"(x7));
#define OP_FMA64(x) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(x) : "v"(fa), "v"(fb));
#define OP_MOV(x) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(a));
#define OP_MAD32(x) asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
#define OP_DOT4(x) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));

KERNEL(k_mad64, ACC64, REP8(OP_MAD64), FOLD8)
KERNEL(k_mullo, ACC32, REP8(OP_MULLO), FOLD8)
KERNEL(k_mulhi, ACC32, REP8(OP_MULHI), FOLD8)
KERNEL(k_mad24, ACC32, REP8(OP_MAD24), FOLD8)
KERNEL(k_mulhi24, ACC32, REP8(OP_MULHI24), FOLD8)
KERNEL(k_add32, ACC32, REP8(OP_ADD32), FOLD8)
KERNEL(k_addco, ACC32, REP8(OP_ADDCO), FOLD8)
KERNEL(k_lshladd64, ACC64, REP8(OP_LSHLADD64), FOLD8)
KERNEL(k_fma64, ACCF, REP8(OP_FMA64), FOLDF)
KERNEL(k_mov, ACC32, REP8(OP_MOV), FOLD8)
KERNEL(k_mad16, ACC32, REP8(OP_MAD32), FOLD8)
KERNEL(k_dot4, ACC32, REP8(OP_DOT4), FOLD8)

typedef void (*kern_t)(uint64_t *, uint32_t);

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    double ghz = prop.clockRate / 1e6;
    int cus = prop.multiProcessorCount;
    printf("device %s  CUs %d  clock %.2f GHz\n", prop.name, cus, ghz);
    const int blocks = cus * 8;  // 8 waves per SIMD
    uint64_t *d;
    CHECK(hipMalloc(&d, (size_t)blocks * 256 * 8));
    struct { const char *name; kern_t k; int per_op; } ks[] = {
        {"v_mad_u64_u32", k_mad64, 1}, {"v_mul_lo_u32", k_mullo, 1}, {"v_mul_hi_u32", k_mulhi, 1},
        {"v_mad_u32_u24", k_mad24, 1}, {"v_mul_hi_u32_u24", k_mulhi24, 1}, {"v_add_u32", k_add32, 1},
        {"v_add_co+v_addc_co (pair)", k_addco, 1}, {"v_lshl_add_u64", k_lshladd64, 1}, {"v_fma_f64", k_fma64, 1},
        {"v_mov_b32", k_mov, 1}, {"v_mad_u32_u16", k_mad16, 1}, {"v_dot4_u32_u8", k_dot4, 1}};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (auto &k : ks) {
        hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, d, 1u);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, d, 2u + r);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        double wave_instrs = 5.0 * blocks * 4 * (double)ITERS * 8;  // 4 waves per block
        double per_simd_per_s = wave_instrs / (cus * 4) / (ms * 1e-3);
        printf("%-28s %8.3f ms   %.3f wave-instr/ns/SIMD   => %.2f cycles per wave-instr at %.2f GHz\n", k.name, ms / 5,
               per_simd_per_s * 1e-9, ghz / (per_simd_per_s * 1e-9), ghz);
    }
    return 0;
}
