// How does a copy from PAGEABLE host memory behave on this box?  (a) does hipMemcpyAsync return before the copy is done,
// (b) do chunks on several streams run back to back, (c) does a copy proceed while a kernel fills the GPU,
// (d) what does pinning the caller's buffer cost (hipHostRegister), (e) a kernel reading registered host memory directly.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/h2d_probe tools/h2d_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_spin(unsigned long long cycles, int *sink) {
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) *sink = 1;
}
__global__ void k_sum(const uint4 *src, size_t n16, unsigned *out) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { uint4 v = src[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    atomicAdd(out, acc);
}
int main() {
    const size_t bytes = 32u << 20;
    char *h = (char *)malloc(bytes);
    memset(h, 1, bytes);
    char *d; int *sink; unsigned *out;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&out, 4));
    hipStream_t s[4]; for (auto &x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now(); CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s[0])); double t1 = now(); CK(hipStreamSynchronize(s[0])); double t2 = now();
        printf("(a) 32 MiB pageable, one call: returns after %.3f ms, done after %.3f ms\n", t1 - t0, t2 - t0);
    }
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now(), tr[4];
        for (int k = 0; k < 4; ++k) { CK(hipMemcpyAsync(d + k * (bytes / 4), h + k * (bytes / 4), bytes / 4, hipMemcpyHostToDevice, s[k])); tr[k] = now() - t0; }
        for (int k = 0; k < 4; ++k) CK(hipStreamSynchronize(s[k]));
        printf("(b) 4 x 8 MiB on 4 streams: calls return at %.3f %.3f %.3f %.3f ms, all done %.3f ms\n", tr[0], tr[1], tr[2], tr[3], now() - t0);
    }
    for (int rep = 0; rep < 3; ++rep) {
        // wall_clock64 ticks at 100 MHz: 200000 ticks = 2 ms
        k_spin<<<2048, 256, 0, s[1]>>>(200000ull, sink);
        double t0 = now(); CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s[0])); double t1 = now(); CK(hipStreamSynchronize(s[0])); double t2 = now();
        CK(hipStreamSynchronize(s[1])); double t3 = now();
        printf("(c) 32 MiB pageable beside a 2 ms kernel that fills the GPU: returns %.3f ms, copy done %.3f ms, kernel done %.3f ms\n", t1 - t0, t2 - t0, t3 - t0);
    }
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now(); CK(hipHostRegister(h, bytes, hipHostRegisterDefault)); double t1 = now();
        CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s[0])); double t2 = now(); CK(hipStreamSynchronize(s[0])); double t3 = now();
        void *dp = nullptr; CK(hipHostGetDevicePointer(&dp, h, 0));
        CK(hipMemsetAsync(out, 0, 4, s[0]));
        double t4 = now(); k_sum<<<1024, 256, 0, s[0]>>>((const uint4 *)dp, bytes / 16, out); CK(hipStreamSynchronize(s[0])); double t5 = now();
        CK(hipHostUnregister(h)); double t6 = now();
        printf("(d) hipHostRegister %.3f ms, async copy from it: returns %.3f, done %.3f ms; (e) kernel reading it in place %.3f ms; unregister %.3f ms\n",
               t1 - t0, t2 - t1, t3 - t1, t5 - t4, t6 - t5);
    }
    // (f) chunked: register + copy per 8 MiB chunk, pipelined by hand over 2 threads?  Only the plain numbers here.
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        for (int k = 0; k < 4; ++k) CK(hipMemcpyAsync(d + k * (bytes / 4), h + k * (bytes / 4), bytes / 4, hipMemcpyHostToDevice, s[0]));
        double t1 = now(); CK(hipStreamSynchronize(s[0]));
        printf("(f) 4 x 8 MiB on ONE stream: calls return after %.3f ms, done %.3f ms\n", t1 - t0, now() - t0);
    }
    // (g) the way back: 32 MiB device -> pageable host memory that has been touched / never been touched (a fresh Vec), and through
    // a pinned staging buffer + memcpy
    {
        char *pin; CK(hipHostMalloc(&pin, bytes));
        for (int rep = 0; rep < 3; ++rep) {
            double t0 = now(); CK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost)); double t1 = now();
            char *fresh = (char *)malloc(bytes);
            double t2 = now(); CK(hipMemcpy(fresh, d, bytes, hipMemcpyDeviceToHost)); double t3 = now();
            free(fresh);
            char *fresh2 = (char *)malloc(bytes);
            double t4 = now(); CK(hipMemcpy(pin, d, bytes, hipMemcpyDeviceToHost)); double t5 = now(); memcpy(fresh2, pin, bytes); double t6 = now();
            free(fresh2);
            char *fresh3 = (char *)calloc(bytes, 1);
            double t7 = now(); CK(hipMemcpy(fresh3, d, bytes, hipMemcpyDeviceToHost)); double t8 = now();
            free(fresh3);
            printf("(g) D2H 32 MiB: into touched pageable %.3f ms; into fresh malloc %.3f ms; pinned staging %.3f + memcpy into fresh malloc %.3f ms; into fresh calloc %.3f ms\n",
                   t1 - t0, t3 - t2, t5 - t4, t6 - t5, t8 - t7);
        }
    }
    return 0;
}
