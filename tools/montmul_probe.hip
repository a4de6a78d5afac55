// montmul_probe.hip -- which instruction sequence should mont_mul (field.hpp) compile to on gfx950?
//   v0: the 64-bit product form (hipcc emits v_mad_u64_u32 + v_lshl_add + v_mul_hi + a 64-bit compare: 8 VALU)
//   v2: v_mad_u64_u32 + v_lshl_add + v_mul_hi + v_sub_co + v_add + v_cndmask (6 VALU: what field.hpp now compiles to)
//   v1: v_mul_lo + v_mul_hi + v_lshl_add + v_mul_hi + v_sub_co + v_add + v_cndmask (7 VALU, borrow from the subtract)
// and the issue rate of the single ops involved (v_mad_u64_u32, v_cmp_lt_u64, v_sub_co_u32, v_cndmask_b32).
// Four independent chains per lane so that dependent-issue latency does not hide the instruction cost; 10 launches
// back to back (steady state), 1/2/4/8 waves per SIMD.  Results must agree bit for bit (checked on the host).
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/montmul_probe tools/montmul_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr uint32_t P = 3221225473u;
constexpr int ITER = 512, UNROLL = 8;

__device__ __forceinline__ uint32_t mm0(uint32_t a, uint32_t b) {
    uint64_t t = (uint64_t)a * b;
    uint32_t lo = (uint32_t)t, hi = (uint32_t)(t >> 32);
    uint32_t m = lo + (lo << 30);
    uint32_t mp_hi = (uint32_t)(((uint64_t)m * P) >> 32);
    uint32_t r = hi - mp_hi;
    return hi < mp_hi ? r + P : r;
}
__device__ __forceinline__ uint32_t mm1(uint32_t a, uint32_t b) {
    uint32_t lo = a * b, hi = __umulhi(a, b);
    uint32_t m = lo + (lo << 30);
    uint32_t mp_hi = __umulhi(m, P);
    uint32_t r;
    bool borrow = __builtin_sub_overflow(hi, mp_hi, &r);
    return borrow ? r + P : r;
}

__device__ __forceinline__ uint32_t mm2(uint32_t a, uint32_t b) {   // field.hpp as of round 3: mad_u64 + borrow from the subtract
    uint64_t t = (uint64_t)a * b;
    uint32_t lo = (uint32_t)t, hi = (uint32_t)(t >> 32);
    uint32_t m = lo + (lo << 30);
    uint32_t mp_hi = __umulhi(m, P);
    uint32_t r;
    bool borrow = __builtin_sub_overflow(hi, mp_hi, &r);
    return borrow ? r + P : r;
}

template <int V>
__global__ __launch_bounds__(256) void chain(uint32_t* out, uint32_t seed) {
    uint32_t a = (seed + threadIdx.x) % P, b = (seed * 3 + threadIdx.x + 1) % P, c = (seed * 5 + 7 + blockIdx.x) % P, d = (seed * 7 + 11) % P;
    const uint32_t w = (seed * 977 + 5) % P;
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (V == 0) { a = mm0(a, b); b = mm0(b, c); c = mm0(c, d); d = mm0(d, w ^ (uint32_t)u); }
            else if (V == 1) { a = mm1(a, b); b = mm1(b, c); c = mm1(c, d); d = mm1(d, w ^ (uint32_t)u); }
            else             { a = mm2(a, b); b = mm2(b, c); c = mm2(c, d); d = mm2(d, w ^ (uint32_t)u); }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d;
}

// single-op issue rates (inline assembly: nothing folds)
template <int OP>
__global__ __launch_bounds__(256) void op_rate(uint32_t* out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c = seed * 5 + 2, d = seed * 7 + 3;
    unsigned long long q0 = seed, q1 = seed + 1, q2 = seed + 2, q3 = seed + 3;
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (OP == 0) {   // v_mad_u64_u32 (4 independent)
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(q0) : "v"(a), "v"(b) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(q1) : "v"(b), "v"(c) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(q2) : "v"(c), "v"(d) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(q3) : "v"(d), "v"(a) : "vcc");
            } else if (OP == 1) {   // v_mul_hi_u32
                asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b));
                asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(b) : "v"(b), "v"(c));
                asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(c) : "v"(c), "v"(d));
                asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(d) : "v"(d), "v"(a));
            } else if (OP == 2) {   // v_cmp_lt_u64
                asm volatile("v_cmp_lt_u64 vcc, %0, %1" :: "v"(q0), "v"(q1) : "vcc");
                asm volatile("v_cmp_lt_u64 vcc, %0, %1" :: "v"(q1), "v"(q2) : "vcc");
                asm volatile("v_cmp_lt_u64 vcc, %0, %1" :: "v"(q2), "v"(q3) : "vcc");
                asm volatile("v_cmp_lt_u64 vcc, %0, %1" :: "v"(q3), "v"(q0) : "vcc");
            } else if (OP == 3) {   // v_sub_co_u32
                asm volatile("v_sub_co_u32 %0, vcc, %1, %2" : "=v"(a) : "v"(a), "v"(b) : "vcc");
                asm volatile("v_sub_co_u32 %0, vcc, %1, %2" : "=v"(b) : "v"(b), "v"(c) : "vcc");
                asm volatile("v_sub_co_u32 %0, vcc, %1, %2" : "=v"(c) : "v"(c), "v"(d) : "vcc");
                asm volatile("v_sub_co_u32 %0, vcc, %1, %2" : "=v"(d) : "v"(d), "v"(a) : "vcc");
            } else if (OP == 4) {   // v_cndmask_b32
                asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a) : "v"(a), "v"(b) : "vcc");
                asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(b) : "v"(b), "v"(c) : "vcc");
                asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(c) : "v"(c), "v"(d) : "vcc");
                asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(d) : "v"(d), "v"(a) : "vcc");
            } else {                // v_mul_lo_u32
                asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b));
                asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(b) : "v"(b), "v"(c));
                asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(c) : "v"(c), "v"(d));
                asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(d) : "v"(d), "v"(a));
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3);
}

template <typename F>
int timeit(const char* name, double units_per_lane, int cus, F launch) {
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = cus * wps, reps = 10;
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        launch(blocks, 1u);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch(blocks, 12345u + r);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s %d waves/SIMD: %7.3f ns per unit per SIMD\n", name, wps, ms * 1e6 / ((double)reps * wps * units_per_lane));
    }
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t* d_out;
    const size_t n = (size_t)cus * 8 * 256;
    CHK(hipMalloc(&d_out, n * 4));
    // equality of the two variants
    std::vector<uint32_t> r0(n), r1(n);
    hipLaunchKernelGGL(chain<0>, dim3(cus * 8), dim3(256), 0, 0, d_out, 99u);
    CHK(hipMemcpy(r0.data(), d_out, n * 4, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(chain<1>, dim3(cus * 8), dim3(256), 0, 0, d_out, 99u);
    CHK(hipMemcpy(r1.data(), d_out, n * 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> r2(n);
    hipLaunchKernelGGL(chain<2>, dim3(cus * 8), dim3(256), 0, 0, d_out, 99u);
    CHK(hipMemcpy(r2.data(), d_out, n * 4, hipMemcpyDeviceToHost));
    printf("variants agree: %s\n", (r0 == r1 && r0 == r2) ? "yes" : "NO");
    const double mm = (double)ITER * UNROLL * 4;
    timeit("mont_mul v0 (mad_u64)", mm, cus, [&](int b, uint32_t s) { hipLaunchKernelGGL(chain<0>, dim3(b), dim3(256), 0, 0, d_out, s); });
    timeit("mont_mul v1 (mul_lo/hi)", mm, cus, [&](int b, uint32_t s) { hipLaunchKernelGGL(chain<1>, dim3(b), dim3(256), 0, 0, d_out, s); });
    timeit("mont_mul v2 (mad_u64+sub_co)", mm, cus, [&](int b, uint32_t s) { hipLaunchKernelGGL(chain<2>, dim3(b), dim3(256), 0, 0, d_out, s); });
    timeit("v_mad_u64_u32", mm, cus, [&](int b, uint32_t s) { hipLaunchKernelGGL(op_rate<0>, dim3(b), dim3(256), 0, 0, d_out, s); });
    timeit("v_mul_hi_u32", mm, cus, [&](int b, uint32_t s) { hipLaunchKernelGGL(op_rate<1>, dim3(b), dim3(256), 0, 0, d_out, s); });
    timeit("v_mul_lo_u32", mm, cus, [&](int b, uint32_t s) { hipLaunchKernelGGL(op_rate<5>, dim3(b), dim3(256), 0, 0, d_out, s); });
    timeit("v_cmp_lt_u64", mm, cus, [&](int b, uint32_t s) { hipLaunchKernelGGL(op_rate<2>, dim3(b), dim3(256), 0, 0, d_out, s); });
    timeit("v_sub_co_u32", mm, cus, [&](int b, uint32_t s) { hipLaunchKernelGGL(op_rate<3>, dim3(b), dim3(256), 0, 0, d_out, s); });
    timeit("v_cndmask_b32", mm, cus, [&](int b, uint32_t s) { hipLaunchKernelGGL(op_rate<4>, dim3(b), dim3(256), 0, 0, d_out, s); });
    return (r0 == r1 && r0 == r2) ? 0 : 2;
}
