// valu_microbench.hip -- measures the issue rate of the integer VALU ops SHA-256 is made of,
// to price the Merkle kernels against the right roofline (DESIGN.md "SHA-256 roofline").
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_microbench valu_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITER = 4096;
constexpr int ACC = 8;

template <int OP>
__global__ __launch_bounds__(256) void bench(uint32_t* out, uint32_t seed, unsigned long long* clk) {
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t a[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) a[i] = seed * (i + 1) + threadIdx.x;
    uint32_t k = seed ^ 0x9e3779b9u, m = seed + 77;
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) {
            if (OP == 0) a[i] = __builtin_amdgcn_alignbit(a[i], a[i], 7);           // v_alignbit_b32
            if (OP == 1) a[i] = a[i] + a[(i + 1) & (ACC - 1)] + k;                                         // v_add3_u32
            if (OP == 2) a[i] = __builtin_amdgcn_bitop3_b32(a[i], k, m, 0x96);        // v_bitop3_b32
            if (OP == 3) a[i] = a[i] + k;                                              // v_add_u32
            if (OP == 4) a[i] = a[i] ^ k;                                              // v_xor_b32
            if (OP == 5) a[i] = (uint32_t)(((uint64_t)a[i] * k) >> 32);               // v_mul_hi_u32
            if (OP == 6) a[i] = a[i] * k;                                              // v_mul_lo_u32
            if (OP == 7) { float f = __uint_as_float(a[i]); f = __builtin_fmaf(f, 1.0001f, 0.5f); a[i] = __float_as_uint(f); }  // v_fma_f32
            if (OP == 8) a[i] = (a[i] << 30) + a[i];                                   // v_lshl_add_u32
            if (OP == 9) a[i] = (a[i] >> 3) ^ k;                                       // shift + xor (2 ops or fused)
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < ACC; ++i) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 7) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

template <int OP>
int run(const char* name, uint32_t* d_out, int blocks, unsigned long long* d_clk) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, d_clk);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u + r, d_clk);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    double ops = 5.0 * (double)blocks * 256 * ITER * ACC;
    double tops = ops / (ms * 1e-3) / 1e12;
    unsigned long long h[2];
    CHK(hipMemcpy(h, d_clk, 16, hipMemcpyDeviceToHost));
    double ghz = (double)h[0] / (double)h[1] * 0.1;   // s_memrealtime ticks at 100 MHz
    printf("%-16s %8.3f ms  %7.2f T lane-ops/s  clock %.2f GHz  = %5.1f lanes/clk/CU; wave-cycles/instr at 8 waves/SIMD: %.2f\n", name, ms, tops, ghz,
           tops * 1e12 / 256 / (ghz * 1e9), (double)h[0] / ((double)ITER * ACC * 8));
    return 0;
}

int main() {
    int blocks = 256 * 8;   // 8 blocks of 256 per CU = 8 waves/SIMD
    uint32_t* d_out;
    CHK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
    unsigned long long* d_clk;
    CHK(hipMalloc(&d_clk, 16));
    run<0>("v_alignbit_b32", d_out, blocks, d_clk);
    run<1>("v_add3_u32", d_out, blocks, d_clk);
    run<2>("v_bitop3_b32", d_out, blocks, d_clk);
    run<3>("v_add_u32", d_out, blocks, d_clk);
    run<4>("v_xor_b32", d_out, blocks, d_clk);
    run<5>("v_mul_hi_u32", d_out, blocks, d_clk);
    run<6>("v_mul_lo_u32", d_out, blocks, d_clk);
    run<7>("v_fma_f32", d_out, blocks, d_clk);
    run<8>("v_lshl_add_u32", d_out, blocks, d_clk);
    run<9>("shr+xor", d_out, blocks, d_clk);
    return 0;
}
