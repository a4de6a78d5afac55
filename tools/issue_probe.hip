// issue_probe.hip -- what non-VALU instructions cost a wave that has its SIMD to itself (the latency-bound tree levels):
// cycles per loop body of 8 dependent v_add_u32, alone and with s_mov_b32 / s_nop / DPP variants mixed in.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/issue_probe tools/issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int IT = 4096;

#define A8 "v_add_u32 %0, %0, %1\n\t" "v_add_u32 %0, %0, %1\n\t" "v_add_u32 %0, %0, %1\n\t" "v_add_u32 %0, %0, %1\n\t" \
           "v_add_u32 %0, %0, %1\n\t" "v_add_u32 %0, %0, %1\n\t" "v_add_u32 %0, %0, %1\n\t" "v_add_u32 %0, %0, %1\n\t"
#define AS(x) "v_add_u32 %0, %0, %1\n\t" x "\n\t"
#define A8S(x) AS(x) AS(x) AS(x) AS(x) AS(x) AS(x) AS(x) AS(x)
#define D "v_add_u32_dpp %0, %0, %1 row_ror:4 row_mask:0xf bank_mask:0x1\n\t"
#define D2 "v_add_u32_dpp %0, %2, %1 row_ror:4 row_mask:0xf bank_mask:0x1\n\t"

template <int V>
__global__ __launch_bounds__(256) void k(uint32_t* out, unsigned long long* rec, uint32_t b) {
    uint32_t a = threadIdx.x, c = threadIdx.x * 3;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < IT; ++i) {
        if (V == 0) asm volatile(A8 : "+v"(a) : "v"(b));
        if (V == 1) asm volatile(A8S("s_mov_b32 s20, 0x12345678") : "+v"(a) : "v"(b) : "s20");
        if (V == 2) asm volatile(A8S("s_nop 0") : "+v"(a) : "v"(b));
        if (V == 3) asm volatile(A8S("s_nop 1") : "+v"(a) : "v"(b));
        if (V == 4) asm volatile(D D D D D D D D : "+v"(a) : "v"(b));                       // dependent DPP chain, NO nops (wrong values, timing only)
        if (V == 5) asm volatile(D2 D2 D2 D2 D2 D2 D2 D2 : "+v"(a) : "v"(b), "v"(c));            // DPP source constant
        if (V == 6) asm volatile(A8S("v_add_u32 %0, 0x12345678, %0") : "+v"(a) : "v"(b));   // literal operand
        if (V == 7) asm volatile(A8S("v_add3_u32 %0, %0, %1, s20") : "+v"(a) : "v"(b));      // SGPR operand
        if (V == 8) asm volatile(A8S("s_add_u32 s20, s20, 1") : "+v"(a) : "v"(b) : "s20", "scc");
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = a;
    if ((threadIdx.x & 63) == 0) rec[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
}
template <int V>
int run(const char* what, int extra, uint32_t* d_out, unsigned long long* d_rec) {
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), 0, 0, d_out, d_rec, 7u);
    CHK(hipDeviceSynchronize());
    unsigned long long h[1024];
    CHK(hipMemcpy(h, d_rec, sizeof h, hipMemcpyDeviceToHost));
    double s = 0;
    for (int i = 0; i < 1024; ++i) s += (double)h[i];
    s /= 1024.0 * IT;
    printf("%-58s %6.1f cycles per body (8 v_add_u32%s) = %.2f per instruction%s\n", what, s, extra ? " + 8 others" : "", s / (8 + 0), extra ? "" : "");
    return 0;
}
int main() {
    uint32_t* d_out; unsigned long long* d_rec;
    CHK(hipMalloc(&d_out, 256 * 256 * 4)); CHK(hipMalloc(&d_rec, 1024 * 8));
    if (run<0>("8 dependent v_add_u32", 0, d_out, d_rec)) return 1;
    if (run<1>("... each followed by s_mov_b32 sN, literal", 1, d_out, d_rec)) return 1;
    if (run<2>("... each followed by s_nop 0", 1, d_out, d_rec)) return 1;
    if (run<3>("... each followed by s_nop 1", 1, d_out, d_rec)) return 1;
    if (run<4>("8 v_add_u32_dpp, source = previous result (no nops)", 0, d_out, d_rec)) return 1;
    if (run<5>("8 v_add_u32_dpp, source constant", 0, d_out, d_rec)) return 1;
    if (run<6>("... each followed by v_add_u32 with a literal", 1, d_out, d_rec)) return 1;
    if (run<7>("... each followed by v_add3_u32 with an SGPR", 1, d_out, d_rec)) return 1;
    if (run<8>("... each followed by s_add_u32", 1, d_out, d_rec)) return 1;
    return 0;
}
