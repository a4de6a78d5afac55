// bank_probe.hip -- does the VGPR bank (register number mod 4) of the three sources of a VOP3 instruction matter?
// 64 independent instructions per loop body with hand-picked registers; 1 and 4 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/bank_probe tools/bank_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int IT = 2048;
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define CLOB "v20","v21","v22","v23","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15"
template <int V>
__global__ __launch_bounds__(256) void k(uint32_t* out, unsigned long long* rec) {
    asm volatile("v_mov_b32 v4, 1\n v_mov_b32 v5, 2\n v_mov_b32 v6, 3\n v_mov_b32 v7, 4\n v_mov_b32 v8, 5\n v_mov_b32 v9, 6\n v_mov_b32 v10, 7\n v_mov_b32 v11, 8\n"
                 "v_mov_b32 v12, 9\n v_mov_b32 v13, 10\n v_mov_b32 v14, 11\n v_mov_b32 v15, 12\n" ::: CLOB);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < IT; ++i) {
        if (V == 0) asm volatile(R16("v_add3_u32 v20, v4, v8, v12\n v_add3_u32 v21, v4, v8, v12\n v_add3_u32 v22, v4, v8, v12\n v_add3_u32 v23, v4, v8, v12\n") ::: CLOB);     // three sources in bank 0
        if (V == 1) asm volatile(R16("v_add3_u32 v20, v5, v10, v15\n v_add3_u32 v21, v5, v10, v15\n v_add3_u32 v22, v5, v10, v15\n v_add3_u32 v23, v5, v10, v15\n") ::: CLOB);  // banks 1, 2, 3
        if (V == 2) asm volatile(R16("v_add3_u32 v20, v4, v8, v13\n v_add3_u32 v21, v4, v8, v13\n v_add3_u32 v22, v4, v8, v13\n v_add3_u32 v23, v4, v8, v13\n") ::: CLOB);     // two in bank 0
        if (V == 3) asm volatile(R16("v_alignbit_b32 v20, v4, v4, v8\n v_alignbit_b32 v21, v4, v4, v8\n v_alignbit_b32 v22, v4, v4, v8\n v_alignbit_b32 v23, v4, v4, v8\n") ::: CLOB);   // x, x, amount same bank
        if (V == 4) asm volatile(R16("v_alignbit_b32 v20, v4, v4, v9\n v_alignbit_b32 v21, v4, v4, v9\n v_alignbit_b32 v22, v4, v4, v9\n v_alignbit_b32 v23, v4, v4, v9\n") ::: CLOB);   // amount other bank
        if (V == 5) asm volatile(R16("v_alignbit_b32 v20, v4, v4, 7\n v_alignbit_b32 v21, v4, v4, 7\n v_alignbit_b32 v22, v4, v4, 7\n v_alignbit_b32 v23, v4, v4, 7\n") ::: CLOB);     // inline constant amount
        if (V == 6) asm volatile(R16("v_bitop3_b32 v20, v4, v8, v12 bitop3:0x96\n v_bitop3_b32 v21, v4, v8, v12 bitop3:0x96\n v_bitop3_b32 v22, v4, v8, v12 bitop3:0x96\n v_bitop3_b32 v23, v4, v8, v12 bitop3:0x96\n") ::: CLOB);
        if (V == 7) asm volatile(R16("v_bitop3_b32 v20, v5, v10, v15 bitop3:0x96\n v_bitop3_b32 v21, v5, v10, v15 bitop3:0x96\n v_bitop3_b32 v22, v5, v10, v15 bitop3:0x96\n v_bitop3_b32 v23, v5, v10, v15 bitop3:0x96\n") ::: CLOB);
        if (V == 8) asm volatile(R16("v_add_u32 v20, v4, v8\n v_add_u32 v21, v4, v8\n v_add_u32 v22, v4, v8\n v_add_u32 v23, v4, v8\n") ::: CLOB);       // VOP2, same bank
        if (V == 9) asm volatile(R16("v_add_u32 v20, v5, v10\n v_add_u32 v21, v5, v10\n v_add_u32 v22, v5, v10\n v_add_u32 v23, v5, v10\n") ::: CLOB);     // VOP2, different banks
        if (V == 10) asm volatile(R16("v_add_u32 v20, 0x12345678, v8\n v_add_u32 v21, 0x12345678, v8\n v_add_u32 v22, 0x12345678, v8\n v_add_u32 v23, 0x12345678, v8\n") ::: CLOB);   // literal
        if (V == 11) asm volatile(R16("v_add3_u32 v20, v5, v10, s20\n v_add3_u32 v21, v5, v10, s20\n v_add3_u32 v22, v5, v10, s20\n v_add3_u32 v23, v5, v10, s20\n") ::: CLOB);     // SGPR source
        if (V == 12) asm volatile(R16("v_add3_u32 v20, v20, v10, v15\n v_add3_u32 v20, v20, v10, v15\n v_add3_u32 v20, v20, v10, v15\n v_add3_u32 v20, v20, v10, v15\n") ::: CLOB);  // dependent chain
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    uint32_t a;
    asm volatile("v_mov_b32 %0, v20" : "=v"(a));
    out[blockIdx.x * 256 + threadIdx.x] = a;
    if ((threadIdx.x & 63) == 0) rec[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
}
template <int V>
int run(const char* what, uint32_t* d_out, unsigned long long* d_rec) {
    for (int wps : {1, 4}) {
        const int blocks = 256 * wps;
        hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d_out, d_rec);
        CHK(hipDeviceSynchronize());
        static unsigned long long h[4096];
        CHK(hipMemcpy(h, d_rec, (size_t)blocks * 4 * 8, hipMemcpyDeviceToHost));
        double s = 0;
        for (int i = 0; i < blocks * 4; ++i) s += (double)h[i];
        s /= (double)blocks * 4 * IT * 64;
        printf("%-44s %d wave(s)/SIMD: %5.2f cycles per instruction per wave = %5.2f per SIMD\n", what, wps, s, s / wps);
    }
    return 0;
}
int main() {
    uint32_t* d_out; unsigned long long* d_rec;
    CHK(hipMalloc(&d_out, 1024 * 256 * 4)); CHK(hipMalloc(&d_rec, 4096 * 8));
    if (run<0>("v_add3_u32, sources in banks 0 0 0", d_out, d_rec)) return 1;
    if (run<2>("v_add3_u32, sources in banks 0 0 1", d_out, d_rec)) return 1;
    if (run<1>("v_add3_u32, sources in banks 1 2 3", d_out, d_rec)) return 1;
    if (run<11>("v_add3_u32, banks 1 2 + SGPR", d_out, d_rec)) return 1;
    if (run<12>("v_add3_u32, dependent chain", d_out, d_rec)) return 1;
    if (run<3>("v_alignbit_b32 x, x, amount: banks 0 0 0", d_out, d_rec)) return 1;
    if (run<4>("v_alignbit_b32 x, x, amount: banks 0 0 1", d_out, d_rec)) return 1;
    if (run<5>("v_alignbit_b32 x, x, inline constant", d_out, d_rec)) return 1;
    if (run<6>("v_bitop3_b32, banks 0 0 0", d_out, d_rec)) return 1;
    if (run<7>("v_bitop3_b32, banks 1 2 3", d_out, d_rec)) return 1;
    if (run<8>("v_add_u32, banks 0 0", d_out, d_rec)) return 1;
    if (run<9>("v_add_u32, banks 1 2", d_out, d_rec)) return 1;
    if (run<10>("v_add_u32, literal + bank 0", d_out, d_rec)) return 1;
    return 0;
}
