// valu_microbench.hip -- issue rate of the 32-bit VALU ops SHA-256 and the field arithmetic are made of, at
// 1 / 2 / 4 / 8 resident waves per SIMD, to fix the denominator of the Merkle kernels' VALU roofline
// (64 lanes per clock per CU).
//
// For every op and residency it prints
//   * the chip-level rate from wall time (HIP events) and the in-kernel clock (s_memtime / s_memrealtime):
//     lanes per clock per CU -- the roofline figure;
//   * the cycles one wave needs per instruction (its s_memtime lifetime / instructions, median over waves): 4 when
//     a wave has a SIMD to itself, 4 x residency when the SIMD is shared and a wave64 op holds it for 4 cycles;
//   * the MEASURED residency: every wave records the SIMD it ran on (HW_ID / XCC_ID registers); the table shows
//     how many SIMDs were used and the most waves any SIMD held.
// v_pk_fma_f32 is listed to show where AMD's 157 TFLOP/s fp32 vector figure comes from (two lanes' worth per
// lane per issue): no packed form exists for the 32-bit integer ops, so for them 64 lanes/clk/CU is the peak.
//
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_microbench valu_microbench.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITER = 128;    // outer iterations (a loop branch each: amortised over UNROLL * ACC = 256 VALU instructions)
constexpr int UNROLL = 32;
constexpr int ACC = 8;

struct WaveRec { unsigned long long cycles, real; uint32_t hw_id, xcc_id; };

typedef float float2v __attribute__((ext_vector_type(2)));

// Every measured instruction is inline assembly: the compiler can neither fold a chain of rotates / adds / xors into
// one operation nor re-associate it (an earlier version of this file measured folded loops).  ACC independent
// dependency chains per lane; each instruction reads its own chain's value and a second runtime register.
template <int OP>
__global__ __launch_bounds__(256) void bench(uint32_t* out, uint32_t seed, WaveRec* rec) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t a[ACC], b[ACC];
    float2v pf[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) { a[i] = seed * (i + 1) + threadIdx.x; b[i] = (seed ^ 0x9e3779b9u) * (i + 3) + threadIdx.x * 7; pf[i] = float2v{(float)a[i], 1.0f}; }
    const float2v pk = float2v{1.0001f, 0.9999f}, pm = float2v{0.5f, 0.25f};
    const float fk = 1.0001f, fm = 0.5f;
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) {
            const uint32_t y = b[(i + u) & (ACC - 1)];
            if (OP == 0) asm volatile("v_alignbit_b32 %0, %1, %2, 7" : "=v"(a[i]) : "v"(a[i]), "v"(y));
            if (OP == 1) asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(y), "v"(b[i]));
            if (OP == 2) asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(a[i]) : "v"(a[i]), "v"(y), "v"(b[i]));
            if (OP == 3) asm volatile("v_add_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(y));
            if (OP == 4) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(y));
            if (OP == 5) asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(y));
            if (OP == 6) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(y));
            if (OP == 7) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(fk), "v"(fm));
            if (OP == 8) asm volatile("v_lshl_add_u32 %0, %1, 30, %2" : "=v"(a[i]) : "v"(a[i]), "v"(y));
            if (OP == 9) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(pf[i]) : "v"(pf[i]), "v"(pk), "v"(pm));
        }
      }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < ACC; ++i) r ^= a[i] ^ __float_as_uint(pf[i].x) ^ __float_as_uint(pf[i].y);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        WaveRec w;
        w.cycles = c1 - c0; w.real = r1 - r0;
        w.hw_id = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));       // HW_REG_HW_ID, 32 bits
        w.xcc_id = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));     // HW_REG_XCC_ID (gfx94x+)
        rec[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = w;
    }
}

template <int OP>
int run(const char* name, int lanes_per_op, uint32_t* d_out, WaveRec* d_rec, int cus) {
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = cus * wps;             // a 256-thread block = one wave per SIMD of one CU
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, d_rec);
        CHK(hipDeviceSynchronize());
        const int reps = 5;
        CHK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u + r, d_rec);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<WaveRec> h((size_t)blocks * 4);
        CHK(hipMemcpy(h.data(), d_rec, h.size() * sizeof(WaveRec), hipMemcpyDeviceToHost));
        std::vector<double> cpi, ghz;
        std::map<uint64_t, int> per_simd;
        for (const WaveRec& w : h) {
            cpi.push_back((double)w.cycles / ((double)ITER * UNROLL * ACC));
            ghz.push_back((double)w.cycles / (double)w.real * 0.1);            // s_memrealtime ticks at 100 MHz
            // HW_ID (gfx9 layout): simd_id [5:4], cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID [3:0]
            const uint64_t key = ((uint64_t)(w.xcc_id & 0xF) << 16) | (w.hw_id & 0xFF30u);
            per_simd[key] += 1;
        }
        std::sort(cpi.begin(), cpi.end());
        std::sort(ghz.begin(), ghz.end());
        int max_res = 0;
        for (auto& kv : per_simd) max_res = std::max(max_res, kv.second);
        const double clock = ghz[ghz.size() / 2];
        const double ops = (double)reps * blocks * 256.0 * ITER * UNROLL * ACC * lanes_per_op;
        const double tops = ops / (ms * 1e-3) / 1e12;
        printf("%-15s %d waves/SIMD launched | %7.3f ms %6.2f T lane-ops/s  clock %.2f GHz  %6.1f lanes/clk/CU | wave cycles/instr %5.2f | SIMDs used %4zu, most waves on one SIMD %d\n",
               name, wps, ms / reps, tops, clock, tops * 1e12 / cus / (clock * 1e9), cpi[cpi.size() / 2], per_simd.size(), max_res);
    }
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s: %d CUs, max clock %.2f GHz\n", prop.name, cus, prop.clockRate / 1e6);
    uint32_t* d_out;
    CHK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * 4));
    WaveRec* d_rec;
    CHK(hipMalloc(&d_rec, (size_t)cus * 8 * 4 * sizeof(WaveRec)));
    run<0>("v_alignbit_b32", 1, d_out, d_rec, cus);
    run<1>("v_add3_u32", 1, d_out, d_rec, cus);
    run<2>("v_bitop3_b32", 1, d_out, d_rec, cus);
    run<3>("v_add_u32", 1, d_out, d_rec, cus);
    run<4>("v_xor_b32", 1, d_out, d_rec, cus);
    run<5>("v_mul_hi_u32", 1, d_out, d_rec, cus);
    run<6>("v_mul_lo_u32", 1, d_out, d_rec, cus);
    run<8>("v_lshl_add_u32", 1, d_out, d_rec, cus);
    run<7>("v_fma_f32", 1, d_out, d_rec, cus);
    run<9>("v_pk_fma_f32", 2, d_out, d_rec, cus);
    return 0;
}
