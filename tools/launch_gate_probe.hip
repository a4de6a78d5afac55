// launch_gate_probe.hip -- how long does it take from "the host knows the challenge" to "the next kernel runs"?
//
// One proof is a chain of 16+ commitments: root on the host -> transcript -> challenge -> launch of the next layer's kernels
// (prover.rs:198-225).  docs/LOG.md (round 4) measured a pre-launched GATE KERNEL (one wave polling a host word in front of the
// real launch) and found no gain over a plain launch.  This probe measures the alternatives side by side, 300 repetitions each,
// median / p10 / p90 of the host-observed latency from the host's store (or launch call) to the kernel's first store arriving in
// host memory:
//   A  plain launch:            hipLaunchKernelGGL when the value is known (what the prover does)
//   B  stream wait-value:       hipStreamWaitValue32 on a host-written word + the kernel, both enqueued EARLY; the host only stores the word
//   C  gate kernel:             a one-wave kernel polling the word + the kernel behind it (round 4's experiment)
//   D  gate inside the kernel:  the kernel itself enqueued early, its first wave polls the word
// and the cost, for a full-chip launch (1024 workgroups), of taking a 4-byte parameter from host memory (coherent / non-coherent)
// instead of from the kernel arguments (B and D need the challenge to travel through memory).
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/launch_gate_probe tools/launch_gate_probe.hip && /tmp/launch_gate_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void done_kernel(volatile uint32_t* done, uint32_t seq) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { *done = seq; __threadfence_system(); }
}
// bounded: ~2 s of polling at most, then gives up (never parks a wave for ever)
__global__ void gate_kernel(volatile const uint32_t* gate, uint32_t seq) {
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (*gate != seq && __builtin_amdgcn_s_memrealtime() - t0 < 200000000ull) __builtin_amdgcn_s_sleep(1);
    }
}
__global__ void gated_done_kernel(volatile const uint32_t* gate, volatile uint32_t* done, uint32_t seq) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (*gate != seq && __builtin_amdgcn_s_memrealtime() - t0 < 200000000ull) __builtin_amdgcn_s_sleep(1);
        *done = seq; __threadfence_system();
    }
}
// a full-chip launch whose every wave needs a 4-byte parameter: from the arguments, or from memory
__global__ void __launch_bounds__(256) param_kernel(const uint32_t* src, uint32_t arg, uint32_t* out, volatile uint32_t* done, uint32_t seq, uint32_t* counter) {
    const uint32_t v = src ? *reinterpret_cast<const volatile uint32_t*>(src) : arg;
    out[blockIdx.x * 256 + threadIdx.x] = v + threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(counter, 1u) == gridDim.x - 1) { *counter = 0; *done = seq; __threadfence_system(); }
    }
}

// the same with an ORDINARY (cacheable) load: how kernel arguments themselves are read.  Freshness is checked on the host (out[] must
// hold the value written just before this launch was released).
__global__ void __launch_bounds__(256) param_plain_kernel(const uint32_t* __restrict__ src, uint32_t* out, volatile uint32_t* done, uint32_t seq, uint32_t* counter) {
    const uint32_t v = src[0];
    out[blockIdx.x * 256 + threadIdx.x] = v + threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(counter, 1u) == gridDim.x - 1) { *counter = 0; *done = seq; __threadfence_system(); }
    }
}

static void stats(const char* name, std::vector<double>& v) {
    std::sort(v.begin(), v.end());
    printf("%-58s median %6.2f us   p10 %6.2f   p90 %6.2f   min %6.2f\n", name, v[v.size() / 2], v[v.size() / 10], v[v.size() * 9 / 10], v[0]);
}
static bool wait_done(volatile uint32_t* done, uint32_t seq) {
    const double t0 = now_us();
    while (*done != seq) if (now_us() - t0 > 3e6) return false;
    return true;
}

int main() {
    CHECK(hipSetDevice(0));
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    uint32_t *h_gate, *h_done, *d_gate, *d_done;
    CHECK(hipHostMalloc((void**)&h_gate, 64, hipHostMallocMapped | hipHostMallocCoherent));
    CHECK(hipHostMalloc((void**)&h_done, 64, hipHostMallocMapped | hipHostMallocCoherent));
    CHECK(hipHostGetDevicePointer((void**)&d_gate, h_gate, 0));
    CHECK(hipHostGetDevicePointer((void**)&d_done, h_done, 0));
    *h_gate = 0; *h_done = 0;
    const int reps = 300;
    uint32_t seq = 0;
    std::vector<double> v;
    auto settle = []() { const double t = now_us(); while (now_us() - t < 150.0) {} };   // the early enqueue has reached the queue

    // A: plain launch
    v.clear();
    for (int i = 0; i < reps + 20; ++i) {
        ++seq; settle();
        const double t0 = now_us();
        hipLaunchKernelGGL(done_kernel, dim3(1), dim3(64), 0, st, d_done, seq);
        if (!wait_done(h_done, seq)) { fprintf(stderr, "A timed out\n"); return 1; }
        if (i >= 20) v.push_back(now_us() - t0);
    }
    stats("A  plain launch when the value is known", v);
    // A': the same onto a HOT queue: the previous kernel ended 8 us ago (what a commitment's host turn looks like), not 150 us ago
    for (double gap : {8.0, 20.0}) {
        v.clear();
        for (int i = 0; i < reps + 20; ++i) {
            ++seq;
            { const double t = now_us(); while (now_us() - t < gap) {} }
            const double t0 = now_us();
            hipLaunchKernelGGL(done_kernel, dim3(1), dim3(64), 0, st, d_done, seq);
            if (!wait_done(h_done, seq)) { fprintf(stderr, "A' timed out\n"); return 1; }
            if (i >= 20) v.push_back(now_us() - t0);
        }
        char nm[96];
        snprintf(nm, sizeof nm, "A' plain launch, previous kernel ended %.0f us ago", gap);
        stats(nm, v);
    }
    // B'': wait-value + kernel enqueued early, released 8 us after the previous kernel ended (the hot-queue counterpart of A')
    {
        v.clear();
        bool ok = true;
        for (int i = 0; i < reps + 20 && ok; ++i) {
            ++seq;
            if (hipStreamWaitValue32(st, d_gate, seq, hipStreamWaitValueEq, 0xFFFFFFFFu) != hipSuccess) { ok = false; break; }
            hipLaunchKernelGGL(done_kernel, dim3(1), dim3(64), 0, st, d_done, seq);
            { const double t = now_us(); while (now_us() - t < 8.0) {} }
            const double t0 = now_us();
            *reinterpret_cast<volatile uint32_t*>(h_gate) = seq;
            if (!wait_done(h_done, seq)) { fprintf(stderr, "B'' timed out\n"); return 1; }
            if (i >= 20) v.push_back(now_us() - t0);
        }
        if (ok) stats("B\" wait-value + kernel enqueued early, released 8 us after the previous kernel", v);
    }

    // B: hipStreamWaitValue32 on a host-written word, enqueued early
    int can = 0;
    (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    for (int kind = 0; kind < 2 && can; ++kind) {
        uint32_t* w_host = h_gate; void* w_dev = d_gate;
        uint64_t* sig = nullptr;
        if (kind == 1) {                                   // signal memory: what the API documents as the fast path
            if (hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory) != hipSuccess) { printf("B' signal memory: not available\n"); (void)hipGetLastError(); break; }
            w_host = reinterpret_cast<uint32_t*>(sig); w_dev = sig;
            *reinterpret_cast<volatile uint64_t*>(sig) = 0;
        }
        v.clear();
        bool ok = true;
        for (int i = 0; i < reps + 20 && ok; ++i) {
            ++seq;
            if (hipStreamWaitValue32(st, w_dev, seq, hipStreamWaitValueEq, 0xFFFFFFFFu) != hipSuccess) { printf("B hipStreamWaitValue32 failed: %s\n", hipGetErrorString(hipGetLastError())); ok = false; break; }
            hipLaunchKernelGGL(done_kernel, dim3(1), dim3(64), 0, st, d_done, seq);
            settle();
            const double t0 = now_us();
            *reinterpret_cast<volatile uint32_t*>(w_host) = seq;
            if (!wait_done(h_done, seq)) { fprintf(stderr, "B timed out (kind %d)\n", kind); ok = false; break; }
            if (i >= 20) v.push_back(now_us() - t0);
        }
        if (ok) stats(kind ? "B' wait-value on SIGNAL memory + kernel, enqueued early" : "B  wait-value on a host word + kernel, enqueued early", v);
        CHECK(hipStreamSynchronize(st));
    }

    // C: gate kernel + kernel, enqueued early
    v.clear();
    for (int i = 0; i < reps + 20; ++i) {
        ++seq;
        hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(64), 0, st, d_gate, seq);
        hipLaunchKernelGGL(done_kernel, dim3(1), dim3(64), 0, st, d_done, seq);
        settle();
        const double t0 = now_us();
        *reinterpret_cast<volatile uint32_t*>(h_gate) = seq;
        if (!wait_done(h_done, seq)) { fprintf(stderr, "C timed out\n"); return 1; }
        if (i >= 20) v.push_back(now_us() - t0);
    }
    stats("C  gate kernel + kernel, enqueued early (round 4)", v);

    // D: the gate inside the kernel
    v.clear();
    for (int i = 0; i < reps + 20; ++i) {
        ++seq;
        hipLaunchKernelGGL(gated_done_kernel, dim3(1), dim3(64), 0, st, d_gate, d_done, seq);
        settle();
        const double t0 = now_us();
        *reinterpret_cast<volatile uint32_t*>(h_gate) = seq;
        if (!wait_done(h_done, seq)) { fprintf(stderr, "D timed out\n"); return 1; }
        if (i >= 20) v.push_back(now_us() - t0);
    }
    stats("D  gate inside the kernel, enqueued early", v);

    // the parameter through memory: full-chip launch, time from launch to the last workgroup's flag
    uint32_t *d_out, *d_counter, *d_param, *h_nc, *d_nc;
    CHECK(hipMalloc((void**)&d_out, 1024 * 256 * 4));
    CHECK(hipMalloc((void**)&d_counter, 4));
    CHECK(hipMalloc((void**)&d_param, 4));
    CHECK(hipMemset(d_counter, 0, 4));
    CHECK(hipHostMalloc((void**)&h_nc, 64, hipHostMallocMapped | hipHostMallocNonCoherent));
    CHECK(hipHostGetDevicePointer((void**)&d_nc, h_nc, 0));
    const char* names[4] = {"P  1024-workgroup launch, parameter in the kernel arguments", "P  ... parameter read from device memory",
                            "P  ... parameter read from COHERENT host memory", "P  ... parameter read from NON-COHERENT host memory"};
    for (int kind = 0; kind < 4; ++kind) {
        v.clear();
        for (int i = 0; i < reps + 20; ++i) {
            ++seq; settle();
            *h_gate = seq; *h_nc = seq;
            const uint32_t* src = kind == 0 ? nullptr : kind == 1 ? d_param : kind == 2 ? d_gate : d_nc;
            const double t0 = now_us();
            hipLaunchKernelGGL(param_kernel, dim3(1024), dim3(256), 0, st, src, seq, d_out, d_done, seq, d_counter);
            if (!wait_done(h_done, seq)) { fprintf(stderr, "P timed out\n"); return 1; }
            if (i >= 20) v.push_back(now_us() - t0);
        }
        stats(names[kind], v);
    }
    // Q: ordinary loads of a parameter block in host memory (coherent / non-coherent), plain launch and pre-enqueued behind a wait-value;
    //    the value changes every repetition and the first output word is checked against it (a stale cache line would show)
    uint32_t* h_out;
    CHECK(hipHostMalloc((void**)&h_out, 64));
    for (int kind = 0; kind < 4 && can; ++kind) {
        const bool noncoh = kind & 1, early = kind & 2;
        uint32_t* hp = noncoh ? h_nc : h_gate + 8;             // the parameter word (coherent: another word of the gate's line + 32 B)
        const uint32_t* dp = noncoh ? d_nc : d_gate + 8;
        v.clear();
        int stale = 0;
        for (int i = 0; i < reps + 20; ++i) {
            ++seq;
            if (early) {
                if (hipStreamWaitValue32(st, d_gate, seq, hipStreamWaitValueEq, 0xFFFFFFFFu) != hipSuccess) { printf("Q wait-value failed\n"); break; }
                hipLaunchKernelGGL(param_plain_kernel, dim3(1024), dim3(256), 0, st, dp, d_out, d_done, seq, d_counter);
            }
            settle();
            const double t0 = now_us();
            *reinterpret_cast<volatile uint32_t*>(hp) = seq * 7u + 1u;                    // the "challenge"
            __sync_synchronize();
            if (early) *reinterpret_cast<volatile uint32_t*>(h_gate) = seq;
            else hipLaunchKernelGGL(param_plain_kernel, dim3(1024), dim3(256), 0, st, dp, d_out, d_done, seq, d_counter);
            if (!wait_done(h_done, seq)) { fprintf(stderr, "Q timed out\n"); return 1; }
            if (i >= 20) v.push_back(now_us() - t0);
            CHECK(hipMemcpy(h_out, d_out, 8, hipMemcpyDeviceToHost));
            if (h_out[0] != seq * 7u + 1u) ++stale;
        }
        char name[160];
        snprintf(name, sizeof name, "Q  1024 wgs, ORDINARY load from %s host memory, %s [stale %d]", noncoh ? "NON-COHERENT" : "COHERENT", early ? "wait-value, early" : "plain launch", stale);
        stats(name, v);
    }
    CHECK(hipStreamSynchronize(st));
    printf("done\n");
    return 0;
}
