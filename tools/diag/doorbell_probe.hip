// doorbell_probe.hip — GPU box diagnostic (never in the product library): what a callback costs at the host boundary when
//   A  the kernel is launched per call and the host waits in hipStreamSynchronize               (what hipnlp_eval did in round 2)
//   B  the kernel is launched per call and the host polls a completion word in pinned memory that the kernel raises WITHOUT a
//      system-scope fence (every storing wave waits for its own stores, one lane per workgroup counts, the last one stores the word)
//   C  the kernel is RESIDENT: it waits for a doorbell word in pinned host memory (one leader workgroup polls over PCIe and
//      publishes the command to the others through device memory), works, raises the completion word as in B and waits again;
//      it leaves by itself after `idle` without a doorbell (every wave's wait is bounded by the real-time counter)
// around an 8 us stand-in for the knot program that reads x (2 x 151 KB) from pinned memory and stores 8 B / 371 KB / 1.47 MB of
// outputs straight into pinned memory.  Every variant is followed by a verification pass: the host checks EVERY output word of every
// call against the call's number (a completion word that overtakes its data shows up as a stale word).
//   hipcc --offload-arch=gfx950 -O3 -o tools/diag/_build/doorbell_probe tools/diag/doorbell_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); std::exit(1); } } while (0)

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

typedef unsigned long long u64;
constexpr u64 EXIT_WORD = ~0ull;

struct Cmd {            // pinned host memory, one 64-byte line: the host writes per_wg first, seq last
    u64 per_wg;
    u64 pad_[6];
    u64 seq;
};
struct Status {         // pinned host memory
    u64 done_seq;       // raised by the kernel: outputs of call `done_seq` are complete
    u64 pad0_[7];
    u64 exited;         // resident kernel: last + 1 when it has left (0 while it runs)
    u64 pad1_[7];
};
struct Ctl {            // device memory
    u64 go;             // leader -> workers: (seq << 24) | per_wg, or EXIT_WORD
    u64 pad0_[15];
    unsigned done;      // workgroups that have finished, cumulative
    unsigned pad1_[31];
};

__device__ __forceinline__ void stand_in(const double* x, int rd_per_wg, double* out, long per_wg, long ticks, u64 seq, double* sx, int mode = 0) {
    double acc = 0.0;
    if (rd_per_wg > 0) {   // the knot record and its halo, read over PCIe
        for (int i = threadIdx.x; i < rd_per_wg; i += blockDim.x) acc += x[size_t(blockIdx.x) * (rd_per_wg / 2) + i];
        sx[threadIdx.x] = acc;
        __syncthreads();
        acc = sx[(threadIdx.x + 1) & 511] * 0.0;
    }
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    while (long(__builtin_amdgcn_s_memrealtime() - t0) < ticks) { }
    double* o = out + size_t(blockIdx.x) * per_wg;
    if (mode == 3) {   // every output a system-scope (sc0 sc1) store
        for (long i = threadIdx.x; i < per_wg; i += blockDim.x) __hip_atomic_store(&o[i], acc + double(seq) * 4096.0 + double(i & 4095), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
        for (long i = threadIdx.x; i < per_wg; i += blockDim.x) o[i] = acc + double(seq) * 4096.0 + double(i & 4095);
    }
}

// Completion: when may the word that tells the host go out?  The outputs are stores to fine-grained host memory.
//   mode 0  every storing wave waits for its own stores (s_waitcnt vmcnt(0)), one lane per workgroup counts, the last one stores the
//           word — NO fence.  (Measured: the word overtakes the data, stale words on the host: kept as the negative control.)
//   mode 1  the same behind a system-scope release fence in every workgroup (buffer_wbl2 sc0 sc1)
//   mode 2  PCIe's own rule instead of a fence: a read cannot pass the posted writes in front of it.  Every WAVE reads one word of
//           host memory back behind its stores and waits for it, then counts as in mode 0
//   mode 4  as mode 2 with ONE read-back per workgroup (behind the workgroup barrier that follows every wave's vmcnt(0))
//   mode 3  every output stored with a system-scope (sc0 sc1) store, then as mode 0
__device__ __forceinline__ void complete(Ctl* ctl, Status* st, u64 seq, unsigned expected, int mode = 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (mode == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if (mode == 2 && (threadIdx.x & 63) == 0) {
        const u64 v = __hip_atomic_load(&st->pad0_[threadIdx.x >> 6 & 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (v == 12345) __hip_atomic_store(&st->pad1_[0], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (keeps the load)
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (mode == 4) {
            const u64 v = __hip_atomic_load(&st->pad0_[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (v == 12345) __hip_atomic_store(&st->pad1_[0], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        const unsigned old = __hip_atomic_fetch_add(&ctl->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == expected) __hip_atomic_store(&st->done_seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(512) void launch_kernel(const double* x, int rd_per_wg, double* out, long per_wg, long ticks, Ctl* ctl, Status* st, u64 seq,
                                                     unsigned expected, int mode) {
    __shared__ double sx[512];
    stand_in(x, rd_per_wg, out, per_wg, ticks, seq, sx, mode);
    if (st) complete(ctl, st, seq, expected, mode);
}

__global__ __launch_bounds__(512) void resident_kernel(const Cmd* cmd, Status* st, Ctl* ctl, const double* x, int rd_per_wg, double* out, long ticks,
                                                       u64 first_seq, long idle_ticks, long life_ticks, int mode) {
    __shared__ double sx[512];
    __shared__ u64 s_go;
    const bool leader = blockIdx.x == gridDim.x - 1;
    const u64 born = __builtin_amdgcn_s_memrealtime();
    u64 last = first_seq - 1, last_word = 0;
    unsigned rounds = 0;
    for (;;) {
        if (threadIdx.x == 0) {
            const u64 t0 = __builtin_amdgcn_s_memrealtime();
            u64 word;
            if (leader) {
                for (;;) {
                    const u64 s = __hip_atomic_load(&cmd->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (s == EXIT_WORD) { word = EXIT_WORD; break; }
                    if (s != last) {
                        const u64 per = __hip_atomic_load(&cmd->per_wg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (written before seq)
                        word = (s << 24) | per;
                        break;
                    }
                    const u64 now = __builtin_amdgcn_s_memrealtime();
                    if (long(now - t0) > idle_ticks || long(now - born) > life_ticks) { word = EXIT_WORD; break; }
                }
                __hip_atomic_store(&ctl->go, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                for (;;) {
                    word = __hip_atomic_load(&ctl->go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (word != last_word) break;
                    const u64 now = __builtin_amdgcn_s_memrealtime();
                    if (long(now - t0) > 2 * idle_ticks || long(now - born) > 2 * life_ticks) { word = EXIT_WORD; break; }   // (a leader that never ran)
                }
            }
            s_go = word;
        }
        __syncthreads();
        const u64 word = s_go;
        __syncthreads();
        if (word == EXIT_WORD) break;
        const u64 seq = word >> 24;
        const long per_wg = long(word & 0xFFFFFFull);
        stand_in(x, rd_per_wg, out, per_wg, ticks, seq, sx, mode);
        ++rounds;
        complete(ctl, st, seq, rounds * gridDim.x, mode);
        last = seq;
        last_word = word;
    }
    if (leader && threadIdx.x == 0) __hip_atomic_store(&st->exited, last + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static std::vector<double> g_samples;
template <class F> static double median_us(int reps, F&& f) {
    g_samples.clear();
    for (int i = 0; i < reps + 10; ++i) {
        const double a = now_us();
        f();
        const double b = now_us();
        if (i >= 10) g_samples.push_back(b - a);
    }
    std::sort(g_samples.begin(), g_samples.end());
    return g_samples[g_samples.size() / 2];
}
static double pct(double q) { return g_samples[size_t(q * (g_samples.size() - 1))]; }

int main() {
    const int REPS = 400, WGS = 101;
    std::setvbuf(stdout, nullptr, _IOLBF, 0);
    CK(hipSetDevice(0));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t NX = 18906, MAXOUT = 200000;
    double *h_x, *h_out, *user;
    Cmd* cmd;
    Status* st;
    Ctl* ctl;
    CK(hipHostMalloc(&h_x, (NX + 512) * 8, hipHostMallocDefault));
    CK(hipHostMalloc(&h_out, MAXOUT * 8, hipHostMallocDefault));
    CK(hipHostMalloc(&cmd, sizeof(Cmd), hipHostMallocDefault));
    CK(hipHostMalloc(&st, sizeof(Status), hipHostMallocDefault));
    CK(hipMalloc(&ctl, sizeof(Ctl)));
    user = static_cast<double*>(std::malloc((NX + 512) * 8));
    for (size_t i = 0; i < NX + 512; ++i) user[i] = double(i);
    const long TICKS = 800;   // 8 us of the 100 MHz real-time counter
    const long sizes[] = {1, 464, 1838};   // doubles per workgroup (x 101 workgroups)
    const char* names[] = {"f (8 B per workgroup: 0.8 KB)", "f + grad + g (371 KB)", "all four (1.47 MB)"};
    const int RD = 378;
    if (size_t(sizes[2]) * WGS > MAXOUT) { std::printf("output buffer too small\n"); return 1; }

    auto check = [&](long per, u64 seq) {   // every output word of the call
        long bad = 0;
        for (long w = 0; w < WGS; ++w)
            for (long i = 0; i < per; ++i) bad += h_out[w * per + i] != double(seq) * 4096.0 + double(i & 4095);
        return bad;
    };

    const int modes[] = {0, 1, 2, 4, 3};
    const char* mode_names[] = {"no fence (negative control)", "system release fence per workgroup", "read-back per wave", "sc0 sc1 stores", "read-back per workgroup"};
    for (int si = 0; si < 3; ++si) {
        const long per = sizes[si];
        std::printf("== outputs: %s, x (2 x 151 KB) read from pinned memory, 8 us stand-in program, %d workgroups of 512\n", names[si], WGS);
        // ---- A: launch + hipStreamSynchronize
        u64 seq = 0;
        std::memset(st, 0, sizeof(Status));
        std::printf("A launch + hipStreamSynchronize                          median %6.1f us", median_us(REPS, [&] {
            ++seq;
            std::memcpy(h_x, user, NX * 8);
            hipLaunchKernelGGL(launch_kernel, dim3(WGS), dim3(512), 0, s, h_x, RD, h_out, per, TICKS, ctl, (Status*)nullptr, seq, 0u, 0);
            CK(hipStreamSynchronize(s)); }));
        std::printf("   p10 %6.1f  p90 %6.1f\n", pct(0.1), pct(0.9));
        for (int mode : modes) {
            const char* mn = mode_names[mode];
            // ---- B: launch + completion word
            CK(hipMemset(ctl, 0, sizeof(Ctl)));
            std::memset(st, 0, sizeof(Status));
            seq = 0;
            unsigned launches = 0;
            long stale = 0;
            auto call_b = [&] {
                ++seq; ++launches;
                std::memcpy(h_x, user, NX * 8);
                hipLaunchKernelGGL(launch_kernel, dim3(WGS), dim3(512), 0, s, h_x, RD, h_out, per, TICKS, ctl, st, seq, launches * unsigned(WGS), mode);
                const double t0 = now_us();
                while (__atomic_load_n(&st->done_seq, __ATOMIC_ACQUIRE) != seq) { if (now_us() - t0 > 2e6) { std::printf("B: completion word never came\n"); std::exit(2); } }
            };
            const double mb = median_us(REPS, call_b);
            for (int i = 0; i < 300; ++i) { call_b(); stale += check(per, seq); }
            CK(hipStreamSynchronize(s));
            std::printf("B launch + completion word, %-36s median %6.1f us   p10 %6.1f  p90 %6.1f   stale words in 300 calls: %ld\n", mn, mb, pct(0.1), pct(0.9), stale);
            // ---- C: resident kernel + doorbell
            CK(hipMemset(ctl, 0, sizeof(Ctl)));
            std::memset(st, 0, sizeof(Status));
            std::memset(cmd, 0, sizeof(Cmd));
            seq = 0;
            const long IDLE = 2000000 /* 20 ms */, LIFE = 400000000 /* 4 s */;
            hipLaunchKernelGGL(resident_kernel, dim3(WGS), dim3(512), 0, s, cmd, st, ctl, h_x, RD, h_out, TICKS, u64(1), IDLE, LIFE, mode);
            CK(hipGetLastError());
            bool dead = false;
            auto call_c = [&] {
                ++seq;
                std::memcpy(h_x, user, NX * 8);
                cmd->per_wg = u64(per);
                __atomic_store_n(&cmd->seq, seq, __ATOMIC_RELEASE);
                const double t0 = now_us();
                while (__atomic_load_n(&st->done_seq, __ATOMIC_ACQUIRE) != seq) {
                    if (__atomic_load_n(&st->exited, __ATOMIC_ACQUIRE) != 0 || now_us() - t0 > 2e6) { dead = true; break; }
                }
            };
            const double med = median_us(REPS, [&] { if (!dead) call_c(); });
            const double p10 = pct(0.1), p90 = pct(0.9);
            stale = 0;
            for (int i = 0; i < 300 && !dead; ++i) { call_c(); stale += check(per, seq); }
            std::printf("C resident + doorbell,      %-36s median %6.1f us   p10 %6.1f  p90 %6.1f   stale words in 300 calls: %ld%s\n", mn, med, p10, p90, stale,
                        dead ? "   (KERNEL LEFT EARLY)" : "");
            if (!dead && mode == 2) {   // the doorbell after 1 ms of host-side silence (IPOPT's linear algebra between callbacks)
                const double m2 = median_us(50, [&] { const double t0 = now_us(); while (now_us() - t0 < 1000.0) { } call_c(); });
                std::printf("C the same after 1 ms of silence (minus the 1 ms)        median %6.1f us\n", m2 - 1000.0);
            }
            __atomic_store_n(&cmd->seq, EXIT_WORD, __ATOMIC_RELEASE);
            const double t0 = now_us();
            CK(hipStreamSynchronize(s));
            if (mode == 2) std::printf("C resident kernel left %.1f us after the exit word (exited = %llu, calls = %llu)\n", now_us() - t0, st->exited, seq);
        }
    }
    // ---- the resident kernel leaves by itself: no doorbell for `idle`
    {
        CK(hipMemset(ctl, 0, sizeof(Ctl)));
        std::memset(st, 0, sizeof(Status));
        std::memset(cmd, 0, sizeof(Cmd));
        const double t0 = now_us();
        hipLaunchKernelGGL(resident_kernel, dim3(WGS), dim3(512), 0, s, cmd, st, ctl, h_x, RD, h_out, TICKS, u64(1), 100000L /* 1 ms */, 400000000L, 2);
        CK(hipStreamSynchronize(s));
        std::printf("idle exit: a resident kernel with a 1 ms idle limit and no doorbell left after %.0f us (exited = %llu)\n", now_us() - t0, st->exited);
    }
    return 0;
}
