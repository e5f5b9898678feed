// hipnlp_pose.hip — gfx950 kernel and the C-ABI (include/hipnlp.h, hipnlp_pose_*) of the static pose finder NLP
// (BASELINE config 2; turnkey_planners/humanoid_pose_finder/planner.py:323-788).
//
// One 256-thread workgroup (four role-specialised wavefronts, pose_body.h HIPNLP_POSE_PROGRAM) per pose: the 81 variables are
// gathered into a knot record in LDS (velocities zero), the kinematic tasks of knot_body.h and the pose tasks fill the native
// g / jac / grad slots in LDS, and the workgroup streams out the pose's CCS value run, its rows of g, grad f and the cost terms.
// The total cost of a pose is summed inside the same workgroup in a fixed order (bitwise reproducible); no second kernel.
#include <hip/hip_runtime.h>

#include <cstddef>
#include <type_traits>

#include <cmath>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

#include "pose_layout.h"

// (hipnlp.hip: hipMemcpy that survives a host buffer inside a stale self-registered range)
extern "C" hipError_t hipnlp_internal_memcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);

using namespace hipnlp;

namespace {

constexpr int WG = 256;
constexpr int XR_STRIDE = 64;
constexpr int POSE_MAX_NNZ = 1024;
constexpr int POSE_MAX_HNNZ = hs::COUNT;
constexpr int POSE_MAX_M = 256;   // one thread per row stages the multipliers (m = 89 .. 104 by mode)
static_assert(XR_COUNT <= XR_STRIDE, "pose reference record");

struct PoseTables {
    HeadTables head;
    int32_t g_row[gs::COUNT];
    int32_t jperm[POSE_MAX_NNZ];
    int32_t nnz, m;
    int32_t hperm[POSE_MAX_HNNZ];
    int32_t hnnz, pad_;
    PoseHands hands;
    int32_t row_slot[POSE_MAX_M];   // native g slot of row i (the inverse of g_row; -1 behind m): the Hessian kernel scatters the multipliers with it
};

struct PArgs {
    const PoseTables* tb;
    const double* x;    // [batch][81]
    const double* pk;   // [batch][PK_STRIDE]
    const double* xr;   // [batch][XR_STRIDE]
    const GParams* gp;  // [batch]
    double* f;          // [batch]           or null
    double* grad;       // [batch][81]       or null
    double* g;          // [batch][m]        or null
    double* jac;        // [batch][nnz]      or null
    double* cost_terms; // [batch][POSE_NCT]
    int32_t* flags;     // [batch]
    // Hessian kernel only
    const double* sigma;   // [batch]   objective factor
    const double* lambda;  // [batch][m]
    double* hess;          // [batch][hnnz]
    int32_t m;             // rows of one pose (a kernel argument: the address of a pose's multipliers does not wait for a table)
#ifdef HIPNLP_STAMPS
    unsigned long long* stamps;   // diagnostic build only (tools/diag/pose_stamps.py): [batch][4 waves][64] s_memtime words
#endif
};

constexpr int pose_jslots(int terrain) { return pjs::slots(terrain == HIPNLP_TERRAIN_PLANAR); }
template <int TERRAIN> struct PoseEm {
    static constexpr int kTerrain = TERRAIN;
    static constexpr bool kStatic = true;   // (knot_body.h em_static: the velocities of a pose are zero)
    using Scratch = KnotScratchT<LAYOUT_COMPACT, pose_jslots(TERRAIN), true>;   // (PoseScratchT below)
    double* g;
    double* jac;    // the staging (pose_body.h pjs): [per-point constants | kept global constants | varying regions]
    double* jacv;   // the same, moved back so that a native slot of the varying regions indexes it directly
    __device__ __forceinline__ void G(int slot, int, double v) { g[slot] = v; }
    __device__ __forceinline__ void J(int slot, int, int, double v) { jacv[slot] = v; }                 // (entries that depend on x: slots >= js::V0)
    __device__ __forceinline__ void JC(int slot, int, int, double v) { jac[pjs::index(slot)] = v; }     // (constants: any region)
    __device__ __forceinline__ void JD(int slot, int, int, double v) { jac[pjs::index(slot)] = v; }     // (the pose finder's own entries in slots of region D)
};

template <int TERRAIN> struct PoseHessEm {
    static constexpr int kTerrain = TERRAIN;
    static constexpr bool kStatic = true;
    using Scratch = KnotScratchT<LAYOUT_COMPACT_NOG, pose_jslots(TERRAIN), true>;   // (no row, no gradient entry is emitted: neither staging exists)
    double* g;
    double* jac;
    double* h;
    __device__ __forceinline__ void G(int slot, int, double v) { g[slot] = v; }
    __device__ __forceinline__ void J(int slot, int, int, double v) { jac[slot] = v; }
    __device__ __forceinline__ void H(int slot, int, int, double v) { h[slot] = v; }
};

// LDS of a pose workgroup: the COMPACT knot scratch (knot_body.h: own[] on the joint records, which are dead once the forward kinematics
// has read them) in its STATIC form (no velocity arrays, a 64-word reference record in place of the previous knot's), a Jacobian staging
// cut to the pieces the pose program touches (pose_body.h pjs), and the LITE tables of the four-wave callback kernels (the joint frames
// and link inertials, read once per pose, come from global memory) instead of the whole HeadTables: 31.8 KB on the planar terrain ->
// FIVE workgroups per CU (LDS is handed out in granules of 1 280 B: 25 granules each; 38.5 KB and four per CU until round 6).
template <int TERRAIN> using PoseScratchT = KnotScratchT<LAYOUT_COMPACT, pose_jslots(TERRAIN), true>;
static_assert(pose_jslots(HIPNLP_TERRAIN_SMOOTH_STEPS) != js::COUNT && pose_jslots(HIPNLP_TERRAIN_PLANAR) != js::vary_slots(false) && pose_jslots(HIPNLP_TERRAIN_SMOOTH_STEPS) != js::vary_slots(false),
              "pose scratch: trimmed (the hand tables live in the tables block), and hd[] (the hand buffer) must exist");

// the hand tables (224 B; read by several task groups of both kernels: an LDS copy, not a global load per use) live in the tables block (every
// pose scratch is a trimmed one since round 6; an untrimmed scratch would park them on the periodicity variables xo[] no pose task reads)
template <bool OWN_HANDS> struct PoseSharedT;
// (bad: the workgroup's non-finite vote — a word here instead of __syncthreads_or, whose reduction brings 256 B of LDS of its own)
template <> struct alignas(16) PoseSharedT<true> { KSettings ks; KinLite kt; GParamsLite gp; PoseHands hands; int32_t bad, pad_[3]; };
template <> struct alignas(16) PoseSharedT<false> { KSettings ks; KinLite kt; GParamsLite gp; int32_t bad, pad_[3]; };
// every wave ORs its vote into tabs.bad (cleared by pose_stage); returns the workgroup's vote in every thread
template <class T> __device__ __forceinline__ int pose_vote(T& tabs, int bad, int lane) {
    if (__any(bad) && lane == 0) tabs.bad = 1;   // (all writers store the same value)
    __syncthreads();
    return tabs.bad;
}
static_assert(sizeof(PoseHands) % 8 == 0 && sizeof(PoseHands) <= sizeof(double) * NPER, "staged as 8-byte words; fits on xo[]");
template <class S, class T> __device__ __forceinline__ PoseHands* pose_hands(S& s, T& tabs) {
    if constexpr (S::trimmed) return &tabs.hands;
    else return reinterpret_cast<PoseHands*>(s.xo);
}

// workgroup barrier that orders LDS traffic only (see hipnlp.hip): the copy-out tables prefetched into registers stay in flight across it
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// tables, parameters and the pose's variables -> LDS (ends with a workgroup barrier)
struct PoseStageNoMid { __device__ __forceinline__ void operator()() const {} };
template <class S, class T, class Mid = PoseStageNoMid> __device__ __forceinline__ void pose_stage(const PArgs& a, S& s, T& tabs, int b, int tid, Mid mid = Mid()) {
    const PoseTables& tb = *a.tb;
    // every global load in flight before the first LDS store waits for one (see hipnlp_knot_kernel).  Four pieces, as 8-byte words:
    // settings | lite kinematic tables (the KinLite prefix of the full ones) | lite parameters of this pose | hand tables
    constexpr int W0 = int(sizeof(KSettings) / 8), W1 = W0 + int(sizeof(KinLite) / 8), W2 = W1 + int(sizeof(GParamsLite) / 8), W3 = W2 + int(sizeof(PoseHands) / 8);
    constexpr int T_ITERS = (W3 + WG - 1) / WG;
    static_assert(sizeof(KSettings) % 8 == 0 && sizeof(KinLite) % 8 == 0 && sizeof(GParamsLite) % 8 == 0, "copied in 8-byte words");
    static_assert(S::xpad <= WG && PK_STRIDE <= WG && POSE_NX <= WG && POSE_MAX_NNZ % WG == 0 && XR_STRIDE <= S::xm_len, "one record word per thread");
    const double* src0 = reinterpret_cast<const double*>(&tb.head.ks);
    const double* src1 = reinterpret_cast<const double*>(static_cast<const KinLite*>(&tb.head.kt));
    const double* src2 = reinterpret_cast<const double*>(static_cast<const GParamsLite*>(a.gp + b));
    const double* src3 = reinterpret_cast<const double*>(&tb.hands);
    double* dst0 = reinterpret_cast<double*>(&tabs.ks);
    double* dst1 = reinterpret_cast<double*>(&tabs.kt);
    double* dst2 = reinterpret_cast<double*>(&tabs.gp);
    double* dst3 = reinterpret_cast<double*>(pose_hands(s, tabs));
    double tv[T_ITERS];
#pragma unroll
    for (int it = 0; it < T_ITERS; ++it) {
        const int i = tid + it * WG;
        tv[it] = i < W0 ? src0[i] : (i < W1 ? src1[i - W0] : (i < W2 ? src2[i - W1] : (i < W3 ? src3[i - W2] : 0.0)));
    }
    const double xrv = tid < XR_STRIDE ? a.xr[size_t(b) * XR_STRIDE + tid] : 0.0;
    const double pkv = tid < PK_STRIDE ? a.pk[size_t(b) * PK_STRIDE + tid] : 0.0;
    const double xval = tid < POSE_NX ? a.x[size_t(b) * POSE_NX + tid] : 0.0;
#pragma unroll
    for (int it = 0; it < T_ITERS; ++it) {
        const int i = tid + it * WG;
        if (i < W0) dst0[i] = tv[it]; else if (i < W1) dst1[i - W0] = tv[it]; else if (i < W2) dst2[i - W1] = tv[it]; else if (i < W3) dst3[i - W2] = tv[it];
    }
    if (tid < S::xpad) s.x[tid] = 0.0;
    if (tid < XR_STRIDE) s.xm[tid] = xrv;
    if (tid < PK_STRIDE) s.pk[tid] = pkv;
    if (tid == 0) tabs.bad = 0;
    __syncthreads();
    if (tid < POSE_NX) s.x[pose_to_knot_col(tid)] = xval;
    mid();   // (LDS stores of the caller that must follow stores it made before the call)
    __syncthreads();
}

// Planar terrain: FIVE workgroups per CU (<= 96 VGPRs: 90 used since the static kinematics; <= 32 000 B of LDS); smooth steps: four (123 VGPRs).
template <int TERRAIN> __global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(TERRAIN == HIPNLP_TERRAIN_PLANAR ? 5 : 4, TERRAIN == HIPNLP_TERRAIN_PLANAR ? 5 : 4)))
void hipnlp_pose_kernel(PArgs a) {
    using Scratch = PoseScratchT<TERRAIN>;
    static_assert(std::is_same_v<Scratch, typename PoseEm<TERRAIN>::Scratch>, "the emitter names the scratch the tasks run on");
    __shared__ Scratch s;
    __shared__ PoseSharedT<Scratch::trimmed> tabs;
    static_assert(sizeof(Scratch) + sizeof(PoseSharedT<Scratch::trimmed>) <= (TERRAIN == HIPNLP_TERRAIN_PLANAR ? 32000 : 40960), "five / four workgroups per CU");
#ifdef HIPNLP_STAMPS
    const unsigned long long st_entry = __builtin_amdgcn_s_memtime();
#endif
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = blockIdx.x;
    const PoseTables& tb = *a.tb;
    pose_stage(a, s, tabs, b, tid);
#ifdef HIPNLP_STAMPS
    const unsigned long long st_staged = __builtin_amdgcn_s_memtime();
    unsigned long long st_arr[8], st_dep[8], st_task[24];
    int st_nb = 0, st_nt = 0;
#endif
    // copy-out tables (CCS permutation, g row map): fetched now, consumed at the very end
    constexpr int JP_ITERS = (POSE_MAX_NNZ + WG - 1) / WG, GR_ITERS = (gs::COUNT + WG - 1) / WG;
    int32_t jp[JP_ITERS], gr[GR_ITERS];
#pragma unroll
    for (int it = 0; it < JP_ITERS; ++it) jp[it] = tb.jperm[tid + it * WG];          // (POSE_MAX_NNZ is a multiple of the workgroup)
#pragma unroll
    for (int it = 0; it < GR_ITERS; ++it) { const int slot = tid + it * WG; gr[it] = slot < gs::COUNT ? tb.g_row[slot] : -1; }
    const int nnz = tb.nnz, m = tb.m;

    KnotInfo ki{1, 3, 0, 0};   // "interior knot": the k >= 1 rows / costs of the shared tasks are active
    PoseEm<TERRAIN> em{s.g, s.jac, s.jac - (js::V0 - pjs::DSLOTS)};
    Ctx<PoseEm<TERRAIN>> cx(s, tabs.kt, tabs.ks, tabs.gp, ki, em, &tb.head.kt, a.gp + b);   // (full tables: global memory)
    cx.hands = pose_hands(s, tabs);
    // (one contiguous program instance per wave, as in hipnlp_knot_kernel: no wave jumps over the other waves' code)
#ifdef HIPNLP_STAMPS
#define DEV_R(w, w8, fn, nt) if constexpr ((w) == W) { for (int t_ = lane; t_ < (nt); t_ += 64) fn(cx, t_); if (st_nt < 24) st_task[st_nt++] = __builtin_amdgcn_s_memtime(); }
#define DEV_BARRIER st_arr[st_nb] = __builtin_amdgcn_s_memtime(); lds_barrier(); st_dep[st_nb] = __builtin_amdgcn_s_memtime(); st_nb++;
#elif defined(HIPNLP_TASK_MARKS)
    // diagnostic compile (assembly only, never a library): comment markers around every task group for tools/diag/isa_mix.py / reading the ISA
#define DEV_R(w, w8, fn, nt) if constexpr ((w) == W) { asm volatile("; TASK_BEGIN " #fn); for (int t_ = lane; t_ < (nt); t_ += 64) fn(cx, t_); asm volatile("; TASK_END " #fn); }
#define DEV_BARRIER lds_barrier();
#else
#define DEV_R(w, w8, fn, nt) if constexpr ((w) == W) { for (int t_ = lane; t_ < (nt); t_ += 64) fn(cx, t_); }
#define DEV_BARRIER lds_barrier();
#endif
    auto run_wave = [&](auto wc) __attribute__((always_inline)) {
        constexpr int W = decltype(wc)::value;
        HIPNLP_POSE_PROGRAM(DEV_R, DEV_BARRIER)
    };
    switch (wave) {
        case 0: run_wave(std::integral_constant<int, 0>{}); break;
        case 1: run_wave(std::integral_constant<int, 1>{}); break;
        case 2: run_wave(std::integral_constant<int, 2>{}); break;
        default: run_wave(std::integral_constant<int, 3>{}); break;
    }
#undef DEV_R
#undef DEV_BARRIER

    // copy-out: every LDS read in flight before the first store (as read -> wait -> store per piece, behind a branch each, the eight
    // pieces went one after the other: 2.3 k cycles at batch against 0.9 k in the Hessian kernel, which has two)
    int bad = 0;
    double jv[JP_ITERS], gv[GR_ITERS];
#pragma unroll
    for (int it = 0; it < JP_ITERS; ++it) jv[it] = s.jac[jp[it]];                    // (the table holds STAGING indices, pjs::index; behind nnz: 0)
#pragma unroll
    for (int it = 0; it < GR_ITERS; ++it) gv[it] = s.g[gr[it] >= 0 ? tid + it * WG : 0];
    const double gradv = s.grad[pose_to_knot_col(tid < POSE_NX ? tid : 0)];
    const double costv = pose_cost_terms_out(s)[tid < POSE_NCT ? tid : 0];           // (t_pose_cost_total: the nine terms side by side)
    const double tot = s.cost[CT_POSE_TOTAL];                                         // (and their sum, in their order)
    if (a.jac) {
        double* out = a.jac + size_t(b) * nnz;
#pragma unroll
        for (int it = 0; it < JP_ITERS; ++it) { const int e = tid + it * WG; if (e < nnz) { bad |= !isfinite(jv[it]); out[e] = jv[it]; } }
    }
    if (a.g) {
        double* out = a.g + size_t(b) * m;
#pragma unroll
        for (int it = 0; it < GR_ITERS; ++it) if (gr[it] >= 0) { bad |= !isfinite(gv[it]); out[gr[it]] = gv[it]; }
    }
    if (a.grad && tid < POSE_NX) { bad |= !isfinite(gradv); a.grad[size_t(b) * POSE_NX + tid] = gradv; }
    if (tid < POSE_NCT) a.cost_terms[size_t(b) * POSE_NCT + tid] = costv;
    if (tid == 0) {
        bad |= !isfinite(tot);
        if (a.f) a.f[b] = tot;
    }
    const int anybad = pose_vote(tabs, bad, lane);
    if (tid == 0) a.flags[b] = anybad;
#ifdef HIPNLP_STAMPS
    if (a.stamps && lane == 0) {   // [0] entry [1] staged [2] barriers [3] task groups [4] end | [8 + 2 i] arrival at / [9 + 2 i] departure from barrier i | [32 + t] end of task group t
        unsigned long long* o = a.stamps + (size_t(b) * 4 + wave) * 64;
        o[0] = st_entry; o[1] = st_staged; o[2] = st_nb; o[3] = st_nt; o[4] = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < st_nb; ++i) { o[8 + 2 * i] = st_arr[i]; o[9 + 2 * i] = st_dep[i]; }
        for (int i = 0; i < st_nt; ++i) o[32 + i] = st_task[i];
    }
#endif
}

// Exact Hessian of the Lagrangian (IPOPT eval_h, pose_hess_body.h): the pose program runs as in hipnlp_pose_kernel (its g / jac
// values stay in LDS, unused), the Hessian tasks run behind it and the workgroup streams out the lower-triangle CCS value run.
// The Hessian program runs none of the tasks that emit Jacobian entries: the jac staging area (the last 13.8 / 15.7 KB of the scratch) is
// not allocated.
// Workgroups per CU (round 6): the scratch ends in front of the gradient and carries no staging of g (the Hessian program emits neither), the
// (q_b, q_b) block is formed by three small task groups instead of one that held 124 registers: 31.0 KB and 92 VGPRs on the planar terrain ->
// FIVE workgroups per CU (25 granules of 1 280 B, 96 registers); the smooth terrain's kernel (158 VGPRs for the second-order jets) stays at three.
template <int TERRAIN> __global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(TERRAIN == HIPNLP_TERRAIN_PLANAR ? 5 : 3, TERRAIN == HIPNLP_TERRAIN_PLANAR ? 5 : 3)))
void hipnlp_pose_hess_kernel(PArgs a) {
    using Scratch = typename PoseHessEm<TERRAIN>::Scratch;
    static_assert(offsetof(Scratch, jac) + sizeof(Scratch::jac) + 16 > sizeof(Scratch) && offsetof(Scratch, grad) + sizeof(Scratch::grad) == offsetof(Scratch, jac),
                  "grad and jac are the last two members of the scratch (up to tail padding)");
    __shared__ alignas(16) double s_raw[offsetof(Scratch, grad) / sizeof(double)];
    Scratch& s = *reinterpret_cast<Scratch*>(s_raw);
    __shared__ PoseSharedT<Scratch::trimmed> tabs;
    __shared__ HessScratch hx;
#ifdef HIPNLP_STAMPS
    const unsigned long long st_entry = __builtin_amdgcn_s_memtime();
#endif
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = blockIdx.x;
    const PoseTables& tb = *a.tb;
    // the multiplier of every native g slot: thread i fetches the multiplier of ROW i and the slot that row lives in — two independent
    // loads, in flight together with the staging loads below — and scatters it between the two barriers of the staging, over zeros.
    // (Until round 6: slot -> row -> multiplier, two dependent round trips in front of the staging: 6.1 k cycles of a 21.8 k cycle pose.)
    const int m = a.m;
    double lam_i = 0.0, sig = 0.0;
    int slot_i = -1;
    if (tid < m) { lam_i = a.lambda[size_t(b) * m + tid]; slot_i = tb.row_slot[tid]; }
    if (tid == 0) sig = a.sigma[b];
    for (int slot = tid; slot < gs::COUNT; slot += WG) hx.lam[slot] = 0.0;
    pose_stage(a, s, tabs, b, tid, [&]() { if (slot_i >= 0) hx.lam[slot_i] = lam_i; if (tid == 0) hx.sigma = sig; });
#ifdef HIPNLP_STAMPS
    const unsigned long long st_staged = __builtin_amdgcn_s_memtime();
    unsigned long long st_arr[10], st_dep[10], st_task[24];
    int st_nb = 0, st_nt = 0;
#endif

    constexpr int HP_ITERS = (POSE_MAX_HNNZ + WG - 1) / WG;
    int32_t hp[HP_ITERS];
#pragma unroll
    for (int it = 0; it < HP_ITERS; ++it) { const int e = tid + it * WG; hp[it] = e < POSE_MAX_HNNZ ? tb.hperm[e] : 0; }   // (copy-out permutation, consumed at the end)
    const int hnnz = tb.hnnz;
    KnotInfo ki{1, 3, 0, 0};
    PoseHessEm<TERRAIN> em{nullptr, nullptr, hx.H};   // (no task of the Hessian program emits a row or a Jacobian entry)
    Ctx<PoseHessEm<TERRAIN>> cx(s, tabs.kt, tabs.ks, tabs.gp, ki, em, &tb.head.kt, a.gp + b);
    cx.hands = pose_hands(s, tabs);
    HCtx<PoseHessEm<TERRAIN>> hcx{cx, hx};
    // (one contiguous program instance per wave, as in hipnlp_knot_kernel: no wave jumps over the other waves' code)
#ifdef HIPNLP_STAMPS
#define DEV_KIN(w, fn, nt) if constexpr ((w) == W) { for (int t_ = lane; t_ < (nt); t_ += 64) fn(cx, t_); if (st_nt < 24) st_task[st_nt++] = __builtin_amdgcn_s_memtime(); }
#define DEV_RH(w, fn, nt) if constexpr ((w) == W) { for (int t_ = lane; t_ < (nt); t_ += 64) fn(hcx, t_); if (st_nt < 24) st_task[st_nt++] = __builtin_amdgcn_s_memtime(); }
#define DEV_BARRIER st_arr[st_nb] = __builtin_amdgcn_s_memtime(); lds_barrier(); st_dep[st_nb] = __builtin_amdgcn_s_memtime(); st_nb++;
#else
#define DEV_KIN(w, fn, nt) if constexpr ((w) == W) { for (int t_ = lane; t_ < (nt); t_ += 64) fn(cx, t_); }
#define DEV_RH(w, fn, nt) if constexpr ((w) == W) { for (int t_ = lane; t_ < (nt); t_ += 64) fn(hcx, t_); }
#define DEV_BARRIER lds_barrier();
#endif
    auto run_wave = [&](auto wc) __attribute__((always_inline)) {
        constexpr int W = decltype(wc)::value;
        HIPNLP_POSE_HESS_PROGRAM(DEV_KIN, DEV_RH, DEV_BARRIER)
    };
    switch (wave) {
        case 0: run_wave(std::integral_constant<int, 0>{}); break;
        case 1: run_wave(std::integral_constant<int, 1>{}); break;
        case 2: run_wave(std::integral_constant<int, 2>{}); break;
        default: run_wave(std::integral_constant<int, 3>{}); break;
    }
#undef DEV_KIN
#undef DEV_RH
#undef DEV_BARRIER
    int bad = 0;
    double* out = a.hess + size_t(b) * hnnz;
#pragma unroll
    for (int it = 0; it < HP_ITERS; ++it) { const int e = tid + it * WG; if (e < hnnz) { const double v = hx.H[hp[it]]; bad |= !isfinite(v); out[e] = v; } }
    const int anybad = pose_vote(tabs, bad, lane);
    if (tid == 0) a.flags[b] = anybad;
#ifdef HIPNLP_STAMPS
    if (a.stamps && lane == 0) {
        unsigned long long* o = a.stamps + (size_t(b) * 4 + wave) * 64;
        o[0] = st_entry; o[1] = st_staged; o[2] = st_nb; o[3] = st_nt; o[4] = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < st_nb; ++i) { o[8 + 2 * i] = st_arr[i]; o[9 + 2 * i] = st_dep[i]; }
        for (int i = 0; i < st_nt; ++i) o[32 + i] = st_task[i];
    }
#endif
}

thread_local std::string g_pose_create_error;

}  // namespace

struct hipnlp_pose_handle {
    hipnlp_pose_desc d;
    PoseLayout L;
    KinTables kt;
    int batch = 1, dev = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timing_valid = false, params_set = false, have_result = false;
    bool time_host = false;   // hipnlp_pose_set_host_timing: the host-buffer calls bracket their launch with events (each record drains the stream: ~3 us of a 25 us call)
    PoseTables* d_tb = nullptr;
    double *d_x = nullptr, *d_pk = nullptr, *d_xr = nullptr, *d_f = nullptr, *d_grad = nullptr, *d_g = nullptr, *d_jac = nullptr, *d_cost = nullptr;
    GParams* d_gp = nullptr;
    int32_t* d_flags = nullptr;
    double *h_x = nullptr, *h_f = nullptr, *h_grad = nullptr, *h_g = nullptr, *h_jac = nullptr, *h_cost = nullptr;
    // host-buffer path: the kernel reads x straight out of the pinned staging copy and stores its outputs straight into the pinned
    // blocks (device-visible addresses of the same memory: no copy command on either side of the launch, as in hipnlp_eval)
    double *hd_x = nullptr, *hd_f = nullptr, *hd_grad = nullptr, *hd_g = nullptr, *hd_jac = nullptr, *hd_cost = nullptr;
    int32_t* hd_flags = nullptr;
    double *hd_sigma = nullptr, *hd_lambda = nullptr, *hd_hess = nullptr;
    int32_t* h_flags = nullptr;
    // Hessian path (allocated on first use)
    double *d_sigma = nullptr, *d_lambda = nullptr, *d_hess = nullptr, *h_sigma = nullptr, *h_lambda = nullptr, *h_hess = nullptr;
    std::vector<double> p;
    std::string err;
#ifdef HIPNLP_STAMPS
    unsigned long long* d_stamps = nullptr;   // diagnostic build only
#endif
};

#define HIP_TRY(h, call)                                                                            \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                           \
            (void)hipGetLastError();                                                                \
            return HIPNLP_E_NODEVICE;                                                               \
        }                                                                                           \
    } while (0)

static void pose_free_all(hipnlp_pose_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->dev);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void* dptrs[] = {h->d_tb, h->d_x, h->d_pk, h->d_xr, h->d_f, h->d_grad, h->d_g, h->d_jac, h->d_cost, h->d_gp, h->d_flags, h->d_sigma, h->d_lambda, h->d_hess};
    for (void* q : dptrs) if (q) (void)hipFree(q);
    void* hptrs[] = {h->h_x, h->h_f, h->h_grad, h->h_g, h->h_jac, h->h_cost, h->h_flags, h->h_sigma, h->h_lambda, h->h_hess};
    for (void* q : hptrs) if (q) (void)hipHostFree(q);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

static int pose_launch(hipnlp_pose_handle* h, const double* x_dev, double* f_dev, double* grad_dev, double* g_dev, double* jac_dev, hipStream_t s,
                       bool timed, bool host_block = false) {
    PArgs a{};
    a.tb = h->d_tb; a.x = x_dev; a.pk = h->d_pk; a.xr = h->d_xr; a.gp = h->d_gp;
    a.f = f_dev; a.grad = grad_dev; a.g = g_dev; a.jac = jac_dev;
    a.cost_terms = host_block ? h->hd_cost : h->d_cost; a.flags = host_block ? h->hd_flags : h->d_flags;
#ifdef HIPNLP_STAMPS
    if (!h->d_stamps) { HIP_TRY(h, hipMalloc(&h->d_stamps, size_t(h->batch) * 256 * sizeof(unsigned long long))); HIP_TRY(h, hipMemset(h->d_stamps, 0, size_t(h->batch) * 256 * sizeof(unsigned long long))); }
    a.stamps = h->d_stamps;
#endif
    if (timed) HIP_TRY(h, hipEventRecord(h->ev0, s));   // host-buffer path only (an event record drains the stream)
    if (h->d.settings.terrain == HIPNLP_TERRAIN_PLANAR)
        hipLaunchKernelGGL(hipnlp_pose_kernel<HIPNLP_TERRAIN_PLANAR>, dim3(unsigned(h->batch)), dim3(WG), 0, s, a);
    else
        hipLaunchKernelGGL(hipnlp_pose_kernel<HIPNLP_TERRAIN_SMOOTH_STEPS>, dim3(unsigned(h->batch)), dim3(WG), 0, s, a);
    if (timed) HIP_TRY(h, hipEventRecord(h->ev1, s));
    HIP_TRY(h, hipGetLastError());
    if (timed) h->timing_valid = true;
    return HIPNLP_OK;
}

extern "C" {

const char* hipnlp_pose_last_error(const hipnlp_pose_handle* h) { return h ? h->err.c_str() : g_pose_create_error.c_str(); }

int hipnlp_pose_create(const hipnlp_pose_desc* desc, hipnlp_pose_handle** out) {
    if (!desc || !out) { g_pose_create_error = "null argument"; return HIPNLP_E_INVALID; }
    *out = nullptr;
    hipnlp_pose_handle* h = new (std::nothrow) hipnlp_pose_handle();
    if (!h) { g_pose_create_error = "out of memory"; return HIPNLP_E_ALLOC; }
    h->d = *desc;
    const hipnlp_pose_settings& st = desc->settings;
    auto fail = [&](int code, const std::string& msg) { g_pose_create_error = msg; pose_free_all(h); return code; };
    if (desc->abi_version != HIPNLP_ABI_VERSION)
        return fail(HIPNLP_E_INVALID, "hipnlp_pose_desc.abi_version is " + std::to_string(desc->abi_version) + ", this library implements HIPNLP_ABI_VERSION " +
                                          std::to_string(HIPNLP_ABI_VERSION) + " (the caller was built against another include/hipnlp.h, or left the field unset)");
    if (desc->flags != 0) return fail(HIPNLP_E_INVALID, "unknown bits in hipnlp_pose_desc.flags");
    if (const char* te = Layout::check_terrain(st.terrain, st.n_terrain_steps, st.terrain_steps)) return fail(HIPNLP_E_INVALID, te);
    for (int t : {st.com_position_type, st.left_point_position_type, st.right_point_position_type})
        if (t != HIPNLP_EXPR_SKIP && t != HIPNLP_EXPR_SUBJECT_TO && t != HIPNLP_EXPR_MINIMIZE) return fail(HIPNLP_E_INVALID, "bad expression type");
    if (desc->batch < 1) return fail(HIPNLP_E_INVALID, "batch must be >= 1");
    h->batch = desc->batch;
    std::string e;
    if (!Layout::make_kin_tables(desc->model, h->kt, e)) return fail(HIPNLP_E_INVALID, e);
    Layout::fill_terrain_tops(h->kt, st.terrain, st.n_terrain_steps, st.terrain_steps);
    if (!h->L.build(st, h->kt)) return fail(HIPNLP_E_INVALID, h->L.error);
    if (h->L.nnz > POSE_MAX_NNZ) return fail(HIPNLP_E_INVALID, "internal: pose pattern larger than POSE_MAX_NNZ");
    if (h->L.m > POSE_MAX_M) return fail(HIPNLP_E_INVALID, "internal: more rows than POSE_MAX_M");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(HIPNLP_E_NODEVICE, "no HIP device available (the engine has no CPU fallback)");
    if (desc->device < 0 || desc->device >= ndev) return fail(HIPNLP_E_INVALID, "bad device ordinal");
    h->dev = desc->device;
#define CREATE_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(HIPNLP_E_NODEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
    CREATE_TRY(hipSetDevice(h->dev));
    CREATE_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    CREATE_TRY(hipEventCreate(&h->ev0));
    CREATE_TRY(hipEventCreate(&h->ev1));
    const size_t B = size_t(h->batch), m = size_t(h->L.m), nnz = size_t(h->L.nnz);
    CREATE_TRY(hipMalloc(&h->d_tb, sizeof(PoseTables)));
    CREATE_TRY(hipMalloc(&h->d_x, B * POSE_NX * sizeof(double)));
    CREATE_TRY(hipMalloc(&h->d_pk, B * PK_STRIDE * sizeof(double)));
    CREATE_TRY(hipMalloc(&h->d_xr, B * XR_STRIDE * sizeof(double)));
    CREATE_TRY(hipMalloc(&h->d_gp, B * sizeof(GParams)));
    CREATE_TRY(hipMalloc(&h->d_f, B * sizeof(double)));
    CREATE_TRY(hipMalloc(&h->d_grad, B * POSE_NX * sizeof(double)));
    CREATE_TRY(hipMalloc(&h->d_g, B * m * sizeof(double)));
    CREATE_TRY(hipMalloc(&h->d_jac, B * nnz * sizeof(double)));
    CREATE_TRY(hipMalloc(&h->d_cost, B * POSE_NCT * sizeof(double)));
    CREATE_TRY(hipMalloc(&h->d_flags, B * sizeof(int32_t)));
    CREATE_TRY(hipHostMalloc(&h->h_x, B * POSE_NX * sizeof(double)));
    CREATE_TRY(hipHostMalloc(&h->h_f, B * sizeof(double)));
    CREATE_TRY(hipHostMalloc(&h->h_grad, B * POSE_NX * sizeof(double)));
    CREATE_TRY(hipHostMalloc(&h->h_g, B * m * sizeof(double)));
    CREATE_TRY(hipHostMalloc(&h->h_jac, B * nnz * sizeof(double)));
    CREATE_TRY(hipHostMalloc(&h->h_cost, B * POSE_NCT * sizeof(double)));
    CREATE_TRY(hipHostMalloc(&h->h_flags, B * sizeof(int32_t)));
    {
        void* dv = nullptr;
        CREATE_TRY(hipHostGetDevicePointer(&dv, h->h_x, 0)); h->hd_x = static_cast<double*>(dv);
        CREATE_TRY(hipHostGetDevicePointer(&dv, h->h_f, 0)); h->hd_f = static_cast<double*>(dv);
        CREATE_TRY(hipHostGetDevicePointer(&dv, h->h_grad, 0)); h->hd_grad = static_cast<double*>(dv);
        CREATE_TRY(hipHostGetDevicePointer(&dv, h->h_g, 0)); h->hd_g = static_cast<double*>(dv);
        CREATE_TRY(hipHostGetDevicePointer(&dv, h->h_jac, 0)); h->hd_jac = static_cast<double*>(dv);
        CREATE_TRY(hipHostGetDevicePointer(&dv, h->h_cost, 0)); h->hd_cost = static_cast<double*>(dv);
        CREATE_TRY(hipHostGetDevicePointer(&dv, h->h_flags, 0)); h->hd_flags = static_cast<int32_t*>(dv);
    }
    PoseTables* tb = new PoseTables();
    std::memset(tb, 0, sizeof(PoseTables));
    tb->head.kt = h->kt;
    tb->head.ks = PoseLayout::make_ksettings(st);
    tb->hands = PoseLayout::make_hands(st);
    for (int s = 0; s < gs::COUNT; ++s) tb->g_row[s] = h->L.g_row[size_t(s)];
    for (int r = 0; r < POSE_MAX_M; ++r) tb->row_slot[r] = -1;
    for (int s = 0; s < gs::COUNT; ++s) if (h->L.g_row[size_t(s)] >= 0) tb->row_slot[h->L.g_row[size_t(s)]] = s;   // (PoseLayout::build: every row in exactly one slot)
    for (int e = 0; e < h->L.nnz; ++e) {   // (the device table holds indices of the compacted staging)
        const int slot = h->L.jperm[size_t(e)];
        if (!pjs::kept(slot)) { delete tb; return fail(HIPNLP_E_INVALID, "internal: a Jacobian slot of the pose pattern outside the staged pieces (pose_body.h pjs)"); }
        tb->jperm[e] = pjs::index(slot);
    }
    tb->nnz = h->L.nnz; tb->m = h->L.m;
    for (int e = 0; e < h->L.hnnz; ++e) tb->hperm[e] = h->L.hperm[size_t(e)];
    tb->hnnz = h->L.hnnz;
    hipError_t ce = hipnlp_internal_memcpy(h->d_tb, tb, sizeof(PoseTables), hipMemcpyHostToDevice);
    delete tb;
    if (ce != hipSuccess) return fail(HIPNLP_E_NODEVICE, std::string("hipMemcpy tables: ") + hipGetErrorString(ce));
#undef CREATE_TRY
    *out = h;
    return HIPNLP_OK;
}

void hipnlp_pose_destroy(hipnlp_pose_handle* h) { pose_free_all(h); }

int hipnlp_pose_get_dims(const hipnlp_pose_handle* h, hipnlp_pose_dims* o) {
    if (!h || !o) return HIPNLP_E_INVALID;
    o->n = POSE_NX; o->m = h->L.m; o->nnz = h->L.nnz; o->np = POSE_NP;
    return HIPNLP_OK;
}

int hipnlp_pose_set_params(hipnlp_pose_handle* h, const double* p) {
    if (!h || !p) return HIPNLP_E_INVALID;
    HIP_TRY(h, hipSetDevice(h->dev));
    const size_t B = size_t(h->batch);
    h->p.assign(p, p + B * POSE_NP);
    std::vector<double> pk(B * PK_STRIDE), xr(B * XR_STRIDE);
    std::vector<GParams> gp(B);
    for (size_t b = 0; b < B; ++b) pack_pose_params(p + b * POSE_NP, pk.data() + b * PK_STRIDE, xr.data() + b * XR_STRIDE, gp[b]);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipnlp_internal_memcpy(h->d_pk, pk.data(), pk.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipnlp_internal_memcpy(h->d_xr, xr.data(), xr.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipnlp_internal_memcpy(h->d_gp, gp.data(), gp.size() * sizeof(GParams), hipMemcpyHostToDevice));
    h->params_set = true;
    h->have_result = false;
    return HIPNLP_OK;
}

int hipnlp_pose_bounds(const hipnlp_pose_handle* h, double* lbg, double* ubg) {
    if (!h || !lbg || !ubg) return HIPNLP_E_INVALID;
    if (!h->params_set) return HIPNLP_E_PARAMS;
    for (int b = 0; b < h->batch; ++b) h->L.bounds(h->p.data() + size_t(b) * POSE_NP, lbg + size_t(b) * size_t(h->L.m), ubg + size_t(b) * size_t(h->L.m));
    return HIPNLP_OK;
}

int hipnlp_pose_sparsity(const hipnlp_pose_handle* h, int32_t* irow, int32_t* jcol) {
    if (!h || !irow || !jcol) return HIPNLP_E_INVALID;
    std::memcpy(irow, h->L.irow.data(), size_t(h->L.nnz) * sizeof(int32_t));
    std::memcpy(jcol, h->L.jcol.data(), size_t(h->L.nnz) * sizeof(int32_t));
    return HIPNLP_OK;
}

int hipnlp_pose_eval_device(hipnlp_pose_handle* h, const double* x_dev, double* f_dev, double* grad_dev, double* g_dev, double* jac_dev, void* stream) {
    if (!h || !x_dev) return HIPNLP_E_INVALID;
    if (!h->params_set) { h->err = "parameters not set (hipnlp_pose_set_params)"; return HIPNLP_E_PARAMS; }
    HIP_TRY(h, hipSetDevice(h->dev));
    h->have_result = false;
    return pose_launch(h, x_dev, f_dev ? f_dev : h->d_f, grad_dev, g_dev, jac_dev, stream ? hipStream_t(stream) : h->stream, false);
}

int hipnlp_pose_eval(hipnlp_pose_handle* h, const double* x, double* f, double* grad_f, double* g, double* jac) {
    if (!h || !x) return HIPNLP_E_INVALID;
    if (!h->params_set) { h->err = "parameters not set (hipnlp_pose_set_params)"; return HIPNLP_E_PARAMS; }
    const size_t B = size_t(h->batch), m = size_t(h->L.m), nnz = size_t(h->L.nnz);
    HIP_TRY(h, hipSetDevice(h->dev));
    // x is read by the kernel from the pinned staging copy, every output is stored by the kernel into its pinned block (PCIe stores):
    // one launch and one synchronisation, no copy command (56 -> 31 us per call for one pose through the ctypes binding, the Hessian 63 -> 42 us; tools/diag/pose_host_time.py)
    std::memcpy(h->h_x, x, B * POSE_NX * sizeof(double));
    const int rc = pose_launch(h, h->hd_x, h->hd_f, h->hd_grad, h->hd_g, h->hd_jac, h->stream, h->time_host, true);
    if (rc != HIPNLP_OK) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->have_result = true;
    if (f) std::memcpy(f, h->h_f, B * sizeof(double));
    if (grad_f) std::memcpy(grad_f, h->h_grad, B * POSE_NX * sizeof(double));
    if (g) std::memcpy(g, h->h_g, B * m * sizeof(double));
    if (jac) std::memcpy(jac, h->h_jac, B * nnz * sizeof(double));
    for (size_t b = 0; b < B; ++b)
        if (h->h_flags[b]) { h->err = "non-finite value produced by the evaluation"; return HIPNLP_E_NUMERIC; }
    return HIPNLP_OK;
}

int hipnlp_pose_hess_nnz(const hipnlp_pose_handle* h, int32_t* nnz_h) {
    if (!h || !nnz_h) return HIPNLP_E_INVALID;
    *nnz_h = h->L.hnnz;
    return HIPNLP_OK;
}

int hipnlp_pose_hess_sparsity(const hipnlp_pose_handle* h, int32_t* irow, int32_t* jcol) {
    if (!h || !irow || !jcol) return HIPNLP_E_INVALID;
    std::memcpy(irow, h->L.hrow.data(), size_t(h->L.hnnz) * sizeof(int32_t));
    std::memcpy(jcol, h->L.hcol.data(), size_t(h->L.hnnz) * sizeof(int32_t));
    return HIPNLP_OK;
}

static int pose_hess_launch(hipnlp_pose_handle* h, const double* x_dev, const double* sigma_dev, const double* lambda_dev, double* hess_dev,
                            hipStream_t s, bool timed, bool host_block = false) {
    PArgs a{};
    a.tb = h->d_tb; a.x = x_dev; a.pk = h->d_pk; a.xr = h->d_xr; a.gp = h->d_gp;
    a.cost_terms = h->d_cost; a.flags = host_block ? h->hd_flags : h->d_flags;
    a.sigma = sigma_dev; a.lambda = lambda_dev; a.hess = hess_dev; a.m = h->L.m;
#ifdef HIPNLP_STAMPS
    if (!h->d_stamps) { HIP_TRY(h, hipMalloc(&h->d_stamps, size_t(h->batch) * 256 * sizeof(unsigned long long))); HIP_TRY(h, hipMemset(h->d_stamps, 0, size_t(h->batch) * 256 * sizeof(unsigned long long))); }
    a.stamps = h->d_stamps;
#endif
    if (timed) HIP_TRY(h, hipEventRecord(h->ev0, s));
    if (h->d.settings.terrain == HIPNLP_TERRAIN_PLANAR)
        hipLaunchKernelGGL(hipnlp_pose_hess_kernel<HIPNLP_TERRAIN_PLANAR>, dim3(unsigned(h->batch)), dim3(WG), 0, s, a);
    else
        hipLaunchKernelGGL(hipnlp_pose_hess_kernel<HIPNLP_TERRAIN_SMOOTH_STEPS>, dim3(unsigned(h->batch)), dim3(WG), 0, s, a);
    if (timed) HIP_TRY(h, hipEventRecord(h->ev1, s));
    HIP_TRY(h, hipGetLastError());
    if (timed) h->timing_valid = true;
    return HIPNLP_OK;
}

int hipnlp_pose_eval_hess_device(hipnlp_pose_handle* h, const double* x_dev, const double* obj_factor_dev, const double* lambda_dev,
                                 double* hess_dev, void* stream) {
    if (!h || !x_dev || !obj_factor_dev || !lambda_dev || !hess_dev) return HIPNLP_E_INVALID;
    if (!h->params_set) { h->err = "parameters not set (hipnlp_pose_set_params)"; return HIPNLP_E_PARAMS; }
    HIP_TRY(h, hipSetDevice(h->dev));
    return pose_hess_launch(h, x_dev, obj_factor_dev, lambda_dev, hess_dev, stream ? hipStream_t(stream) : h->stream, false);
}

int hipnlp_pose_eval_hess(hipnlp_pose_handle* h, const double* x, const double* obj_factor, const double* lambda, double* hess) {
    if (!h || !x || !obj_factor || !lambda || !hess) return HIPNLP_E_INVALID;
    if (!h->params_set) { h->err = "parameters not set (hipnlp_pose_set_params)"; return HIPNLP_E_PARAMS; }
    const size_t B = size_t(h->batch), m = size_t(h->L.m), hn = size_t(h->L.hnnz);
    HIP_TRY(h, hipSetDevice(h->dev));
    // every piece allocated at most once: a failure half way leaves the pieces that exist for the next attempt (nothing leaks)
    // pinned blocks, each with its device-visible address: the kernel gathers the multipliers out of host memory (m words per pose:
    // the rows of ONE pose) and stores the values into host memory; no copy command
    auto pinned = [&](double*& hostp, double*& devp, size_t count) -> int {
        if (!hostp) {
            HIP_TRY(h, hipHostMalloc(&hostp, count * sizeof(double)));
            void* dv = nullptr;
            HIP_TRY(h, hipHostGetDevicePointer(&dv, hostp, 0));
            devp = static_cast<double*>(dv);
        }
        return HIPNLP_OK;
    };
    int prc = pinned(h->h_sigma, h->hd_sigma, B);
    if (prc == HIPNLP_OK) prc = pinned(h->h_lambda, h->hd_lambda, B * m);
    if (prc == HIPNLP_OK) prc = pinned(h->h_hess, h->hd_hess, B * hn);
    if (prc != HIPNLP_OK) return prc;
    std::memcpy(h->h_x, x, B * POSE_NX * sizeof(double));
    std::memcpy(h->h_sigma, obj_factor, B * sizeof(double));
    std::memcpy(h->h_lambda, lambda, B * m * sizeof(double));
    const int rc = pose_hess_launch(h, h->hd_x, h->hd_sigma, h->hd_lambda, h->hd_hess, h->stream, h->time_host, true);
    if (rc != HIPNLP_OK) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    std::memcpy(hess, h->h_hess, B * hn * sizeof(double));
    for (size_t b = 0; b < B; ++b)
        if (h->h_flags[b]) { h->err = "non-finite value produced by the evaluation"; return HIPNLP_E_NUMERIC; }
    return HIPNLP_OK;
}

int hipnlp_pose_cost_terms(hipnlp_pose_handle* h, double* values) {
    if (!h || !values) return HIPNLP_E_INVALID;
    if (!h->have_result) {
        HIP_TRY(h, hipSetDevice(h->dev));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        HIP_TRY(h, hipnlp_internal_memcpy(h->h_cost, h->d_cost, size_t(h->batch) * POSE_NCT * sizeof(double), hipMemcpyDeviceToHost));
    }
    std::memcpy(values, h->h_cost, size_t(h->batch) * POSE_NCT * sizeof(double));
    return HIPNLP_OK;
}

const char* hipnlp_pose_cost_term_name(int i) {
    static const char* names[POSE_NCT] = {"base_quaternion_error", "frame_rotation_error", "com_position_error", "joint_positions_error",
                                          "average_force_regularization", "point_position_regularization", "force_regularization",
                                          "left_hand_position_error", "right_hand_position_error"};
    return (i >= 0 && i < POSE_NCT) ? names[i] : "";
}

int hipnlp_pose_num_row_blocks(const hipnlp_pose_handle* h) { return h ? int(h->L.blocks.size()) : HIPNLP_E_INVALID; }

int hipnlp_pose_row_block(const hipnlp_pose_handle* h, int i, const char** name, int32_t* first_row, int32_t* rows) {
    if (!h || i < 0 || i >= int(h->L.blocks.size())) return HIPNLP_E_INVALID;
    const PoseRowBlock& b = h->L.blocks[size_t(i)];
    if (name) *name = b.name.c_str();
    if (first_row) *first_row = b.first_row;
    if (rows) *rows = b.rows;
    return HIPNLP_OK;
}

int hipnlp_pose_set_host_timing(hipnlp_pose_handle* h, int on) {
    if (!h) return HIPNLP_E_INVALID;
    h->time_host = on != 0;
    if (!h->time_host) h->timing_valid = false;
    return HIPNLP_OK;
}

int hipnlp_pose_last_kernel_ms(hipnlp_pose_handle* h, float* ms) {
    if (!h || !ms) return HIPNLP_E_INVALID;
    if (!h->timing_valid) { h->err = "no timed evaluation yet (hipnlp_pose_set_host_timing(h, 1), then hipnlp_pose_eval / hipnlp_pose_eval_hess)"; return HIPNLP_E_INVALID; }
    HIP_TRY(h, hipSetDevice(h->dev));
    HIP_TRY(h, hipEventSynchronize(h->ev1));
    HIP_TRY(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
    return HIPNLP_OK;
}

#ifdef HIPNLP_STAMPS
int hipnlp_pose_debug_stamps(hipnlp_pose_handle* h, unsigned long long* out /*[batch][4][64]*/) {
    if (!h || !out || !h->d_stamps) return HIPNLP_E_INVALID;
    HIP_TRY(h, hipSetDevice(h->dev));
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipnlp_internal_memcpy(out, h->d_stamps, size_t(h->batch) * 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return HIPNLP_OK;
}
#endif

}  // extern "C"
