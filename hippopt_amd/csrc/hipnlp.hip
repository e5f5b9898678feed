// hipnlp.hip — gfx950 kernels and the C-ABI (include/hipnlp.h) of the multiple-shooting NLP-callback engine.
//
// Execution model: one workgroup of four (or eight) role-specialised wavefronts per shooting knot.
// The workgroup stages the knot record x_k (189 fp64, contiguous => coalesced loads), its predecessor x_{k-1} (trapezoid
// halo), the per-knot parameter record and the read-only tables in LDS, runs the knot program of knot_body.h (task groups on
// waves, workgroup barriers between dependent phases; lanes mapped to contact points / joints / (joint, row) / (link,
// component) / Jacobian entries), collects all outputs of the knot in LDS at compile-time native slots and finally streams
// them out: the knot's CCS column block of jac g as ONE contiguous run (permuted through a prefetched int table), grad f
// contiguous, g scattered into the reference's constraint-type-major order, the cost partials per knot.  The total cost is summed
// in a fixed order (bitwise reproducible f) inside the same launch by one more workgroup per trajectory that polls the knots' tagged
// partials (one kernel launch per callback set), or by a second tiny kernel for trajectories of more than 256 knots / long launches.
// The build splits the compiler's two-address LDS reads in the assembly of these kernels (tools/asm_patch.py, __graft_entry__.build).
// hipnlp_knot_hess_kernel evaluates the exact Hessian of the Lagrangian (knot_hess_body.h) behind the same knot program.
#include <hip/hip_runtime.h>

#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <new>
#include <string>
#include <cctype>
#include <sched.h>
#include <tuple>
#include <type_traits>
#include <vector>

#include "knot_hess_layout.h"
#include "layout.h"

using namespace hipnlp;
typedef unsigned long long u64;

namespace {

// four (or eight) wavefronts per knot: role-specialised waves (knot_body.h, HIPNLP_KNOT_PROGRAM)
static_assert(gs::COUNT == HIPNLP_G_STAGE, "HIPNLP_G_STAGE must equal the native g slot count");

// (copy-out tables padded to whole workgroups of 512 lanes — -1 / 0 behind the entries — so that their prefetch is unconditional:
//  a load inside a branch makes the compiler wait for ALL younger loads at the first s_waitcnt vmcnt)
constexpr int GS_PAD = (gs::COUNT + 511) / 512 * 512, JS_PAD = (js::COUNT + 511) / 512 * 512;
// Entries of a knot's block that depend on x — the only ones a VARY instantiation stores (at most: 791 on the planar terrain, 1041 on
// the smooth steps; tests/test_constant_jacobian.py) — rounded up to whole workgroups of 512: the trip count of its copy-out.
constexpr int vary_cap(int terrain, int wg) { return terrain == HIPNLP_TERRAIN_PLANAR ? 1024 : (wg == 256 ? 1280 : 1536); }
constexpr int VCAP_MAX = 1536;
// Entries of the thread-major copy-out tables (jperm_t, jpermv_t, the g_b half of gab_t) carry COPY_EARLY when their native slot gets its
// value in the first two phases of the knot program (Layout::jslot_phase / gslot_phase: recorded, not declared): the eight-wave kernel of
// a launch into HOST memory stores those behind the second barrier — 42 % of a knot's bytes start over the link while the kinematic
// phases still run — and leaves them out of the copy-out at the end.
constexpr int32_t COPY_EARLY = 1 << 30, COPY_SLOT = COPY_EARLY - 1;
// The g_b half of a gab_t word: bits 0..15 the row stride of the slot's constraint block (rows per knot), bits 16..25 the slot's POSITION
// among the valid slots of its knot variant (a compact g staging — hipnlp_eval_device_shard_vary — lists the rows a knot owns behind one
// another instead of at their 550 native slots), bit 30 COPY_EARLY.
constexpr int32_t GB_STRIDE = 0xffff, GB_POS_SHIFT = 16, GB_POS = 0x3ff;
static_assert(gs::COUNT <= GB_POS + 1, "position field of a g copy-out word");
__host__ __device__ constexpr int gb_stride(int32_t w) { return w & GB_STRIDE; }
__host__ __device__ constexpr int gb_pos(int32_t w) { return (w >> GB_POS_SHIFT) & GB_POS; }
// A word of the Jacobian copy-out tables: native slot in bits 0..11; jpermv_t (VARY instantiations) also carries the entry's POSITION in
// its knot block in bits 12..23 — the index of the varying run for a varying-first handle, the CCS position for a handle in CasADi's order,
// whose varying entries are scattered over the block (device destinations only: scattered 8-byte stores belong in HBM, not on PCIe).
constexpr int32_t JP_SLOT = 0xfff, JP_POS_SHIFT = 12, JP_POS = 0xfff;
static_assert(js::COUNT <= JP_SLOT + 1, "slot field of a copy-out word");
__host__ __device__ constexpr int jp_slot(int32_t w) { return w & JP_SLOT; }
__host__ __device__ constexpr int jp_pos(int32_t w) { return (w >> JP_POS_SHIFT) & JP_POS; }
struct DeviceTables {
    HeadTables head;
    int32_t g_a[3][GS_PAD];
    int32_t g_b[GS_PAD];
    int32_t jperm[3][JS_PAD];
    // the same two tables THREAD-MAJOR for the callback kernel of this handle (workgroup of 256 or 512 threads): the entries
    // (tid, tid + WG, tid + 2 WG, ...) one thread uses are contiguous, so its prefetch is one or two wide loads instead of one dword
    // load per entry (eleven loads and their address arithmetic in the staging of every wave)
    int32_t jperm_t[3][JS_PAD];
    int32_t gab_t[3][2 * GS_PAD];   // pairs {g_a, g_b}
    int32_t jperm_glob[16];
    int32_t nnz_v[3];
    int32_t n_glob, jac_glob_base;
    // VARY instantiations (varying-first order of a block, constants left alone): the varying run only, thread-major as jperm_t,
    // VCAP / WG entries per thread (padded with -1)
    int32_t jpermv_t[3][VCAP_MAX];
    int32_t nvary_v[3], pad_v;
    // SPLIT instantiations (two workgroups per knot: knot_body.h "Workgroup specialisation"): the same thread-major tables with the entries
    // the OTHER half of the program owns blanked (-1 / G_NONE) — [0] the kinematic workgroup's, [1] the model-free one's
    int32_t split_jperm_t[2][3][JS_PAD];
    int32_t split_gab_t[2][3][2 * GS_PAD];
    int32_t split_jpermv_t[2][3][VCAP_MAX];
    int32_t split_jperm_glob[2][16];
};

// (see constants_check_and_repair)
struct ConstCheck {
    double* jac;               // [batch][jac_stride] destination (device memory), or null: nothing to check
    const double* ctpl;        // [templates][ctpl_len]
    const int32_t* ctpl_of_b;
    int32_t* healed;
    int64_t jac_stride, jac_off;
    int32_t ctpl_len, ctpl_off[4], nnz_v[3], first_const[3] /* position of a block's first constant entry, -1: none */, kb, nk, N, n_glob, jac_glob_base;
    int32_t samp[3][4];        // four constant positions of a variant-v block: {first, last, two in between} (-1: none)
    int32_t parity;            // launch number & 1: even launches look at samp[.][0..1], odd ones at samp[.][2..3] (two loads per knot and launch)
};
// the templates hold this word (a quiet NaN no constant has) at the positions of a block that depend on x
constexpr unsigned long long CTPL_VARYING = 0x7FF8C0DEC0DE0001ull;

struct KArgs {
    const DeviceTables* tb;
    const double* x;        // [batch][n]
    const double* pk;       // [batch][N][PK_STRIDE]
    const GParams* gp;      // [batch]
    double* g;              // [batch][m]      or null
    double* jac;            // [batch][nnz]    or null
    double* grad;           // [batch][n]      or null
    double* g_stage;        // [batch][nk][gs::COUNT] knot-major staging of g (sharded mode) or null
    double* cost_knot;      // [batch][nk][NCT] per-knot cost partials (four-wave variant: summed by hipnlp_reduce_kernel)
    double* f;              // [batch]        total cost (eight-wave variant: written by the trajectory's reducer workgroup)
    double* cost_terms;     // [batch][NCT]   per-term totals, same writer
    unsigned long long* cost_pub;   // [batch][nk][NCT][2] eight-wave variant: {partial, partial ^ pub_pattern(seq)}, see the reducer
    int32_t* flag;          // [batch]        generation flag of the non-finite detector: == seq after a launch that produced one
    int32_t* flag_host;     // [batch] or null: the same flag in pinned host memory (host-buffer path: a plain store of seq — every
                            //         writer of a launch stores the same value, launches of a handle are stream ordered — so the
                            //         host needs no copy of the device flag behind the launch)
    int32_t seq, early;     // launch number of this handle (1, 2, ...); early: outputs go to host memory — what is final after the second phase leaves then
    int32_t N, n, m, nnz, knot_begin, nk;
    int64_t jac_stride, jac_off, grad_stride, grad_off;  // output addressing: full arrays (stride nnz / n, offset 0) or shard-local
    // VARY instantiations: entries per knot block of the DESTINATION of jac g — the block lengths of the pattern (nnz_v: the destination
    // holds the constants between the varying runs) or, for a COMPACT destination (hipnlp_eval_device_vary / _shard_vary: what an exchange
    // between GPUs moves), the lengths of the varying runs alone (nvary_v: run behind run, no constants anywhere)
    int32_t jb_first, jb_interior;
    // ... and, for a compact g staging (gsb_first > 0), the rows a first / an interior knot owns, the rows of one trajectory's staging and
    // the rows owned by the knots in front of the handle's first one
    int32_t gsb_first, gsb_interior;
    int64_t gs_stride, gs_off;
    // peer mode (hipnlp_eval_device_peers; batch 1): instead of g / jac / grad / f above, the shard's values go — at their FINAL positions —
    // into the buffer [grad (n) | jac (nnz) | g (m) | f partials (npeer) | f] of every one of npeer ranks (this device's own among them)
    double* const* peer_out;   // [npeer] device-visible base addresses, or null
    int32_t npeer, peer_rank;
    // VARY instantiations, jac in DEVICE memory: the constant entries are expected to be in place (filled once per parameter set by
    // hipnlp_fill_const_kernel); the launch's reducer workgroup checks them (constants_check_and_repair); null pointers: no check
    ConstCheck cc;
#ifdef HIPNLP_STAMPS
    unsigned long long* stamps;  // diagnostic build only: [blocks][32] s_memtime at every barrier (never in the product library)
#endif
};

// The four-wave (throughput) callback kernels run on the COMPACT scratch (knot_body.h, KnotScratchT<LAYOUT_COMPACT>: <= 40 KB of LDS per
// workgroup with the lite tables below: four workgroups per CU).  The eight-wave latency variant keeps the full layout: it has one
// workgroup per CU whatever its LDS, and its duration is that of the SLOWEST workgroup — the last knot, whose end rows would wait
// for global memory in the compact layout (measured: 10.0 -> 10.9 us per 100-knot launch).
template <int TERRAIN, int WAVES = 4, bool VARY = false> struct DevEm {
    static constexpr int kTerrain = TERRAIN;
    static constexpr int kWaves = WAVES;
    static constexpr int kLayout = WAVES == 4 ? LAYOUT_COMPACT : LAYOUT_FULL;
    static constexpr bool kCompact = kLayout != LAYOUT_FULL;
    // VARY on the compact scratch: the staging of jac g holds the slots that may depend on x only (nlp_defs.h: [js::V0, js::V0 +
    // js::vary_slots(terrain))) — 31.5 KB of LDS per workgroup on the planar terrain instead of 40.8: room for a FIFTH workgroup per CU
    static constexpr bool kTrim = VARY && kCompact;
    static constexpr int kJacOff = kTrim ? js::V0 : 0;
    using Scratch = KnotScratchT<kLayout, kTrim ? js::vary_slots(TERRAIN == HIPNLP_TERRAIN_PLANAR) : js::COUNT>;
    double* g;
    double* jac;   // (base of the staging moved back by kJacOff: indexed by native slot)
    __device__ __forceinline__ void G(int slot, int, double v) { g[slot] = v; }   // (horizon-end rows go through emit_g_end)
    __device__ __forceinline__ void J(int slot, int, int, double v) { jac[slot] = v; }
    // entries that do not depend on x (emit_jc): a VARY instantiation neither stages nor stores them — its destination holds them
    __device__ __forceinline__ void JC(int slot, int, int, double v) { if constexpr (!VARY) jac[slot] = v; }
};

#ifdef HIPNLP_DIAG_TASKSET
constexpr bool diag_same(const char* a, const char* b) { while (*a && *a == *b) { ++a; ++b; } return *a == *b; }
constexpr bool diag_task_is_kinematic(const char* n) {
    const char* kin[] = {"t_joints", "t_base", "t_fk_rot_a", "t_fk_rot_b", "t_link_u_a", "t_link_u_b", "t_links", "t_frames", "t_link_inertia", "t_composite_g0",
                         "t_composite_g1", "t_composite_g2", "t_composite_g3", "t_composite_g4", "t_composite_g5", "t_pkin", "t_columns", "t_cmm_columns",
                         "t_frame_columns", "t_kinc", "t_comc", "t_cmmc", "t_kinc_s", "t_feetd"};
    for (const char* k : kin) if (diag_same(n, k)) return true;
    return false;
}
constexpr bool diag_task_kept(const char* n) { return diag_task_is_kinematic(n) == (HIPNLP_DIAG_TASKSET == 1); }
#endif

// LDS image of the read-only tables every phase indexes per lane (global memory would cost one L2 round trip per phase)
template <bool COMPACT> struct SharedTablesT;
template <> struct alignas(16) SharedTablesT<false> {
    HeadTables head;
    GParams gp;
    __device__ __forceinline__ const KinTables& kin() const { return head.kt; }
    __device__ __forceinline__ const KSettings& settings() const { return head.ks; }
};
template <> struct alignas(16) SharedTablesT<true> {   // without the blocks read once per knot / by the end knots only (6.9 KB)
    KSettings ks;
    KinLite kt;
    GParamsLite gp;
    __device__ __forceinline__ const KinLite& kin() const { return kt; }
    __device__ __forceinline__ const KSettings& settings() const { return ks; }
};
using SharedTables = SharedTablesT<false>;
static_assert(sizeof(KSettings) % 8 == 0 && sizeof(KinLite) % 8 == 0 && sizeof(GParamsLite) % 8 == 0, "copied in 8-byte words");

// Workgroup barrier that orders LDS traffic only.  The phases of the knot program exchange data through LDS and through nothing
// else, so global loads may stay in flight across it: __syncthreads() carries a workgroup fence whose s_waitcnt vmcnt(0) makes every
// barrier wait for the copy-out tables prefetched at the top of the kernel.  (The "memory" clobber keeps the compiler from moving
// memory operations across it; the hardware wait covers the LDS operations of this wave.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Tag of a published cost partial (eight-wave variant): the pair {v, v ^ pub_pattern(seq)} is valid for launch `seq` and for no other,
// whichever of its two words a reader happens to see first (see the reducer in hipnlp_knot_kernel).
__device__ __forceinline__ unsigned long long pub_pattern(int32_t seq) { return ((unsigned long long)(uint32_t)seq * 0x9E3779B97F4A7C15ull) | 1ull; }
// knots per trajectory the in-launch reducer workgroup can stage in the scratch of its kernel ([nk][16] doubles): 256, less on the
// trimmed scratch of the four-wave VARY kernels (hipnlp_create keeps `fused` within the smallest cap of the kernels a handle may launch)
template <class Scratch> constexpr int reducer_cap() { return sizeof(Scratch) / (16 * sizeof(double)) >= 256 ? 256 : int(sizeof(Scratch) / (16 * sizeof(double))) / 16 * 16; }
constexpr int PUB_SPIN_CAP = 1 << 20;   // polls of the reducer before it gives up (each at least one memory round trip: > 1 s)

// VARY launches into a DEVICE destination: are the constant entries of trajectory b's knot blocks in place?  Run by the ONE workgroup
// per trajectory that sums its cost (the reducer workgroup of the launch, or hipnlp_reduce_kernel behind it) — never by the knot
// workgroups: two loads on their path, wherever they were issued, cost 4 - 10 % at batch (vector loads of table words with a wait
// right behind them; then the round trip to HBM itself).  Thread t looks at constant entries of the block of knot kb + t (+ 256, ...):
// FOUR positions — the first, the last and two in between (ConstCheck::samp) — two per launch, alternating with the launch number (every
// sampled load is a sector of HBM traffic of its own: four per knot and launch doubled the reduction kernel behind a x 1024 launch,
// 10.6 -> 21.9 us); if any differs from the handle's template the workgroup puts the
// constants of ALL the trajectory's blocks back (rare — a caller that wrote over its buffer — slow, right: the knot workgroups store
// other entries, nothing is written twice).  A SAMPLE: constants overwritten elsewhere in a block are not seen — a caller that hands
// over other memory at an address the handle has filled (a tensor freed and re-allocated) says so with hipnlp_forget_jac_destination.
__device__ __forceinline__ void constants_check_and_repair(const ConstCheck& c, int b, int nthreads, int* lds_flags /* one int per wave of the workgroup */) {
    if (!c.jac || !c.ctpl) return;
    const int tid = threadIdx.x;
    const double* tpl = c.ctpl + int64_t(c.ctpl_of_b[b]) * c.ctpl_len;
    double* out = c.jac + int64_t(b) * c.jac_stride - c.jac_off;
    int miss = 0;
    for (int t = tid; t < c.nk; t += nthreads) {
        const int k = c.kb + t, v = k == 0 ? VAR_FIRST : (k == c.N - 1 ? VAR_LAST : VAR_INTERIOR);
        if (c.first_const[v] < 0) continue;
        const int64_t base = k == 0 ? 0 : int64_t(c.nnz_v[VAR_FIRST]) + int64_t(k - 1) * c.nnz_v[VAR_INTERIOR];
#pragma unroll
        for (int q0 = 0; q0 < 2; ++q0) {
            const int q = 2 * c.parity + q0;
            const int sp = c.samp[v][q] >= 0 ? c.samp[v][q] : c.first_const[v];
            const double seen = out[base + sp], want = tpl[c.ctpl_off[v] + sp];
            miss |= __double_as_longlong(seen) != __double_as_longlong(want);
        }
    }
    // (an OR over the workgroup through the caller's LDS words: __syncthreads_or brings 256 B of LDS of its own, which the four-wave
    //  kernel — 40 KB to the byte for four workgroups per CU — does not have)
    if ((tid & 63) == 0) lds_flags[tid >> 6] = 0;
    __syncthreads();
    if (__any(miss) && (tid & 63) == 0) lds_flags[tid >> 6] = 1;
    __syncthreads();
    int any = 0;
    for (int w = 0; w < nthreads / 64; ++w) any |= lds_flags[w];
    if (!any) return;
    for (int t = 0; t < c.nk; ++t) {
        const int k = c.kb + t, v = k == 0 ? VAR_FIRST : (k == c.N - 1 ? VAR_LAST : VAR_INTERIOR);
        const int64_t base = k == 0 ? 0 : int64_t(c.nnz_v[VAR_FIRST]) + int64_t(k - 1) * c.nnz_v[VAR_INTERIOR];
        if (c.first_const[v] < 0) continue;
        for (int i = c.first_const[v] + tid; i < c.nnz_v[v]; i += nthreads) {
            const double tv = tpl[c.ctpl_off[v] + i];
            if ((unsigned long long)__double_as_longlong(tv) != CTPL_VARYING) out[base + i] = tv;
        }
    }
    if (c.kb + c.nk == c.N) for (int i = tid; i < c.n_glob; i += nthreads) out[int64_t(c.jac_glob_base) + i] = tpl[c.ctpl_off[3] + i];
    if (tid == 0) atomicAdd(c.healed, 1);
}

// WAVES = 4: 256 threads; 4 waves per SIMD = 4 workgroups per CU (<= 128 VGPRs, <= 40 KB of LDS on the compact scratch).  The
//            throughput variant.
// WAVES = 8: 512 threads, the roles of the knot program spread over twice the waves (two per SIMD).  The latency variant, used
//            when the whole launch is resident at once at two workgroups per CU ((knots + 1) x batch <= 512), e.g. one 100-knot
//            trajectory.
// Four workgroups per CU for the four-wave kernels (compact scratch + lite tables: <= 40 KB of LDS, <= 128 VGPRs).
// PEERS: the instantiation behind hipnlp_eval_device_peers (outputs into every rank's buffer); a template parameter rather than a
// run-time branch — as a branch the unused path cost the plain callback 1.2 % at N = 100 (7.99 against 7.89 us) and 0.8 % at x 64.
// VARY: the instantiation for destinations that already hold the constant entries of jac g (varying-first order of a block): the
// tasks do not stage them (DevEm::JC), the copy-out walks the varying run of the block only — half the trips on the planar terrain.
// FIVE workgroups per CU: the four-wave VARY kernels, whose trimmed scratch (KnotScratchT::trimmed) is 31.0 KB of LDS per workgroup on
// the planar terrain and exactly 32 KB on the smooth steps — with a register budget of 96 (five waves per SIMD), which it meets by fetching its copy-out tables behind the last
// barrier but one instead of holding them from the first instruction (LATE_TABLES below).
#ifndef HIPNLP_FIVE_PER_CU
#define HIPNLP_FIVE_PER_CU 1
#endif
// (the peer-store VARY kernels — one store loop per rank behind the program — keep the four-per-CU register budget)
template <int TERRAIN, int WAVES, bool VARY, bool PEERS = false> constexpr bool five_per_cu = HIPNLP_FIVE_PER_CU && VARY && WAVES == 4 && !PEERS;
// STORES_EARLY: the instantiation behind launches that arm the early copy-out (a.early: eight-wave VARY launches into host memory).  The
// hook itself is compiled into every eight-wave kernel (see EARLY_OUT below); what this parameter moves is the fetch of the copy-out tables —
// in front of the staging wait, where no wait for them can end up behind an early store, at 0.1 - 0.15 us of prologue that the launches
// which never store early (the device-resident ones) do not pay.
// SPLIT: two workgroups per knot in one launch (eight-wave kernels of launches that leave CUs idle: device destinations, one workgroup per
// CU) — workgroups [0, nk) of a grid row run the kinematic half of the knot program, [nk, 2 nk) the model-free half (knot_body.h,
// split_task_is_model_free), each stores the outputs its half owns; the row's reducer is workgroup 2 nk.  Measured on the 100-knot launch
// with timing-only builds (tools/diag/taskset_experiment.sh): either half alone ends 0.6 - 1.4 us before the whole program does.
template <int TERRAIN, int WAVES, bool PEERS = false, bool VARY = false, bool STORES_EARLY = false, bool SPLIT = false> __global__ __launch_bounds__(64 * WAVES)
__attribute__((amdgpu_waves_per_eu(WAVES == 4 ? (five_per_cu<TERRAIN, WAVES, VARY, PEERS> ? 5 : 4) : 2, five_per_cu<TERRAIN, WAVES, VARY, PEERS> ? 5 : 4)))
void hipnlp_knot_kernel(const DeviceTables* tb_p, const double* x_p, const double* pk_p, const GParams* gp_p, int N_p, int n_p, int kb_p, int nk_p, KArgs a) {
    static_assert(!SPLIT || (WAVES == 8 && !PEERS && !STORES_EARLY), "two workgroups per knot: the eight-wave kernels of device-resident launches");
    // the leading scalar arguments repeat what the staging loads need: the build preloads them into SGPRs
    // (-mllvm -amdgpu-kernarg-preload-count), so the first global loads do not wait for a kernarg fetch
    constexpr int WG = 64 * WAVES;
    using Em = DevEm<TERRAIN, WAVES, VARY>;
    constexpr bool COMPACT = Em::kCompact;
    using Scratch = typename Em::Scratch;
    __shared__ Scratch s;
    __shared__ SharedTablesT<COMPACT> tabs;
    static_assert(!COMPACT || sizeof(Scratch) + sizeof(SharedTablesT<COMPACT>) <= 40 * 1024, "four workgroups per CU: 40 KB of LDS each");
#ifdef HIPNLP_STAMPS
    const unsigned long long st_entry = __builtin_amdgcn_s_memtime(), st_real0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long st_issued = 0, st_loaded = 0;
#endif
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // workgroup -> knot, XCD-aware: workgroups are dealt round-robin over the 8 XCDs (observed, relied on for speed only), each with
    // its own L2, so within one trajectory (one grid row) the workgroups x, x + 8, x + 16, ... share an XCD: they get CONSECUTIVE
    // knots.  The neighbour record x_{k-1} every knot reads is then found in the L2 its owner just filled, and the short runs that
    // neighbouring knots write into one 128-byte line of a constraint block of g meet in one L2 before they leave for HBM
    // (measured, 100 knots x 64, HBM bytes per launch: 26.6 -> 16.5 MB read, 109.6 -> 96.1 MB written; algorithmic: 13.7 + 94.7 MB).
    // A bijection for any row length: class j = x mod 8 owns q + (j < r) knots, q = nk / 8, r = nk mod 8.
    const int b = blockIdx.y;
    // (launches of at most 256 knots per trajectory: one more workgroup per grid row, the trajectory's cost reducer)
    const unsigned nkx = unsigned(nk_p);   // (= gridDim.x, less the reducer; preloaded: the grid size is a kernarg load away)
    {
        if (blockIdx.x == (SPLIT ? 2u * nkx : nkx)) {
            // ---- total cost without a second kernel and without a tail ---------------------------------------------------------
            // Every knot workgroup publishes its cost partials as soon as they are final (end of phase C) and goes on; THIS
            // workgroup does nothing but poll them (agent-scope loads: the writers sit on other XCDs) until all carry this launch's
            // tag, then sums them in the fixed order of hipnlp_reduce_kernel (bitwise the same f from both variants) — by the
            // time the knot workgroups have streamed their outputs.  A published partial is the pair {v, v ^ pub_pattern(seq)}:
            // two plain 8-byte agent-scope stores, no ordering between them needed — a reader that pairs a value with the tag
            // of another launch sees v ^ tag != pattern unless both launches published the same v, in which case v is right.
            // (The previous protocol — store, acknowledgement, ticket, read-back by the last arriver: three dependent cross-XCD
            //  round trips behind the LAST workgroup's partials — left a tail of 4.4 k cycles on that workgroup: 22.6 k cycles
            //  against 18.4 k for every other one, i.e. 2 of the launch's 10 us.)
            // Forward progress: the reducer of row b waits only for workgroups dispatched before it; it is bounded anyway.
            if constexpr (VARY) constants_check_and_repair(a.cc, b, WG, reinterpret_cast<int*>(&s));   // (while the knot workgroups of the row run their programs)
            __syncthreads();
            double* red = reinterpret_cast<double*>(&s);   // [nk][16]
            static_assert(sizeof(Scratch) >= size_t(reducer_cap<Scratch>()) * 16 * sizeof(double) && reducer_cap<Scratch>() >= 200, "reducer staging");
            const unsigned long long pat = pub_pattern(a.seq);
            const int t = lane & 15, q = lane >> 4;
            // (SPLIT: every cost term is published by the workgroup whose half of the program owns it — the model-free workgroups' partials
            //  live behind those of the kinematic ones: [2][batch][nk][NCT][2])
            const unsigned long long* base = a.cost_pub + size_t(b) * a.nk * NCT * 2 +
                                             ((SPLIT && t < NCT && split_cost_is_model_free(t)) ? size_t(gridDim.y) * a.nk * NCT * 2 : size_t(0));
            constexpr int KR = 4 * WAVES, U = 256 / KR;   // wave w, lane (t, q): knots 4 w + q + KR u
            unsigned pending = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) if (t < NCT && 4 * wave + q + KR * u < a.nk) pending |= 1u << u;
            // (a partial goes to the LDS staging the moment it is seen valid: no register array of them)
            if (t >= NCT) {
#pragma unroll
                for (int u = 0; u < U; ++u) { const int kq = 4 * wave + q + KR * u; if (kq < a.nk) red[kq * 16 + t] = 0.0; }
            }
            for (int spin = 0; __any(pending != 0u); ++spin) {
                if (spin >= PUB_SPIN_CAP) {   // never observed; a knot workgroup that died must not hang the device
#pragma unroll
                    for (int u = 0; u < U; ++u) if ((pending >> u) & 1u) red[(4 * wave + q + KR * u) * 16 + t] = __builtin_nan("");
                    if (lane == 0) { atomicMax(a.flag + b, a.seq); if (a.flag_host) a.flag_host[b] = a.seq; }
                    break;
                }
                // (eight pairs per pass: sixteen in flight at once would not fit the four-wave kernel's 128 VGPRs)
#pragma unroll
                for (int u0 = 0; u0 < U; u0 += 8) {
                    if (!__any(((pending >> u0) & 0xffu) != 0u)) continue;
                    unsigned long long v[8], g[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const unsigned long long* pp = base + (size_t(4 * wave + q + KR * (u0 + u)) * NCT + t) * 2;
                        const bool want = (pending >> (u0 + u)) & 1u;
                        v[u] = want ? __hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                        g[u] = want ? __hip_atomic_load(pp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (((pending >> (u0 + u)) & 1u) && (v[u] ^ g[u]) == pat) {
                            red[(4 * wave + q + KR * (u0 + u)) * 16 + t] = __longlong_as_double((long long)v[u]);
                            pending &= ~(1u << (u0 + u));
                        }
                }
            }
            __syncthreads();
            if (wave == 0) {
                // knot k belongs to group k % 16; a group is summed in ascending k; lane (t, q) holds the groups 4 w + q, w = 0..3
                double acc[4] = {0.0, 0.0, 0.0, 0.0};
                for (int k0 = q; k0 < a.nk; k0 += 32) {   // (eight LDS reads in flight; acc[w] takes its knots in ascending order)
#pragma unroll
                    for (int u = 0; u < 8; ++u) { const int kq = k0 + 4 * u; acc[u & 3] += kq < a.nk ? red[kq * 16 + t] : 0.0; }
                }
                double P[4];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    double pw = acc[w];
                    pw += __shfl_xor(pw, 16, 64);
                    pw += __shfl_xor(pw, 32, 64);
                    P[w] = pw;
                }
                const double term = ((P[0] + P[1]) + P[2]) + P[3];
                if (lane < NCT) a.cost_terms[size_t(b) * NCT + lane] = term;
                double tot = 0.0;
#pragma unroll
                for (int c = 0; c < NCT; ++c) tot += __shfl(term, c, 64);
                if (lane == 0) {
                    a.f[b] = tot;
                    if constexpr (PEERS) {   // this shard's cost, to every rank's slot for this rank
                        const int64_t o_f = int64_t(a.n) + a.nnz + a.m + a.peer_rank;
                        for (int r = 0; r < a.npeer; ++r) a.peer_out[r][o_f] = tot;
                    }
                }
            }
            return;
        }
    }
    const bool wg_m = SPLIT && blockIdx.x >= nkx;              // (SPLIT) this workgroup runs the model-free half of the knot program
    const unsigned bx = wg_m ? blockIdx.x - nkx : blockIdx.x;  // (both halves of a knot walk the XCDs the same way: they read the same records)
    const int kk = int((bx & 7u) * (nkx >> 3) + min(bx & 7u, nkx & 7u) + (bx >> 3));
    const int k = kb_p + kk;
    const int N = N_p;
    const double* x = x_p + size_t(b) * n_p;
    const int first = k == 0, last = k == N - 1;
    const DeviceTables& tb = *tb_p;
    // ---- stage the knot records and the tables in LDS: straight from global memory into LDS (global_load_lds_dwordx4: lane l of a wave
    // delivers 16 bytes at LDS base + 16 l — no VGPR round trip, no ds_write; one instruction moves 1 KB), the 1 KB chunks of all
    // blocks dealt round-robin over the waves.  (The previous version — every thread loads 8-byte words into registers and stores them
    // to LDS — spent ~250 instructions per wave here, a tenth of the 100-knot launch.)
    // Full layout: the whole HeadTables + GParams.  Compact layout: KSettings, the KinLite prefix of the kinematic tables and the
    // GParamsLite prefix of the parameters.
    // Copy-out tables of this knot's variant: fetched once the staging loads have landed (the compiler waits for EVERY outstanding
    // load before an LDS access that follows a direct load), consumed at the very end — they stay in flight across the (LDS-only)
    // barriers of the program.
    constexpr int JP_ITERS = VARY ? vary_cap(TERRAIN, WG) / WG : (js::COUNT + WG - 1) / WG, G_ITERS = (gs::COUNT + WG - 1) / WG;
    const int v = first ? VAR_FIRST : (last ? VAR_LAST : VAR_INTERIOR);
    int nnz_first = 0, nnz_interior = 0, n_glob = 0, jac_glob_base = 0;   // (scalars of the copy-out: read behind the vector loads, used at the end)
    int jpg = 0;
    int32_t jp[JP_ITERS], ga[G_ITERS], gb[G_ITERS];
    constexpr bool LATE_TABLES = five_per_cu<TERRAIN, WAVES, VARY, PEERS>;
    auto fetch_tables = [&]() __attribute__((always_inline)) {
        const DeviceTables& tbl = *tb_p;
        if constexpr (SPLIT) {   // this half's entries only (the other half's are -1 / G_NONE)
            const int hw = wg_m ? 1 : 0;
            jpg = tbl.split_jperm_glob[hw][tid & 15];
#pragma unroll
            for (int it = 0; it < JP_ITERS; ++it) jp[it] = VARY ? tbl.split_jpermv_t[hw][v][tid * JP_ITERS + it] : tbl.split_jperm_t[hw][v][tid * (JS_PAD / WG) + it];
#pragma unroll
            for (int it = 0; it < G_ITERS; ++it) {
                ga[it] = tbl.split_gab_t[hw][v][2 * (tid * (GS_PAD / WG) + it)];
                gb[it] = tbl.split_gab_t[hw][v][2 * (tid * (GS_PAD / WG) + it) + 1];
            }
            return;
        }
        jpg = tbl.jperm_glob[tid & 15];
#pragma unroll
        for (int it = 0; it < JP_ITERS; ++it) jp[it] = VARY ? tbl.jpermv_t[v][tid * JP_ITERS + it] : tbl.jperm_t[v][tid * (JS_PAD / WG) + it];      // (thread-major: wide loads)
#pragma unroll
        for (int it = 0; it < G_ITERS; ++it) {
            ga[it] = tbl.gab_t[v][2 * (tid * (GS_PAD / WG) + it)];
            gb[it] = tbl.gab_t[v][2 * (tid * (GS_PAD / WG) + it) + 1];
        }
    };
    // STORES_EARLY (launches into host memory that store early, early_out below): the tables are fetched WITH the staging loads and
    // CONSUMED — as far as the compiler can tell: an empty assembly statement that reads and "writes" them — right behind the staging wait.
    // Fetched behind that wait and left to their first real use, the compiler's wait for them stands behind early stores (between the
    // early entries of jac g and those of g, and again in front of the final stores): `s_waitcnt vmcnt(0)`, ONE counter for loads and
    // stores, i.e. the wave sits out the acknowledgement of what it has just sent to host memory before it may send the rest.
    static_assert(!STORES_EARLY || (WAVES == 8 && VARY && !PEERS), "launches that store early: the eight-wave VARY kernels of the host path");
    constexpr bool TABLES_WITH_STAGING = STORES_EARLY;
    auto tables_are_here = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < JP_ITERS; ++it) asm volatile("" : "+v"(jp[it]));
#pragma unroll
        for (int it = 0; it < G_ITERS; ++it) asm volatile("" : "+v"(ga[it]), "+v"(gb[it]));
        asm volatile("" : "+v"(jpg));
    };
    {
        static_assert(sizeof(KSettings) % 16 == 0 && sizeof(KinLite) % 16 == 0 && sizeof(GParamsLite) % 16 == 0 && sizeof(HeadTables) % 16 == 0 &&
                      sizeof(GParams) % 16 == 0 && (PK_STRIDE * 8) % 16 == 0, "staged in 16-byte pieces");
        // the record x_k is NXK = 189 doubles: 94 pieces + one double, which goes through a register like the gathered words below
        constexpr int XB = NXK * 8 / 16 * 16, XREM = NXK - XB / 8;
        static_assert(XREM <= 1 && NXG * 8 % 16 == 0 && NXG <= 8, "record staging");
        // (scalars of the copy-out, used at the end: read BEFORE the direct loads — behind them the compiler no longer treats global
        //  memory as unwritten and would fetch these through vector registers)
        nnz_first = tb.nnz_v[VAR_FIRST]; nnz_interior = tb.nnz_v[VAR_INTERIOR]; jac_glob_base = tb.jac_glob_base;
        n_glob = tb.n_glob;
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        // a block of `bytes` (multiple of 16): wave (w + rot) mod WAVES moves the 1 KB chunk w of each round of WG x 16 bytes; `rot`
        // walks on by the chunks used, so that the small blocks land on different waves (everything here folds at compile time
        // except the wave number: no scalar branch, one predicated instruction per block and round)
        int rot = 0;
        auto stage = [&](const void* src, void* dst, int bytes) __attribute__((always_inline)) {
            for (int r0 = 0; r0 < bytes; r0 += WG * 16) {
                const int wv = (wave_u + WAVES - rot % WAVES) & (WAVES - 1);
                const int off = r0 + wv * 1024 + lane * 16;
                if (off < bytes)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(static_cast<const char*>(src) + off),
                                                     (__attribute__((address_space(3))) void*)(static_cast<char*>(dst) + r0 + wv * 1024), 16, 0, 0);
                rot += ((bytes - r0 < WG * 16 ? bytes - r0 : WG * 16) + 1023) / 1024;
            }
        };
        stage(x + size_t(NXK) * k, s.x, XB);
        if (!first) stage(x + size_t(NXK) * (k - 1), s.xm, XB); else rot += (XB + 1023) / 1024;
        stage(pk_p + (size_t(b) * N + k) * PK_STRIDE, s.pk, PK_STRIDE * 8);
        stage(x + size_t(NXK) * N, s.xg, NXG * 8);
        if constexpr (COMPACT) {
            stage(static_cast<const GParamsLite*>(gp_p + b), &tabs.gp, int(sizeof(GParamsLite)));
            stage(&tb.head.ks, &tabs.ks, int(sizeof(KSettings)));
            stage(static_cast<const KinLite*>(&tb.head.kt), &tabs.kt, int(sizeof(KinLite)));
        } else {
#ifdef HIPNLP_DIAG_LITE_STAGE
            // timing-only diagnostic build (wrong values): the eight-wave kernel stages the lite tables only — 10.6 instead of 26.4 KB per
            // workgroup: what a shorter prologue could be worth on the 100-knot launch
            stage(static_cast<const GParamsLite*>(gp_p + b), &tabs.gp, int(sizeof(GParamsLite)));
            stage(static_cast<const KinLite*>(&tb.head.kt), &tabs.head.kt, int(sizeof(KinLite)));
            stage(&tb.head.ks, &tabs.head.ks, int(sizeof(KSettings)));
#else
            stage(gp_p + b, &tabs.gp, int(sizeof(GParams)));
            // (SPLIT: both halves stage every table.  The model-free half without the joint frames and link inertials it never reads — 5.2 of the
            //  8.4 KB — measured no different: 7.47 - 7.54 against 7.46 - 7.48 us per 100-knot launch)
            stage(&tb.head, &tabs.head, int(sizeof(HeadTables)));
#endif
        }
        // through registers: the odd last double of the two records; horizon ends only: the periodicity variables of the other end
        double xrem = 0.0, xov = 0.0;
        if (XREM && tid < 2 && !(first && tid == 1)) xrem = x[size_t(NXK) * (k - tid) + XB / 8];
        if constexpr (!Scratch::trimmed) { if (first || last) { if (tid < NPER) xov = x[size_t(NXK) * (first ? N - 1 : 0) + periodicity_row_var(tid)]; } }
#ifdef HIPNLP_STAMPS
        st_issued = __builtin_amdgcn_s_memtime();
#endif
        if constexpr (TABLES_WITH_STAGING) fetch_tables();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the staging loads of THIS wave have landed in LDS (before it signals the barrier)
        if constexpr (TABLES_WITH_STAGING) tables_are_here();

#ifdef HIPNLP_STAMPS
        st_loaded = __builtin_amdgcn_s_memtime();
#endif
        if constexpr (!Scratch::trimmed) { if (first || last) { if (tid < NPER) s.xo[tid] = xov; } }
        if (XREM && tid < 2) (tid ? s.xm : s.x)[XB / 8] = xrem;
        if (first) { for (int i = tid; i < XB / 8; i += WG) s.xm[i] = 0.0; }
        // pads (no staging load touches them)
        if (tid >= 64 && tid < 64 + Scratch::xpad - NXK) { s.x[NXK + tid - 64] = 0.0; s.xm[NXK + tid - 64] = 0.0; }
        if (tid >= 128 && tid < 128 + 8 - NXG) s.xg[NXG + tid - 128] = 0.0;
        // copy-out tables: issued now, consumed at the very end (LATE_TABLES: issued behind the last barrier but one instead — the
        // registers they occupy from here to the end are what stands between the four-wave VARY kernel and a fifth wave per SIMD)
        if constexpr (!LATE_TABLES && !TABLES_WITH_STAGING) fetch_tables();
    }
    lds_barrier();

    // ---- the knot's cost partials, published for the reducer workgroup of its trajectory (above) --------------------------------
    // One wave (PUBW) stores them as soon as they are final: at the end of
    // phase C for an ordinary knot; only the (rare) minimize-mode horizon-end terms of the first / last knot are final in phase F.
    constexpr int PUBW = WAVES == 8 ? 7 : 3;
    const bool ends_late = (first || last) && (tabs.settings().final_type == HIPNLP_EXPR_MINIMIZE || tabs.settings().periodicity_type == HIPNLP_EXPR_MINIMIZE);
    int pub_bad = 0;
    auto pub_store = [&]() __attribute__((always_inline)) {
        if (lane < NCT && (!SPLIT || split_cost_is_model_free(lane) == wg_m)) {   // (SPLIT: the terms this workgroup's half of the program owns)
            const double cv = (lane == CT_ENDS && !ends_late) ? 0.0 : s.cost[lane];   // (t_ends_finish writes that zero only in phase F)
            pub_bad |= !isfinite(cv);
            unsigned long long* pp = a.cost_pub + ((size_t(b) * a.nk + kk) * NCT + lane) * 2 + (wg_m ? size_t(gridDim.y) * a.nk * NCT * 2 : size_t(0));
            const unsigned long long bits = (unsigned long long)__double_as_longlong(cv);
            __hip_atomic_store(pp, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(pp + 1, bits ^ pub_pattern(a.seq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    // called by every wave right behind barrier number `passed` (0 = the one that ends phase A)
    auto pub_step = [&](int passed) __attribute__((always_inline)) {
        // (a.cost_pub == null: trajectories of more than 256 knots keep the separate reduction kernel)
        if (!a.cost_pub || wave != PUBW) return;
        if (passed == (ends_late ? 5 : 2)) pub_store();   // (every task group that writes a cost term runs in the first three phases, in every variant of the program)
    };

    KnotInfo ki{k, N, first, last};
    Em em{s.g, s.jac - Em::kJacOff};
    // (the full tables: global memory for the compact layouts; the full layout's LDS copy otherwise — the constructor's default)
    Ctx<Em> cx(s, tabs.kin(), tabs.settings(), tabs.gp, ki, em, COMPACT ? &tb.head.kt : nullptr, COMPACT ? gp_p + b : nullptr);
    if constexpr (Scratch::trimmed) cx.x_other = x + size_t(NXK) * (first ? N - 1 : 0);
    static_assert(!Scratch::hd_on_rw || hdyn_entries_early<Em>, "hd on Rw[1..]: only where the momentum rows are summed in the first phase (KnotScratchT::hd_on_rw)");
    // The program is instantiated ONCE PER WAVE (a generic lambda over the wave number as a compile-time constant: a task group is
    // compiled into the one instance whose wave runs it) and dispatched by one switch, so that every wave executes a CONTIGUOUS
    // instruction stream from the first phase to the last.  Written as `if (wave == w) { ... }` blocks phase after phase, every
    // wave jumped over the other waves' code several times per phase, each jump a cold instruction-cache line.  The waves still meet
    // at the same barriers: every instance contains all of them.
#define HIPNLP_W8(p, s) (TERRAIN == HIPNLP_TERRAIN_PLANAR ? (p) : (s))
#define HIPNLP_W4(p, s) (TERRAIN == HIPNLP_TERRAIN_PLANAR ? (p) : (s))
#ifdef HIPNLP_STAMPS
    unsigned long long st_task[24];   // diagnostic build: the time every task group of this wave ends, in program order
    int st_nt = 0;
#define DEV_R(w4, w8, fn, nt) if constexpr ((WAVES == 4 ? (w4) : (w8)) == W) { constexpr bool mf_ = split_task_is_model_free(#fn); if (!SPLIT || mf_ == wg_m) { for (int t_ = lane; t_ < (nt); t_ += 64) fn(cx, t_); } if (st_nt < 24) st_task[st_nt++] = __builtin_amdgcn_s_memtime(); }
#elif defined(HIPNLP_TASK_MARKS)
    // diagnostic compile (tools/diag/isa_tasks.sh: assembly only, never a library): comment markers around every task group, so that
    // tools/diag/isa_mix.py can say which task group carries how many instructions of which kind
#define DEV_R(w4, w8, fn, nt) if constexpr ((WAVES == 4 ? (w4) : (w8)) == W) { asm volatile("; TASK_BEGIN " #fn); for (int t_ = lane; t_ < (nt); t_ += 64) fn(cx, t_); asm volatile("; TASK_END " #fn); }
#elif defined(HIPNLP_DIAG_HALF_LATE)
    // timing-only diagnostic build (wrong values; tools/diag/headline_experiments.sh): the task groups of the LAST TWO phases — derivative
    // columns, row assembly — run on HALF their lanes, what one of TWO workgroups per knot would be left with if those phases were split by
    // column range.  Every group of those phases is one wave iteration whatever its lane count: the launch does not
    // get shorter (profiles/r05_headline_experiments.txt).
#define DEV_R(w4, w8, fn, nt) if constexpr ((WAVES == 4 ? (w4) : (w8)) == W) { for (int t_ = lane; t_ < (bid >= 4 ? ((nt) + 1) / 2 : (nt)); t_ += 64) fn(cx, t_); }
#elif defined(HIPNLP_DIAG_TASKSET)
    // timing-only diagnostic build (wrong values; tools/diag/taskset_experiment.sh): the knot program with ONE of its two halves —
    // 1: the task groups that need the robot model (joints, base, forward kinematics, links, composites, derivative columns, the
    // kinematic consistency rows), 2: the model-free ones (contact rows, linear defects, momentum dynamics, costs, horizon ends) — what a
    // workgroup of either kind would be left with if the two were specialised inside one launch
#define DEV_R(w4, w8, fn, nt) if constexpr ((WAVES == 4 ? (w4) : (w8)) == W && diag_task_kept(#fn)) { for (int t_ = lane; t_ < (nt); t_ += 64) fn(cx, t_); }
#else
    // (SPLIT: a workgroup runs the groups of its half — a scalar branch on a workgroup-uniform value; without SPLIT the test folds away)
#define DEV_R(w4, w8, fn, nt) if constexpr ((WAVES == 4 ? (w4) : (w8)) == W) { constexpr bool mf_ = split_task_is_model_free(#fn); if (!SPLIT || mf_ == wg_m) { for (int t_ = lane; t_ < (nt); t_ += 64) fn(cx, t_); } }
#endif
    // Early copy-out (eight-wave kernel, launches into host memory: a.early): behind the second barrier every thread stores those of ITS
    // entries of g and jac g whose slot is final by then (COPY_EARLY in its table words) — a third of jac g's varying entries and five rows
    // of g in six: 4.2 of a knot's 10.1 KB are on the link three microseconds before the program ends.  The copy-out at the end skips them
    // (its non-finite check still reads every slot).
    // (compiled into EVERY eight-wave instantiation behind a run-time test of a.early, the peer-store ones and the plain ones too, whose
    //  launches never arm it — only the STORES_EARLY instantiation is launched with a.early set: with the hook in some kernels and not in
    //  others the compiler contracts a few multiply-adds of the smooth-terrain tasks differently — last-bit differences between kernels
    //  whose outputs are compared bit for bit.  Tried again in round 5 with the hook as a compile-time property of the one instantiation:
    //  tests/test_gpu_constant_jacobian.py::test_batch_with_different_time_steps_and_the_stairs fails on the last bit of entries of jac g.)
    constexpr bool EARLY_OUT = WAVES == 8;
    // A store the COMPILER does not know about (inline assembly): behind an ordinary store it puts `s_waitcnt vmcnt(0)` in front of the
    // next LDS access (this kernel has used LDS-direct loads, and the counter is shared), i.e. the wave would sit out the round trip of its
    // early stores over PCIe at the next barrier — measured: no gain at all.  Nothing here waits for these stores; waits the compiler
    // places for its own later loads can only wait longer (the counter is in order).  base: uniform; byte_off < 2^31.
    auto store_unwaited = [](double* base, int byte_off, double v) __attribute__((always_inline)) {
        asm volatile("global_store_dwordx2 %0, %1, %2 sc0 sc1" ::"v"(byte_off), "v"(v), "s"(base) : "memory");   // (system scope: written through the L2)
    };
    const bool early_on = EARLY_OUT && a.early != 0;
    auto early_out = [&]() __attribute__((always_inline)) {
        if (!early_on) return;
        if (a.jac) {
            const int64_t jb0 = first ? 0 : (VARY ? int64_t(a.jb_first) + int64_t(k - 1) * a.jb_interior : int64_t(nnz_first) + int64_t(k - 1) * nnz_interior);
            double* out = a.jac + int64_t(b) * a.jac_stride + (jb0 - a.jac_off);
#pragma unroll
            for (int it = 0; it < JP_ITERS; ++it) if (jp[it] >= 0 && (jp[it] & COPY_EARLY)) store_unwaited(out, (VARY ? jp_pos(jp[it]) : tid + it * WG) * 8, em.jac[jp_slot(jp[it])]);
        }
        if (a.g) {
            double* out = a.g + size_t(b) * a.m;
#pragma unroll
            for (int it = 0; it < G_ITERS; ++it) if (ga[it] != G_NONE && (gb[it] & COPY_EARLY)) store_unwaited(out, (ga[it] + gb_stride(gb[it]) * k) * 8, s.g_at(tid + it * WG));
        }
    };
#ifdef HIPNLP_STAMPS
    // diagnostic build: every wave keeps, IN REGISTERS, its arrival time at each barrier and the time it leaves it; one store
    // per wave at the very end (a store before a barrier would make the barrier wait for its acknowledgement).
    unsigned long long st_arr[8], st_dep[8];
    const unsigned long long st_staged = __builtin_amdgcn_s_memtime();
    int bid = 0;
#define DEV_BARRIER st_arr[bid] = __builtin_amdgcn_s_memtime(); lds_barrier(); st_dep[bid] = __builtin_amdgcn_s_memtime(); if constexpr (W == PUBW) pub_step(bid); if constexpr (LATE_TABLES) { if (bid == 4) fetch_tables(); } if constexpr (EARLY_OUT) { if (bid == 1) early_out(); } bid++;
#else
    int bid = 0;
#define DEV_BARRIER lds_barrier(); if constexpr (W == PUBW) pub_step(bid); if constexpr (LATE_TABLES) { if (bid == 4) fetch_tables(); } if constexpr (EARLY_OUT) { if (bid == 1) early_out(); } bid++;
#endif
    auto run_wave = [&](auto wc) __attribute__((always_inline)) {
        constexpr int W = decltype(wc)::value;
        HIPNLP_KNOT_PROGRAM(DEV_R, DEV_BARRIER)
    };
#ifdef HIPNLP_TWOPASS
    // diagnostic: the knot program twice; the stamps of the FIRST pass are kept beside those of the second (instruction fetch: the
    // second pass finds its code in the instruction cache)
    unsigned long long st_first[8], st_first0 = 0;
    for (int pass_ = 0; pass_ < 2; ++pass_) {
    if (pass_ == 1) { for (int i = 0; i < 8; ++i) st_first[i] = st_arr[i]; st_first0 = st_staged; bid = 0; st_nt = 0; __syncthreads(); st_first[7] = __builtin_amdgcn_s_memtime(); }
#endif
    // (on the wave number as a SCALAR: dispatched on the per-lane value the switch is built out of execution masks — every instance skipped
    //  by a branch on an empty mask — and the path that skips them all reaches the copy-out with the table fetch unconsumed: the wait that
    //  tables_are_here() moves in front of the early stores would be back in front of the final ones)
    switch (__builtin_amdgcn_readfirstlane(wave)) {
        case 0: run_wave(std::integral_constant<int, 0>{}); break;
        case 1: run_wave(std::integral_constant<int, 1>{}); break;
        case 2: run_wave(std::integral_constant<int, 2>{}); break;
        case 3: run_wave(std::integral_constant<int, 3>{}); break;
        case 4: if constexpr (WAVES == 8) run_wave(std::integral_constant<int, 4>{}); break;
        case 5: if constexpr (WAVES == 8) run_wave(std::integral_constant<int, 5>{}); break;
        case 6: if constexpr (WAVES == 8) run_wave(std::integral_constant<int, 6>{}); break;
        default: if constexpr (WAVES == 8) run_wave(std::integral_constant<int, 7>{}); break;
    }
#ifdef HIPNLP_TWOPASS
    }
#endif
#undef DEV_R
#undef DEV_BARRIER
#undef HIPNLP_W8
#undef HIPNLP_W4

    // ---- stream the knot's outputs ---------------------------------------------------------------------
    // All LDS reads and the non-finite check come first, then nothing but stores; the non-finite flag is kept per wavefront
    // (the reduction kernel ORs them), so no workgroup barrier stands between the last task and the stores.
    const int64_t jbase = first ? 0 : (VARY ? int64_t(a.jb_first) + int64_t(k - 1) * a.jb_interior : int64_t(nnz_first) + int64_t(k - 1) * nnz_interior);
    int bad = 0;
    constexpr int GR_ITERS = (NXK + WG - 1) / WG;
    double jvals[JP_ITERS], gvals[G_ITERS], grvals[GR_ITERS];
    // (unconditional, clamped indices, no branch: every LDS read of the copy-out is in flight before the first wait)
#pragma unroll
    for (int it = 0; it < JP_ITERS; ++it) jvals[it] = em.jac[jp[it] >= 0 ? jp_slot(jp[it]) : Em::kJacOff];
#pragma unroll
    for (int it = 0; it < G_ITERS; ++it) gvals[it] = s.g_at((tid + it * WG) < gs::COUNT ? tid + it * WG : 0);
#pragma unroll
    for (int it = 0; it < GR_ITERS; ++it) grvals[it] = s.grad[(tid + it * WG) < NXK ? tid + it * WG : 0];
#pragma unroll
    for (int it = 0; it < JP_ITERS; ++it) bad |= (jp[it] >= 0) & !isfinite(jvals[it]);
#pragma unroll
    for (int it = 0; it < G_ITERS; ++it) bad |= (ga[it] != G_NONE) & !isfinite(gvals[it]);
#pragma unroll
    for (int it = 0; it < GR_ITERS; ++it) bad |= ((tid + it * WG) < NXK) & (!SPLIT || split_grad_is_model_free(tid + it * WG) == wg_m) & !isfinite(grvals[it]);
    if (a.cost_pub) { if (wave == PUBW) bad |= pub_bad; }
    else if (tid < NCT) {
        const double cval = s.cost[tid];
        bad |= !isfinite(cval);
        a.cost_knot[(size_t(b) * a.nk + kk) * NCT + tid] = cval;
    }
#ifdef HIPNLP_DIAG_TASKSET
    const int anybad = 0 * __any(bad);   // (the half of the program that is compiled out leaves its slots unwritten: their garbage must not raise the flag — 12 800 atomics on 16 words doubled the stairs launch)
#else
    const int anybad = __any(bad);
#endif
#ifdef HIPNLP_DIAG_SKIP
    if (HIPNLP_DIAG_SKIP & 1) a.jac = nullptr;
    if (HIPNLP_DIAG_SKIP & 2) a.g = nullptr;
    if (HIPNLP_DIAG_SKIP & 4) a.grad = nullptr;
#endif
    if constexpr (PEERS) {
        // peer mode: the same stores as below, once per rank, into that rank's buffer at the entries' final positions (jbase, the g
        // row map and the knot's grad offset are those of the WHOLE problem); the transfers over the links overlap with the knot
        // programs of the workgroups still running — no push pass, no reassembly pass behind the kernel
        const int64_t o_jac = a.n, o_g = int64_t(a.n) + a.nnz;
        for (int r = 0; r < a.npeer; ++r) {
            double* base = a.peer_out[r];
            double* oj = base + o_jac + jbase;
#pragma unroll
            for (int it = 0; it < JP_ITERS; ++it) if (jp[it] >= 0) oj[VARY ? jp_pos(jp[it]) : tid + it * WG] = jvals[it];
            // (VARY: every rank's buffer holds the constants — hipnlp_fill_jac_constants — the horizon-global entries among them)
            if constexpr (!VARY) { if (last && tid < n_glob) base[o_jac + int64_t(jac_glob_base) + tid] = s.jac[jpg]; }
            double* og = base + o_g;
#pragma unroll
            for (int it = 0; it < G_ITERS; ++it) if (ga[it] != G_NONE) og[ga[it] + gb_stride(gb[it]) * k] = gvals[it];
            double* ogr = base + int64_t(NXK) * k;
#pragma unroll
            for (int it = 0; it < GR_ITERS; ++it) { const int i = tid + it * WG; if (i < NXK) ogr[i] = grvals[it]; }
            if (last && tid < NXG) base[int64_t(NXK) * N + tid] = 0.0;
        }
    }
    if (a.jac) {
        double* out = a.jac + int64_t(b) * a.jac_stride + (jbase - a.jac_off);
#pragma unroll
        for (int it = 0; it < JP_ITERS; ++it) if (jp[it] >= 0 && !(early_on && (jp[it] & COPY_EARLY))) out[VARY ? jp_pos(jp[it]) : tid + it * WG] = jvals[it];
        // entries in the horizon-global columns (constants) sit right behind the last knot's block: the last knot writes them
        if constexpr (!VARY) { if (last && tid < n_glob && (!SPLIT || jpg >= 0)) a.jac[int64_t(b) * a.jac_stride + (int64_t(jac_glob_base) - a.jac_off) + tid] = s.jac[SPLIT ? max(jpg, 0) : jpg]; }
    }
    if (a.g) {
        double* out = a.g + size_t(b) * a.m;
#pragma unroll
        for (int it = 0; it < G_ITERS; ++it) if (ga[it] != G_NONE && !(early_on && (gb[it] & COPY_EARLY))) out[ga[it] + gb_stride(gb[it]) * k] = gvals[it];
    }
    if (a.g_stage) {
        bool dense = false;
        if constexpr (VARY) dense = a.gsb_first > 0;
        if (dense) {   // compact staging: the rows this knot owns behind one another, knot behind knot (no zeros for the slots it does not own)
            double* out = a.g_stage + int64_t(b) * a.gs_stride + ((first ? 0 : int64_t(a.gsb_first) + int64_t(k - 1) * a.gsb_interior) - a.gs_off);
#pragma unroll
            for (int it = 0; it < G_ITERS; ++it) if (ga[it] != G_NONE) out[gb_pos(gb[it])] = gvals[it];
        } else {
            double* out = a.g_stage + (size_t(b) * a.nk + kk) * gs::COUNT;
#pragma unroll
            for (int it = 0; it < G_ITERS; ++it) {
                const int slot = tid + it * WG;
                if (slot < gs::COUNT) out[slot] = ga[it] != G_NONE ? gvals[it] : 0.0;
            }
        }
    }
    if (a.grad) {
        double* out = a.grad + int64_t(b) * a.grad_stride + (int64_t(NXK) * k - a.grad_off);
#pragma unroll
        for (int it = 0; it < GR_ITERS; ++it) { const int i = tid + it * WG; if (i < NXK && (!SPLIT || split_grad_is_model_free(i) == wg_m)) out[i] = grvals[it]; }
        if (last && tid < NXG && !wg_m) a.grad[int64_t(b) * a.grad_stride + (int64_t(NXK) * N - a.grad_off) + tid] = 0.0;  // the global variables carry no cost
    }
    if (anybad && lane == 0) {   // generation flag: nothing to reset between launches
        atomicMax(a.flag + b, a.seq);
        if (a.flag_host) a.flag_host[b] = a.seq;
    }
#ifdef HIPNLP_STAMPS
    {
        st_arr[bid] = __builtin_amdgcn_s_memtime();   // after the vote and the store issue
        unsigned long long* stamp_out = a.stamps + ((size_t(blockIdx.y) * nkx + blockIdx.x) * 8 + wave) * 128;   // (knot workgroups only: nkx per row)
        if (lane == 0) {
            stamp_out[0] = st_entry; stamp_out[1] = st_staged; stamp_out[2] = (unsigned long long)bid;
            for (int i = 0; i < 8; ++i) { stamp_out[8 + 2 * i] = i <= bid ? st_arr[i] : 0; stamp_out[9 + 2 * i] = i < bid ? st_dep[i] : 0; }
            stamp_out[3] = st_real0; stamp_out[4] = __builtin_amdgcn_s_memrealtime(); stamp_out[5] = st_issued; stamp_out[6] = st_loaded;
            stamp_out[7] = (unsigned long long)st_nt;
#ifdef HIPNLP_TWOPASS
            stamp_out[64] = st_first0; for (int i = 0; i < 8; ++i) stamp_out[65 + i] = st_first[i];
#endif
            for (int i = 0; i < 24; ++i) stamp_out[32 + i] = i < st_nt ? st_task[i] : 0;
        }
    }
#endif
}

// =====================================================================================================================
// Exact Hessian of the Lagrangian (knot_hess_body.h): one workgroup of four waves per knot runs the KINEMATIC part of the knot program
// (none of its rows or Jacobian columns) with the Hessian tasks that need no kinematics on its idle waves, then the kinematic
// Hessian tasks; the knot's block of the triplet values leaves as one contiguous run.  Multipliers arrive in the reference's row
// order and are gathered through the same slot -> row map that scatters g.
// =====================================================================================================================
#if HIPNLP_HESS_DIAG_PHASES >= 6
#define HIPNLP_HESS_TABLES_AT 5   // behind the barrier that ends the fifth phase
#else
#define HIPNLP_HESS_TABLES_AT HIPNLP_HESS_DIAG_PHASES   // (diagnostic builds that stop early: behind the last barrier they have)
#endif
constexpr int HK_PAD = (hk::COUNT + 7) / 8 * 8;   // (16-byte pieces for the staging loads)
struct alignas(16) HessTables {
    KHFarLists far;               // unrelated joint pairs of the model (first: 16-byte aligned for the staging loads)
    int32_t perm[hk::COUNT];      // position in the knot block -> native slot
    int32_t perm_couple[84];
    int32_t nnz_knot, n_couple;
    // DIRECT instantiation: native slot -> position in the knot's block of the value run (the periodicity coupling entries, which the last
    // knot writes right behind its block: nnz_knot + i), -1: not in the pattern
    alignas(16) int16_t inv[HK_PAD];
};
struct HArgs {
    const DeviceTables* tb;
    const HessTables* ht;
    const double* x;        // [batch][n]
    const double* pk;
    const GParams* gp;
    const double* sigma;    // [batch]
    const double* lambda;   // [batch][m]
    double* hess;           // [batch][nnz_h]
    int32_t* flag;          // [batch] generation flag: == seq after a launch in which the trajectory produced a non-finite value
    int32_t* flag_host;     // [batch] or null: the same flag in pinned host memory (plain store of seq, see KArgs)
    int32_t N, n, m, knot_begin, seq, nnz_knot /* entries of a knot's block (DIRECT: the block base is needed before the tables are) */;
    int64_t hstride, hoff;  // values of trajectory b start at hess + b * hstride; the handle's first knot block sits at -hoff
    // launches into HOST memory (full layout): the entries [0, early_run) of a knot's block — the point columns come first in (column,
    // row) order — are all emitted in the first three phases (HessLayout::early_run: recorded, every entry is emitted exactly once); they
    // leave behind the barrier that makes them final (hess_early_phase: the second on the planar terrain, the third on the smooth steps),
    // written through the L2, while the kinematic phases still run: a quarter (planar) to two fifths (smooth steps) of a knot's bytes are
    // on the link three phases before the program ends.  0: everything at the end.
    int32_t early_run;
    double* trash;          // DIRECT launches: [workgroups] where the emitter puts a value whose slot is not in the pattern (see DevEmH::H)
#ifdef HIPNLP_STAMPS
    unsigned long long* stamps;  // diagnostic build only (tools/diag/hess_stamps.py): [blocks][8][128] s_memtime per wave
#endif
};
// DIRECT: every entry is emitted exactly once per knot (the recorder refuses a second emission, the host emulation poisons the staging and
// finds every entry of the pattern written at every knot) — so an entry can go straight to its place in the destination instead of
// through 17.5 KB of LDS staging and a copy-out pass: `hess` is the knot's block in the value run, `inv` the slot -> position table (an
// LDS copy), `nf` turns NaN when a non-finite value is emitted (v * 0 is NaN for NaN and Inf, +-0 otherwise).  Scattered 8-byte stores:
// for DEVICE destinations only (they meet in the L2; on PCIe fragments cost more than everything they save).
template <int TERRAIN, int LAYOUT, bool DIRECT = false> struct DevEmH {
    static constexpr int kTerrain = TERRAIN;
    static constexpr bool kBatch = LAYOUT == LAYOUT_COMPACT;   // (knot_hess_body.h kh_batch: the placement of task groups in launches longer than the chip)
    using Scratch = KnotScratchT<LAYOUT>;
    double* g;
    double* jac;
    double* hess;
    const int16_t* inv = nullptr;
    double* trash = nullptr;   // (DIRECT: this workgroup's word for values outside the pattern)
    double nf = 0.0;
    __device__ __forceinline__ void G(int slot, int, double v) { g[slot] = v; }
    __device__ __forceinline__ void J(int slot, int, int, double v) { jac[slot] = v; }
    __device__ __forceinline__ void H(int slot, int, int, double v) {
        if constexpr (DIRECT) {
            // (no branch around the store: with one, every emission was its own basic block — slot -> position read, wait, compare,
            //  store, one after the other, 380 cycles per entry at batch, tools/diag/hess_stamps.py; a value outside the pattern goes to a
            //  word of the workgroup's own instead, and the position reads of a task group's entries are in flight together)
            const int p = inv[slot];
            double* dst = p >= 0 ? hess + p : trash;
            *dst = v;
            // (every emitted value, pattern entry or not — ADVICE r04 asked for pattern entries only, as the staged kernel's copy-out; with the
            //  accumulation inside the branch the compiler contracted multiply-adds of the tasks differently and 88 entries of the
            //  "periodicity as a cost" case lost their bit-identity with the staged kernel (r05_gputest1.log): the stronger property is kept.
            //  A slot outside the pattern carries a zero multiplier; it is non-finite only where x itself is, and then pattern entries are too.)
            nf += v * 0.0;
        } else hess[slot] = v;
    }
};

// LDS.  The Hessian program runs none of the tasks that emit Jacobian entries or the cost gradient: the last members of the knot scratch
// (grad, jac: 17.2 KB) are not allocated.  The smooth terrain's point tasks hand their jets over through the per-joint spatial vectors of
// the Hessian scratch, which are dead until the kinematic Hessian tasks (knot_hess_body.h: pp_stage, kh_bump_part).
//   LAYOUT_FULL    (launches that are resident at once: knots x batch <= 512): 64 KB, two workgroups per CU;
//   LAYOUT_COMPACT (longer launches): the compact scratch and the lite tables of the four-wave callback kernel — own[] on the joint
//                  records, the once-per-knot tables read from global memory, no staging of the horizon-end multipliers (no Hessian task
//                  reads them: those rows are linear) — 52 KB: THREE workgroups per CU, 168 VGPRs.
//   DIRECT (compact layout, planar terrain, device destinations): no staging of the entries at all (DevEmH) — 40.5 KB: FOUR workgroups per
//                  CU, 128 VGPRs.
constexpr int hess_early_phase(int terrain) { return terrain == HIPNLP_TERRAIN_PLANAR ? 2 : 3; }
template <int TERRAIN, int LAYOUT, bool DIRECT = false> __global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(DIRECT ? 4 : (LAYOUT == LAYOUT_COMPACT ? 3 : 2), DIRECT ? 4 : (LAYOUT == LAYOUT_COMPACT ? 3 : 2))))
void hipnlp_knot_hess_kernel(HArgs a) {
    static_assert(!DIRECT || LAYOUT == LAYOUT_COMPACT, "the direct stores belong to the batch launches");
    using Em = DevEmH<TERRAIN, LAYOUT, DIRECT>;
    using Scratch = typename Em::Scratch;
    constexpr bool COMPACT = Scratch::compact;
    static_assert(offsetof(Scratch, jac) + sizeof(Scratch::jac) + 16 > sizeof(Scratch) && offsetof(Scratch, grad) + sizeof(Scratch::grad) == offsetof(Scratch, jac),
                  "grad and jac are the last members of the scratch (up to tail padding)");
    constexpr size_t S_BYTES = offsetof(Scratch, grad);
    static_assert(S_BYTES % 16 == 0, "scratch prefix");
    constexpr int WG = 256;
    __shared__ alignas(16) double s_raw[S_BYTES / sizeof(double)];
    Scratch& s = *reinterpret_cast<Scratch*>(s_raw);
    __shared__ SharedTablesT<COMPACT> tabs;
    static_assert(offsetof(KHessScratch, H) + sizeof(KHessScratch::H) == sizeof(KHessScratch) && offsetof(KHessScratch, H) % 16 == 0, "H is the last member of the Hessian scratch");
    constexpr size_t HX_BYTES = DIRECT ? offsetof(KHessScratch, H) : sizeof(KHessScratch);
    __shared__ alignas(16) double hx_raw[HX_BYTES / sizeof(double)];
    KHessScratch& hx = *reinterpret_cast<KHessScratch*>(hx_raw);
    __shared__ alignas(16) int16_t inv_s[DIRECT ? HK_PAD : 8];
    static_assert(S_BYTES + sizeof(SharedTablesT<COMPACT>) + HX_BYTES + sizeof(inv_s) <= (DIRECT ? 40960 : (COMPACT ? 160 * 1024 / 3 : 80 * 1024)), "workgroups per CU");
#ifdef HIPNLP_STAMPS
    const unsigned long long st_entry = __builtin_amdgcn_s_memtime();
#endif
    constexpr int G_USED = COMPACT ? gs::FIN : gs::COUNT;   // multipliers staged by native slot (compact: the horizon-end rows have no storage in g[])
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = blockIdx.y;
    const int kk = int((blockIdx.x & 7u) * (gridDim.x >> 3) + min(blockIdx.x & 7u, gridDim.x & 7u) + (blockIdx.x >> 3));   // (XCD-aware, as in hipnlp_knot_kernel)
    const int k = a.knot_begin + kk, N = a.N;
    const int first = k == 0, last = k == N - 1;
    const double* x = a.x + size_t(b) * a.n;
    const DeviceTables& tb = *a.tb;
    {
        // staged as in hipnlp_knot_kernel: records and tables straight from global memory into LDS, 1 KB chunks dealt over the waves
        constexpr int WAVES = WG / 64;
        constexpr int XB = NXK * 8 / 16 * 16, XREM = NXK - XB / 8;
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        int rot = 0;
        auto stage = [&](const void* src, void* dst, int bytes) __attribute__((always_inline)) {
            for (int r0 = 0; r0 < bytes; r0 += WG * 16) {
                const int wv = (wave_u + WAVES - rot % WAVES) & (WAVES - 1);
                const int off = r0 + wv * 1024 + lane * 16;
                if (off < bytes)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(static_cast<const char*>(src) + off),
                                                     (__attribute__((address_space(3))) void*)(static_cast<char*>(dst) + r0 + wv * 1024), 16, 0, 0);
                rot += ((bytes - r0 < WG * 16 ? bytes - r0 : WG * 16) + 1023) / 1024;
            }
        };
        // multipliers of the rows this knot owns, by native slot (the same slot -> row map that scatters g), in the g staging area of
        // the scratch (the Hessian program emits no g); of the next interval's angular momentum rows.  Two dependent global reads
        // (row map, then multiplier): all row-map words first, all multipliers behind them — two round trips, not two per iteration.
        const double* lam = a.lambda + size_t(b) * a.m;
        const int v = first ? VAR_FIRST : (last ? VAR_LAST : VAR_INTERIOR);
        constexpr int LG_ITERS = (gs::COUNT + WG - 1) / WG;
        int lga[LG_ITERS], lgb[LG_ITERS];
#pragma unroll
        for (int it = 0; it < LG_ITERS; ++it) { lga[it] = tb.g_a[v][tid + it * WG]; lgb[it] = tb.g_b[tid + it * WG]; }   // (padded tables)
        int nga = G_NONE, ngb = 0;
        if (tid < 3 && k + 1 < N) { const int slot = gs::HDYN + 3 + tid; nga = tb.g_a[k + 1 == N - 1 ? VAR_LAST : VAR_INTERIOR][slot]; ngb = tb.g_b[slot]; }
        const double sig = a.sigma[b];
        stage(x + size_t(NXK) * k, s.x, XB);
        if (!first) stage(x + size_t(NXK) * (k - 1), s.xm, XB); else rot += (XB + 1023) / 1024;
        stage(a.pk + (size_t(b) * N + k) * PK_STRIDE, s.pk, PK_STRIDE * 8);
        stage(x + size_t(NXK) * N, s.xg, NXG * 8);
        if constexpr (COMPACT) {
            stage(static_cast<const GParamsLite*>(a.gp + b), &tabs.gp, int(sizeof(GParamsLite)));
            stage(&tb.head.ks, &tabs.ks, int(sizeof(KSettings)));
            stage(static_cast<const KinLite*>(&tb.head.kt), &tabs.kt, int(sizeof(KinLite)));
        } else {
            stage(a.gp + b, &tabs.gp, int(sizeof(GParams)));
            stage(&tb.head, &tabs.head, int(sizeof(HeadTables)));
        }
        stage(&a.ht->far, &hx.far, int(sizeof(KHFarLists)));
        if constexpr (DIRECT) stage(a.ht->inv, inv_s, int(sizeof(inv_s)));
        double xrem = 0.0, xov = 0.0;
        if (XREM && tid < 2 && !(first && tid == 1)) xrem = x[size_t(NXK) * (k - tid) + XB / 8];
        if (first || last) { if (tid < NPER) xov = x[size_t(NXK) * (first ? N - 1 : 0) + periodicity_row_var(tid)]; }
        double lv[LG_ITERS];
#pragma unroll
        for (int it = 0; it < LG_ITERS; ++it) lv[it] = lga[it] != G_NONE ? lam[lga[it] + lgb[it] * k] : 0.0;
        const double lnext = nga != G_NONE ? lam[nga + ngb * (k + 1)] : 0.0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the staging loads of THIS wave have landed in LDS
        if (first || last) { if (tid < NPER) s.xo[tid] = xov; }
        if (XREM && tid < 2) (tid ? s.xm : s.x)[XB / 8] = xrem;
        if (first) { for (int i = tid; i < XB / 8; i += WG) s.xm[i] = 0.0; }
        if (tid >= 64 && tid < 64 + XPAD - NXK) { s.x[NXK + tid - 64] = 0.0; s.xm[NXK + tid - 64] = 0.0; }
        if (tid >= 128 && tid < 128 + 8 - NXG) s.xg[NXG + tid - 128] = 0.0;
#pragma unroll
        for (int it = 0; it < LG_ITERS; ++it) { const int slot = tid + it * WG; if (slot < G_USED) s.g[slot] = lv[it]; }
        if (tid < 3) hx.lam_next[tid] = lnext;
        if (tid == 0) hx.sigma = sig;
    }
    lds_barrier();
    // copy-out permutation.  Compact layout (the batch launches, at their register caps): fetched behind the fifth barrier, so that its
    // latency hides behind the last phase and its registers (ten per lane on the smooth terrain) are free while the point tasks run.
    // Full layout (two workgroups per CU, 134 / 184 of 256 VGPRs): fetched at ENTRY.  Its launches into host memory send the run at the
    // start of the block early (early_out below); a fetch behind those stores is followed by a wait for it, and that wait — the counter is
    // shared and in order — sits out the acknowledgement of the early stores too: 0.3 - 0.6 MB of a hundred workgroups queued on the link.
    // Measured before this was understood: the early run a GAIN of 3.7 - 4.7 us per 100-knot Hessian in two sessions and a LOSS of 2 - 3.6 us
    // (planar) / 6 - 8 us (smooth steps: its run leaves one barrier later and is twice as long) in five others, by the box's link.
    constexpr int TABLES_AT = LAYOUT == LAYOUT_FULL ? 0 : HIPNLP_HESS_TABLES_AT;
    const HessTables& ht = *a.ht;
    constexpr int HP_ITERS = (hk::COUNT + WG - 1) / WG;
    int cnt = 0, hpc = -1;
    int32_t hp[HP_ITERS];
    auto fetch_tables = [&]() __attribute__((always_inline)) {
        cnt = ht.nnz_knot;
#pragma unroll
        for (int it = 0; it < HP_ITERS; ++it) { const int i = tid + it * WG; hp[it] = i < hk::COUNT ? ht.perm[i] : -1; }   // (padded with -1 on the host)
        hpc = (last && tid < 84) ? ht.perm_couple[tid] : -1;   // (padded with -1 on the host)
    };
    int bar = 0;   // barriers passed (the program is straight-line code: a constant at every use)
    if (!DIRECT && TABLES_AT == 0) fetch_tables();
    // ... and CONSUMED, as far as the compiler can tell, behind the first barrier of the program (an empty assembly statement that reads
    // and "writes" them: the wait for the fetch goes there, a phase after its issue, where it is free, and what the copy-out uses is no
    // longer the result of a load).  Left to the first real use — the copy-out at the
    // end — the wait would stand behind the early stores: `s_waitcnt vmcnt(0)`, the one counter for loads and stores, i.e. every wave would
    // sit out the acknowledgement of what it sent to host memory a few microseconds earlier before it may send the rest.
    auto tables_are_here = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < HP_ITERS; ++it) asm volatile("" : "+v"(hp[it]));
        asm volatile("" : "+v"(hpc), "+v"(cnt));
    };
    // early copy-out (see HArgs::early_run): full layout only — the compact kernels sit at their register caps, and their launches are the
    // batch ones, whose hundreds of workgroups spread their stores over the launch by themselves
    constexpr bool EARLY_OUT = !DIRECT && LAYOUT == LAYOUT_FULL;
    constexpr int EARLY_ITERS = 3;   // (early_run <= 768: 394 on the planar terrain, 732 on the smooth steps; hess_launch checks)
    static_assert(!EARLY_OUT || (TABLES_AT == 0 && EARLY_ITERS <= HP_ITERS), "the early run reads the permutation words fetched at entry");
    // the barrier behind which the run leaves: a constant of the program (two phases emit the planar point columns, three those of the smooth
    // steps), so that the stores stand in the code ONCE per wave, not behind every barrier; hess_launch arms the run only where the
    // recorded layout says the same (HessLayout::early_phase)
    constexpr int EARLY_PHASE = hess_early_phase(TERRAIN);
    const int early_run = EARLY_OUT ? a.early_run : 0;
    auto early_out = [&]() __attribute__((always_inline)) {
        if constexpr (EARLY_OUT) {
            if (early_run <= 0) return;
            double* out = a.hess + int64_t(b) * a.hstride + (int64_t(a.nnz_knot) * k - a.hoff);
#pragma unroll
            for (int it = 0; it < EARLY_ITERS; ++it) {
                const int pos = tid + it * WG;
                // (a store the compiler does not know about, as in hipnlp_knot_kernel: nothing waits for it)
                if (pos < early_run) asm volatile("global_store_dwordx2 %0, %1, %2 sc0 sc1" ::"v"(pos * 8), "v"(hx.H[hp[it]]), "s"(out) : "memory");
            }
        }
    };
    KnotInfo ki{k, N, first, last};
    Em em{s.g, s.jac, DIRECT ? a.hess + int64_t(b) * a.hstride + (int64_t(a.nnz_knot) * k - a.hoff) : hx.H, inv_s, DIRECT ? a.trash + (size_t(blockIdx.y) * gridDim.x + blockIdx.x) : nullptr};
    Ctx<Em> cx(s, tabs.kin(), tabs.settings(), tabs.gp, ki, em, COMPACT ? &tb.head.kt : nullptr, COMPACT ? a.gp + b : nullptr);
    KHCtx<Em> hcx{cx, hx, s.g};
    // (one contiguous program instance per wave, as in hipnlp_knot_kernel)
#ifdef HIPNLP_STAMPS
    unsigned long long st_task[24], st_arr[8], st_dep[8];   // diagnostic build: end of every task group of this wave in program order; barrier arrival / departure
    int st_nt = 0, bid = 0;
    const unsigned long long st_staged = __builtin_amdgcn_s_memtime();
#define DEV_KIN(w, fn, nt) if constexpr ((w) == W) { for (int t_ = lane; t_ < (nt); t_ += 64) fn(cx, t_); if (st_nt < 24) st_task[st_nt++] = __builtin_amdgcn_s_memtime(); }
#define DEV_RH(w, fn, nt) if constexpr ((w) == W) { for (int t_ = lane; t_ < (nt); t_ += 64) fn(hcx, t_); if (st_nt < 24) st_task[st_nt++] = __builtin_amdgcn_s_memtime(); }
#define DEV_BARRIER st_arr[bid] = __builtin_amdgcn_s_memtime(); lds_barrier(); st_dep[bid] = __builtin_amdgcn_s_memtime(); bid++; if (++bar == TABLES_AT) { if constexpr (!DIRECT) fetch_tables(); } if constexpr (!DIRECT && TABLES_AT == 0) { if (bar == 1) tables_are_here(); } if (bar == EARLY_PHASE) early_out();
#else
#define DEV_KIN(w, fn, nt) if constexpr ((w) == W) { for (int t_ = lane; t_ < (nt); t_ += 64) fn(cx, t_); }
#define DEV_RH(w, fn, nt) if constexpr ((w) == W) { for (int t_ = lane; t_ < (nt); t_ += 64) fn(hcx, t_); }
#define DEV_BARRIER lds_barrier(); if (++bar == TABLES_AT) { if constexpr (!DIRECT) fetch_tables(); } if constexpr (!DIRECT && TABLES_AT == 0) { if (bar == 1) tables_are_here(); } if (bar == EARLY_PHASE) early_out();
#endif
    auto run_wave = [&](auto wc) __attribute__((always_inline)) {
        constexpr int W = decltype(wc)::value;
        HIPNLP_KNOT_HESS_PROGRAM(DEV_KIN, DEV_RH, DEV_BARRIER)
    };
    // (dispatched on the wave number as a SCALAR: on the per-lane value the compiler builds the switch out of execution masks — every
    //  instance skipped by a branch on an empty mask — and a path that skips all four reaches the copy-out with the table fetch unconsumed:
    //  the wait tables_are_here() moves forward would be back in front of the final stores)
    switch (__builtin_amdgcn_readfirstlane(wave)) {
        case 0: run_wave(std::integral_constant<int, 0>{}); break;
        case 1: run_wave(std::integral_constant<int, 1>{}); break;
        case 2: run_wave(std::integral_constant<int, 2>{}); break;
        default: run_wave(std::integral_constant<int, 3>{}); break;
    }
#undef DEV_KIN
#undef DEV_RH
#undef DEV_BARRIER
    int bad = 0;
    if constexpr (DIRECT) {
        bad = cx.em.nf != cx.em.nf;   // (the entries are in place: every lane has watched what it emitted)
    } else {
        double* out = a.hess + int64_t(b) * a.hstride + (int64_t(cnt) * k - a.hoff);
        double hv[HP_ITERS];
#pragma unroll
        for (int it = 0; it < HP_ITERS; ++it) hv[it] = hx.H[hp[it] >= 0 ? hp[it] : 0];   // every LDS read in flight before the first store
        const double hvc = hx.H[hpc >= 0 ? hpc : 0];
#pragma unroll
        for (int it = 0; it < HP_ITERS; ++it) if (hp[it] >= 0) { bad |= !isfinite(hv[it]); if (tid + it * WG >= early_run) out[tid + it * WG] = hv[it]; }
        if (hpc >= 0) { bad |= !isfinite(hvc); a.hess[int64_t(b) * a.hstride + (int64_t(cnt) * N - a.hoff) + tid] = hvc; }
    }
    if (__any(bad) && lane == 0) {   // nothing to reset between launches
        atomicMax(a.flag + b, a.seq);
        if (a.flag_host) a.flag_host[b] = a.seq;
    }
#ifdef HIPNLP_STAMPS
    if (a.stamps && lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* o = a.stamps + ((size_t(blockIdx.y) * gridDim.x + blockIdx.x) * 8 + wave) * 128;
        o[0] = st_entry; o[1] = st_staged; o[2] = (unsigned long long)bid; o[3] = (unsigned long long)st_nt; o[4] = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 8; ++i) { o[8 + 2 * i] = i < bid ? st_arr[i] : 0; o[9 + 2 * i] = i < bid ? st_dep[i] : 0; }
        for (int i = 0; i < 24; ++i) o[32 + i] = i < st_nt ? st_task[i] : 0;
    }
#endif
}


// Throughput variant only (4 waves; the latency variant sums inside the knot kernel, see pub_step): f[b] = sum over knots and
// terms, cost_terms[b][t] = sum over knots.  One workgroup per trajectory.  Thread (group g = tid / 16, column c = tid % 16) sums
// column c of the knots g, g + 16, ...: all its loads are independent and issued together (ONE memory round trip), then a fixed
// tree: two lane-shuffles inside the wave, four waves through LDS, twelve terms in order.  Bitwise reproducible.
constexpr int RWG = 256, RCOL = 16, RGRP = RWG / RCOL, RUNR = 8;
static_assert(NCT <= RCOL, "reduce columns");
__global__ __launch_bounds__(RWG) void hipnlp_reduce_kernel(const double* cost_knot, int nk, double* f, double* cost_terms, ConstCheck cc) {
    __shared__ double part[RWG / 64][RCOL];
    const int b = blockIdx.x, tid = threadIdx.x, c = tid % RCOL, g = tid / RCOL, lane = tid & 63, wave = tid >> 6;
    __shared__ int cc_flags[RWG / 64];
    constants_check_and_repair(cc, b, RWG, cc_flags);   // (VARY launches into device memory; cc.jac == null otherwise)
    double acc = 0.0;
    for (int k0 = g; k0 < nk; k0 += RGRP * RUNR) {
        double v[RUNR];
#pragma unroll
        for (int u = 0; u < RUNR; ++u) {
            const int k = k0 + u * RGRP;
            v[u] = (c < NCT && k < nk) ? cost_knot[(size_t(b) * nk + k) * NCT + c] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < RUNR; ++u) acc += v[u];
    }
    acc += __shfl_xor(acc, 16, 64);
    acc += __shfl_xor(acc, 32, 64);
    if (lane < RCOL) part[wave][lane] = acc;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int t = 0; t < NCT; ++t) {
            const double term = ((part[0][t] + part[1][t]) + part[2][t]) + part[3][t];
            tot += term;
            if (cost_terms) cost_terms[size_t(b) * NCT + t] = term;
        }
        if (f) f[b] = tot;
    }
}

// Reassembly behind the all-gather of the knot-sharded path: out[i] = gathered[src[i]], and block 0 adds the
// per-rank cost partials in rank order.  One launch instead of an index-select, a second gather and a sum.
__global__ __launch_bounds__(256) void hipnlp_reassemble_kernel(const double* gathered, const int64_t* src, double* out, int64_t count,
                                                                int world, int64_t shard_len, double* f_out) {
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += stride) out[i] = gathered[src[i]];
    if (f_out && blockIdx.x == 0 && threadIdx.x == 0) {
        double f = 0.0;
        for (int r = 0; r < world; ++r) f += gathered[int64_t(r) * shard_len];
        *f_out = f;
    }
}

// ... and with a destination index (out[dst[i]] = gathered[src[i]]): the entries the shards did NOT send — the constant entries of jac g,
// put into `out` once per parameter set — are left alone.
__global__ __launch_bounds__(256) void hipnlp_reassemble_scatter_kernel(const double* gathered, const int64_t* src, const int64_t* dst, double* out, int64_t count,
                                                                        int world, int64_t shard_len, double* f_out) {
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += stride) out[dst[i]] = gathered[src[i]];
    if (f_out && blockIdx.x == 0 && threadIdx.x == 0) {
        double f = 0.0;
        for (int r = 0; r < world; ++r) f += gathered[int64_t(r) * shard_len];
        *f_out = f;
    }
}

// Host path, Jacobian asked for after its evaluation (IPOPT's eval_jac_g with new_x = FALSE): the varying run of every knot block from
// the complete values in HBM into a device-visible HOST array that already holds the constant entries (varying-first order of a block,
// HIPNLP_FLAG_JAC_VARYING_FIRST).  One workgroup per knot; consecutive lanes, consecutive addresses on the link.
__global__ __launch_bounds__(256) void hipnlp_fetch_vary_kernel(const double* __restrict__ src, double* __restrict__ dst, int kb, int N, int64_t nnz, int nnz_first,
                                                                 int nnz_interior, int nv_first, int nv_interior, int nv_last) {
    const int k = kb + int(blockIdx.x);
    const int64_t base = int64_t(blockIdx.y) * nnz + (k == 0 ? 0 : int64_t(nnz_first) + int64_t(k - 1) * nnz_interior);
    const int count = k == 0 ? nv_first : (k == N - 1 ? nv_last : nv_interior);
    for (int i = threadIdx.x; i < count; i += 256) dst[base + i] = src[base + i];
}

// The constant entries of the handle's knots into a DEVICE array jac [batch][nnz] (varying-first order: the tail of every block; CCS order:
// wherever they sit; and the horizon-global entries behind the last block), from the handle's templates: once per destination and parameter set.
__global__ __launch_bounds__(256) void hipnlp_fill_const_kernel(double* __restrict__ jac, const double* __restrict__ ctpl, const int32_t* __restrict__ ctpl_of_b, int ctpl_len,
                                                                 int off_first, int off_interior, int off_last, int off_glob, int kb, int N, int64_t nnz,
                                                                 int nnz_first, int nnz_interior, int nnz_last, int nv_first, int nv_interior, int nv_last, int n_glob) {
    const int k = kb + int(blockIdx.x), b = int(blockIdx.y);
    const double* t = ctpl + int64_t(ctpl_of_b[b]) * ctpl_len;
    const int64_t base = int64_t(b) * nnz + (k == 0 ? 0 : int64_t(nnz_first) + int64_t(k - 1) * nnz_interior);
    const int nv = k == 0 ? nv_first : (k == N - 1 ? nv_last : nv_interior), end = k == 0 ? nnz_first : (k == N - 1 ? nnz_last : nnz_interior);
    const double* tv = t + (k == 0 ? off_first : (k == N - 1 ? off_last : off_interior));
    // (nv: the block's first constant entry — behind the varying run of a varying-first block, anywhere in a CCS block, whose varying
    //  positions carry CTPL_VARYING in the template)
    for (int i = nv + int(threadIdx.x); i < end; i += 256) { const double c = tv[i]; if ((unsigned long long)__double_as_longlong(c) != CTPL_VARYING) jac[base + i] = c; }
    if (k == N - 1) for (int i = int(threadIdx.x); i < n_glob; i += 256) jac[base + end + i] = t[off_glob + i];
}

thread_local std::string g_create_error;

// Diagnostic overrides from the environment (kernel variant, launch numbering, A/B switches of one process): compiled into the test /
// measurement build only (-DHIPNLP_DIAG: tests/_build/libhipnlp_diag.so, same device code).  The shipped library reads NO environment
// variable: what a handle does is decided by hipnlp_desc and the hipnlp_set_* calls alone.
#ifdef HIPNLP_DIAG
inline const char* diag_env(const char* name) { return std::getenv(name); }
#else
inline const char* diag_env(const char*) { return nullptr; }
#endif

// host ranges registered through hipnlp_host_register: a caller-owned output array inside one of them is stored to DIRECTLY by the
// kernel (hipnlp_eval looks its pointers up here), no staging copy
struct HostRange { char* host; size_t bytes; char* dev; const void* owner; unsigned long long id; };   // owner: the handle that registered the range by itself, or null (the caller did); id: of THIS registration (never reused)
std::mutex g_ranges_mutex;
std::vector<HostRange> g_ranges;
unsigned long long g_range_ids = 0;   // (under g_ranges_mutex)
// the registration that answers for p (device_address_of's rule: the newest range that covers it), 0: none
unsigned long long registration_of(const void* p, size_t bytes) {
    if (!p) return 0;
    std::lock_guard<std::mutex> lock(g_ranges_mutex);
    const char* q = static_cast<const char*>(p);
    for (size_t i = g_ranges.size(); i-- > 0;) {
        const HostRange& r = g_ranges[i];
        if (q >= r.host && q + bytes <= r.host + r.bytes) return r.id;
    }
    return 0;
}
double* device_address_of(const void* p, size_t bytes) {
    if (!p) return nullptr;
    std::lock_guard<std::mutex> lock(g_ranges_mutex);
    const char* q = static_cast<const char*>(p);
    // (newest first: where a fresh registration and an older one that has gone stale both cover p, the fresh one answers)
    for (size_t i = g_ranges.size(); i-- > 0;) {
        const HostRange& r = g_ranges[i];
        if (q >= r.host && q + bytes <= r.host + r.bytes) return reinterpret_cast<double*>(r.dev + (q - r.host));
    }
    return nullptr;
}

}  // namespace

struct MultiState;
struct hipnlp_handle {
    hipnlp_desc d;
    // hipnlp_multi_create: this handle is the FRONT of one shard handle per device.  It owns the NLP's host side — layout, bounds, the
    // constant entries of jac g, the pinned staging of x and of the outputs, the bookkeeping of the caller's arrays — and no device
    // memory, no stream: every evaluation is the shards' (multi_* below).
    MultiState* multi = nullptr;
    bool is_front = false;
    Layout L;
    KinTables kt;
    int batch = 1, kb = 0, ke = 0, nk = 0, np = 0;
    bool wide = false;   // eight-wave kernel variant (launches that are resident at once at two workgroups per CU)
    // two workgroups per knot (SPLIT instantiations): device-resident launches of the eight-wave kernels that leave at least half of the
    // CUs idle — (2 knots + 1) x batch workgroups on 256 CUs, one each — and whose horizon ends are not costs (knot_body.h)
    bool split = false;
    bool fused = false;  // the total cost is summed inside the knot launch by one reducer workgroup per trajectory (knots <= 256)
    hipnlp_dims dims{};         // hipnlp_get_dims, filled by hipnlp_create
    bool early_store_always = false;   // diagnostic (HIPNLP_EARLY_STORE=2): also for hipnlp_eval_device launches
    bool early_store = true;    // launches into host memory store what is final after the second phase then (diagnostic override: HIPNLP_EARLY_STORE=0)
    // hipnlp_eval_hess*: the run at the start of every knot block leaves early (HArgs::early_run) — 1 / 0, or -1 = decided by the handle from
    // its own first calls (hipnlp_set_hessian_early_run).  The same kernel, the same values either way; which one is faster depends on
    // the HOST: 4 - 7 us sooner per 100-knot Hessian with the caller on the card's NUMA node, 2 - 3 us later from the other socket
    // (profiles/r05_early_stores_by_box.txt).
    int hess_early_mode = -1, hess_early_choice = -1, hess_tune_calls = 0;
    double hess_tune_best[2] = {1e30, 1e30};
    std::vector<double> hess_tune_us[2];   // (auto mode without a NUMA answer: the samples of each kind; their medians decide)
    int hess_early_why = 0;                // how the choice was made: HESS_WHY_*
    int hess_caller_node = -1, hess_card_node = -1;
    bool hess_compact = false;   // exact Hessian: compact-scratch instantiation (three workgroups per CU) for launches of more than 512 workgroups
    bool hess_direct = true;     // ... and, planar terrain into device memory, the instantiation without LDS staging of the entries (diagnostic override: HIPNLP_HESS_DIRECT=0)
    int dev = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timing_valid = false;
    hipEvent_t last_e0 = nullptr, last_e2 = nullptr;  // events around the last TIMED launch
    std::vector<hipEvent_t> prof_ev;  // triples
    int prof_cap = 0, prof_n = 0, prof_stride = 1, prof_seen = 0;
    int prof_run = 0;         // > 0: brackets of prof_run consecutive launches (hipnlp_profile_begin_runs)
    bool prof_open = false;
    DeviceTables* d_tb = nullptr;
    double *d_x = nullptr, *d_pk = nullptr, *d_g = nullptr, *d_jac = nullptr, *d_grad = nullptr, *d_f = nullptr;
    double *d_cost_knot = nullptr, *d_cost_terms = nullptr;
    GParams* d_gp = nullptr;
    int32_t* d_flag = nullptr;              // [batch] generation flag of the non-finite detector (== seq of the launch that raised it)
    unsigned long long* d_cost_pub = nullptr;  // [batch][nk][NCT][2] tagged partials of the eight-wave variant (zero = never valid)
    int32_t seq = 0, seq_result = -1;        // launches so far; launch the cached host result belongs to
    unsigned long long* d_stamps = nullptr;
    // pinned host staging
    double *h_x = nullptr, *h_g = nullptr, *h_jac = nullptr, *h_grad = nullptr, *h_f = nullptr, *h_cost_terms = nullptr;   // (h_* outputs: views of h_out)
    void *d_out = nullptr, *h_out = nullptr;   // output block [f | cost terms | grad | g | jac | flag] and its pinned mirror
    size_t out_bytes = 0;
    int32_t* h_flag = nullptr;
    // host-buffer path (hipnlp_eval / hipnlp_eval_pinned): the kernel stores the wanted outputs STRAIGHT into the pinned mirror and
    // reads x straight out of the pinned staging copy — device-visible addresses of the pinned blocks:
    double *hd_x = nullptr, *hd_f = nullptr, *hd_cost_terms = nullptr, *hd_grad = nullptr, *hd_g = nullptr, *hd_jac = nullptr;
    int32_t* hd_flag = nullptr;
    bool x_zero_copy = true;      // the kernel reads x from pinned host memory (small problems) instead of an H2D copy first
    bool lam_zero_copy = true;    // hipnlp_eval_hess: the kernel gathers sigma / lambda out of the pinned block too (small problems; HIPNLP_HESS_LAM_ZERO_COPY=0/1 overrides)
    unsigned prefetch = HIPNLP_WANT_F | HIPNLP_WANT_GRAD | HIPNLP_WANT_G;   // brought to the host by every new evaluation
    unsigned on_host = 0;         // outputs of the cached result that are in the pinned block
    unsigned gone = 0;            // outputs of the cached result the kernel stored into registered caller arrays (in neither block)
    // hipnlp_set_early_outputs: registered caller arrays seen in earlier calls (host address; the device address is looked up again at
    // every use: a range may have been unregistered since), and which outputs of the cached result already sit in them
    bool early = false, early_grad = false;   // early outputs: g and jac g; grad f too only with hipnlp_set_early_outputs(h, 2)
    double* seen_host[3] = {nullptr, nullptr, nullptr};   // grad, g, jac
    double* early_host[3] = {nullptr, nullptr, nullptr};
    unsigned early_mask = 0;
    // auto-registration of caller arrays (hipnlp_set_auto_register; on by default): an output array seen at the same address on two
    // consecutive calls is page-locked and mapped (hipHostRegister) so that the kernel stores straight into it; every use is verified
    // with a sentinel (below), at most AUTO_MAX ranges per handle, unregistered by hipnlp_destroy
    bool auto_reg = true;
    bool x_staged = false;        // h_x (and d_x, for handles whose launches read x from HBM) hold the x of the latest host-buffer call
    const void* last_seen[4] = {nullptr, nullptr, nullptr, nullptr};   // grad, g, jac (hipnlp_eval) and Hessian-value (hipnlp_eval_hess) pointers of the previous call (second sight registers)
    const void* in_call[4] = {nullptr, nullptr, nullptr, nullptr};     // output arrays of the call in progress (never evicted by it)
    const void* no_auto[4] = {nullptr, nullptr, nullptr, nullptr};     // pointers that failed to register or failed the sentinel check: left alone
    unsigned long long sentinel_salt = 0;
    long auto_registered = 0, auto_fallbacks = 0;
    // Constant entries of jac g (Layout::jconst_pos; 43 % of the pattern at N = 100: the +-1, -dt/2, mass entries of the linear rows).
    // In the varying-first order of a knot's block (HIPNLP_FLAG_JAC_VARYING_FIRST) a host destination of the Jacobian is FILLED with
    // them once per parameter set — the pinned block and a registered caller array at their first use — and the kernel then stores
    // the entries that depend on x only (d_tb_vary: the same tables with -1 at the constant positions of the copy-out permutation):
    // a third fewer bytes on the PCIe-bound path, one contiguous run per knot.  Caller arrays are spot-checked before every such
    // launch (csample) and re-filled when a check fails or the parameters changed.
    int gs_rows_v[3] = {0, 0, 0};                   // rows of g a first / interior / last knot owns (valid staging slots)
    bool gs_compact_ok = true;
    bool vary_ok = false;                           // the varying entries of every block fit the VARY instantiations' trip counts (either order of a block)
    bool vary_check = true;                         // the VARY kernels look at the constants they find in a device destination (HIPNLP_VARY_CHECK=0: diagnostic)
    double* d_ctpl = nullptr;                       // device copy of the templates [ctpl.size()][ctpl_len] (the VARY kernels' self-healing check, the fill kernel)
    int32_t* d_ctpl_of_b = nullptr;
    int32_t* d_healed = nullptr;                    // wave slices of constants a VARY kernel had to put back (the caller wrote over a device buffer)
    int ctpl_len = 0;
    struct DevFilled { const void* dev; unsigned long long gen; bool remote; unsigned launches; };   // remote: not this device's own memory (a registered host range, a peer's buffer): never read back by the launch; launches: into THIS buffer so far (which pair of sample positions the next one looks at)
    DevFilled dfilled[8] = {};                      // device jac buffers of hipnlp_eval_device that hold this handle's constants, and of which parameter set
    int dfilled_next = 0;
    long dev_const_fills = 0;
    bool skip_const = true;                         // hipnlp_set_constant_jacobian (hipnlp_create: on for varying-first handles, off — until asked for — for CCS handles)
    unsigned long long param_gen = 0;               // hipnlp_set_params calls so far
    std::vector<std::vector<double>> ctpl;          // distinct templates [block of VAR_FIRST | VAR_INTERIOR | VAR_LAST | horizon-global]: the constant values at their positions
    std::vector<int> ctpl_of_b;                     // template of trajectory b
    std::vector<int32_t> cpos[3];                   // constant positions of a variant-v block
    int ctpl_off[4] = {0, 0, 0, 0};
    // (reg: the registration — HostRange::id — the array was filled under.  An array that was unregistered since, by whoever, and comes back
    //  at the same address is a NEW registration: whatever its pages hold now, it is filled again before a launch skips the constants)
    struct ConstFilled { const void* host; unsigned long long gen, reg; };
    ConstFilled cfilled[8] = {};                    // caller arrays that hold this handle's constants, and of which parameter set
    int cfilled_next = 0;
    unsigned long long pinned_const_gen = 0;        // parameter set whose constants the pinned block holds (0: none)
    std::vector<std::pair<size_t, double>> csample; // (index into [batch][nnz], value): the spot check
    size_t jac_first_vary = 0, jac_last_vary = 0;   // first / last entry of [batch][nnz] that depends on x (sentinel words of a store that skips the constants)
    // first / last word of [batch][n], [batch][m], [batch][nnz] that THIS handle's launches write (a shard handle writes the entries of its
    // own knots: the sentinel words that verify a self-registered array must lie where the kernel stores)
    size_t out_first[3] = {0, 0, 0}, out_last[3] = {0, 0, 0};
    long const_fills = 0, const_refills = 0;
    bool time_host = false;       // bracket host-path launches with events (hipnlp_set_host_timing)
    double host_us[4] = {0, 0, 0, 0};   // wall clock of the last host-path evaluation: x staging, enqueue, wait for the GPU, copies out
    std::vector<double> p;
    bool params_set = false, have_result = false;
    // exact Hessian (allocated on first use)
    HessLayout HL;
    int hess_state = 0;   // 0: not built yet, 1: ready, -1: not available (HL.error)
    bool hess_layout_built = false, hess_layout_ok = false;
    HessTables* d_ht = nullptr;
    double *d_sigma = nullptr, *d_lambda = nullptr /* inside the d_sigma block */, *d_hess = nullptr, *h_hess = nullptr, *h_sl = nullptr;
    int32_t *d_hflag = nullptr, *h_hflag = nullptr, *hd_hflag = nullptr /* device-visible address of h_hflag */;
    double* d_htrash = nullptr;   // [nk * batch] DIRECT Hessian launches: one word per workgroup (HArgs::trash)
    double* hd_hess = nullptr;   // device-visible address of h_hess
    int32_t hseq = 0;   // Hessian launches so far (generation of d_hflag)
    std::string err;
};

#define HIP_TRY(h, call)                                                                            \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                           \
            (void)hipGetLastError(); /* (reported here: not left behind for the next launch's check) */ \
            return HIPNLP_E_NODEVICE;                                                               \
        }                                                                                           \
    } while (0)

#define NOT_FRONT(h, who)                                                                                                       \
    do {                                                                                                                      \
        if ((h)->is_front) {                                                                                                  \
            (h)->err = who ": a multi-device handle (hipnlp_multi_create) serves the host-buffer calls only";                \
            return HIPNLP_E_UNSUPPORTED;                                                                                      \
        }                                                                                                                     \
    } while (0)

static void free_all(hipnlp_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->dev);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void* dptrs[] = {h->d_tb, h->d_ctpl, h->d_ctpl_of_b, h->d_healed, h->d_x, h->d_pk, h->d_cost_knot, h->d_gp, h->d_cost_pub, h->d_out, h->d_ht, h->d_sigma, h->d_hess, h->d_hflag, h->d_htrash};
    for (void* q : dptrs) if (q) (void)hipFree(q);
    void* hptrs[] = {h->h_x, h->h_out, h->h_hess, h->h_hflag, h->h_sl};
    for (void* q : hptrs) if (q) (void)hipHostFree(q);
    for (hipEvent_t e : h->prof_ev) (void)hipEventDestroy(e);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" {

hipError_t hipnlp_internal_memcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
static void auto_unregister_all(hipnlp_handle* h);
static void constants_prepare(hipnlp_handle* h, const std::vector<GParams>& gp);
static void constants_fill(const hipnlp_handle* h, double* jac);
static void constants_ensure(hipnlp_handle* h, double* jac_host);
static int drop_stale_range(hipnlp_handle* h, void* p);
static bool auto_owns(const hipnlp_handle* h, const void* p);
static double* caller_array_address(hipnlp_handle* h, int q, double* p, size_t bytes);
static int multi_set_params(hipnlp_handle* h, const double* p);

const char* hipnlp_last_error(const hipnlp_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int hipnlp_abi_version(void) { return HIPNLP_ABI_VERSION; }
#ifndef HIPNLP_BUILD_VARIANT
#define HIPNLP_BUILD_VARIANT "plain hipcc"
#endif
const char* hipnlp_build_info(void) { return "gfx950; " HIPNLP_BUILD_VARIANT; }

// the calling THREAD onto the CPUs of the card's node (those of them it may run on); nothing changes when the node is unknown or none of
// its CPUs is allowed
int hipnlp_pin_thread_to_device_numa_node(int device, int* node_out, int* cpus_out) {
    int node = -1;
    if (node_out) *node_out = -1;
    if (cpus_out) *cpus_out = 0;
    const int rc = hipnlp_device_numa_node(device, &node);
    if (rc != HIPNLP_OK) return rc;
    if (node < 0) return HIPNLP_OK;
    FILE* f = std::fopen(("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist").c_str(), "r");
    if (!f) return HIPNLP_OK;
    char text[4096] = {0};
    const size_t got = std::fread(text, 1, sizeof text - 1, f);
    std::fclose(f);
    text[got] = 0;
    cpu_set_t allowed, want;
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return HIPNLP_OK;
    int count = 0;
    for (const char* c = text; *c;) {   // "0-63,128-191"
        if (!std::isdigit(static_cast<unsigned char>(*c))) { ++c; continue; }
        char* end = nullptr;
        long lo = std::strtol(c, &end, 10), hi = lo;
        if (*end == '-') hi = std::strtol(end + 1, &end, 10);
        for (long cpu = lo; cpu <= hi && cpu < CPU_SETSIZE; ++cpu)
            if (CPU_ISSET(int(cpu), &allowed)) { CPU_SET(int(cpu), &want); ++count; }
        c = end;
    }
    if (count == 0 || sched_setaffinity(0, sizeof want, &want) != 0) return HIPNLP_OK;
    if (node_out) *node_out = node;
    if (cpus_out) *cpus_out = count;
    return HIPNLP_OK;
}

// NUMA node of the host the card hangs off (Linux sysfs of its PCI function); -1: not known
int hipnlp_device_numa_node(int device, int* node) {
    if (!node) return HIPNLP_E_INVALID;
    *node = -1;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, int(sizeof bus), device) != hipSuccess) { (void)hipGetLastError(); return HIPNLP_E_NODEVICE; }
    for (char* c = bus; *c; ++c) *c = char(std::tolower(static_cast<unsigned char>(*c)));
    const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
    if (FILE* f = std::fopen(path.c_str(), "r")) {
        int n = -1;
        if (std::fscanf(f, "%d", &n) == 1) *node = n;
        std::fclose(f);
    }
    return HIPNLP_OK;
}

static void dims_fill(const hipnlp_handle* h, hipnlp_dims* o);
// front: the host side of a handle only (hipnlp_multi_create) — no stream, no device memory, the pinned blocks allocated so that every
// device of the process may read and write them
static int create_handle(const hipnlp_desc* desc, hipnlp_handle** out, bool front) {
    if (!desc || !out) { g_create_error = "null argument"; return HIPNLP_E_INVALID; }
    *out = nullptr;
    hipnlp_handle* h = new (std::nothrow) hipnlp_handle();
    if (!h) { g_create_error = "out of memory"; return HIPNLP_E_ALLOC; }
    h->d = *desc;
    h->is_front = front;
    const hipnlp_settings& st = desc->settings;
    auto fail = [&](int code, const std::string& msg) { g_create_error = msg; free_all(h); return code; };
    if (desc->abi_version != HIPNLP_ABI_VERSION)
        return fail(HIPNLP_E_INVALID, "hipnlp_desc.abi_version is " + std::to_string(desc->abi_version) + ", this library implements HIPNLP_ABI_VERSION " +
                                          std::to_string(HIPNLP_ABI_VERSION) + " (the caller was built against another include/hipnlp.h, or left the field unset)");
    if (desc->flags & ~(HIPNLP_FLAG_DETECT_SIMPLE_BOUNDS | HIPNLP_FLAG_JAC_VARYING_FIRST)) return fail(HIPNLP_E_INVALID, "unknown bits in hipnlp_desc.flags");
    if (st.horizon < 2) return fail(HIPNLP_E_INVALID, "settings.horizon must be >= 2");
    if (const char* te = Layout::check_terrain(st.terrain, st.n_terrain_steps, st.terrain_steps)) return fail(HIPNLP_E_INVALID, te);
    if (desc->batch < 1) return fail(HIPNLP_E_INVALID, "batch must be >= 1");
    for (int f = 0; f < 2; ++f) for (int i = 0; i < 3; ++i)
        if (st.yaw_corner[f][i] < 0 || st.yaw_corner[f][i] > 3) return fail(HIPNLP_E_INVALID, "yaw_corner indices must be in 0..3");
    h->batch = desc->batch;
    h->kb = desc->knot_begin; h->ke = desc->knot_end;
    if (h->kb == 0 && h->ke == 0) h->ke = st.horizon;
    if (h->kb < 0 || h->ke > st.horizon || h->kb >= h->ke) return fail(HIPNLP_E_INVALID, "bad knot shard [knot_begin, knot_end)");
    h->nk = h->ke - h->kb;
    {
        const char* force = diag_env("HIPNLP_WAVES");   // diagnostic override: 4 or 8
        // eight waves per knot while every workgroup of the launch (+ one reducer per trajectory) is resident at once: two per CU
        // (<= 128 VGPRs, 2 x 52 KB of LDS).  Measured at N = 100 (tools/diag/wide_sweep.sh): +28 .. +31 % over the four-wave kernel for
        // batch 2 .. 5, nothing once the launch exceeds the 512 slots (batch 6: 36.9 against 38.0 M knots/s).
        const bool fits = h->nk <= 256 && (long(h->nk) + 1) * long(desc->batch) <= 512;
        h->wide = force ? (std::atoi(force) == 8 && h->nk <= 256) : fits;
        if (const char* s0 = diag_env("HIPNLP_DEBUG_SEQ0")) h->seq = h->hseq = std::atoi(s0);   // diagnostic: launch numbers start here (tests of the wrap)
        const char* sep = diag_env("HIPNLP_SEPARATE_REDUCE");   // diagnostic override: the reduction kernel behind the four-wave kernel
        // (measured, four-wave kernel, N = 100: + 2.4 % at x 64 and + 6.5 % on the stairs 200 x 16, whose second launch is 3 of 53 us;
        //  - 1.1 % at x 1024, where ten rows are in flight and as many reducers spin in workgroup slots: long launches keep the kernel)
        const bool small_launch = long(h->nk) * long(desc->batch) <= 32768;
        // (the reducer workgroup stages [nk][16] doubles in its kernel's scratch: 256 knots, fewer on the trimmed scratch of the four-wave
        //  VARY kernels a handle launches for destinations that hold the constant entries)
        int red_cap = 256;
        if (!h->wide)   // (either order of a block: device destinations get the VARY kernels)
            red_cap = st.terrain == HIPNLP_TERRAIN_PLANAR ? reducer_cap<KnotScratchT<LAYOUT_COMPACT, js::vary_slots(true)>>()
                                                          : reducer_cap<KnotScratchT<LAYOUT_COMPACT, js::vary_slots(false)>>();
        h->fused = h->nk <= red_cap && (h->wide || (small_launch && !(sep && std::atoi(sep) == 1)));
        long split_max = 256;   // workgroups of a split launch: one per CU
        if (const char* sm = diag_env("HIPNLP_SPLIT_MAX")) split_max = std::atol(sm);   // diagnostic: where the regime ends
        h->split = h->wide && h->fused && (2 * long(h->nk) + 1) * long(desc->batch) <= split_max &&
                   st.final_state_type != HIPNLP_EXPR_MINIMIZE && st.periodicity_type != HIPNLP_EXPR_MINIMIZE;
        if (const char* sp = diag_env("HIPNLP_SPLIT")) h->split = h->split && std::atoi(sp) != 0;   // diagnostic override (A/B in one process)
        const char* hl = diag_env("HIPNLP_HESS_LAYOUT");   // diagnostic override: full | compact
        h->hess_compact = hl ? std::strcmp(hl, "compact") == 0 : long(h->nk) * long(desc->batch) > 512;
        if (const char* hd = diag_env("HIPNLP_HESS_DIRECT")) h->hess_direct = std::atoi(hd) != 0;
    }
    std::string e;
    if (!Layout::make_kin_tables(desc->model, h->kt, e)) return fail(HIPNLP_E_INVALID, e);
    Layout::fill_terrain_tops(h->kt, st.terrain, st.n_terrain_steps, st.terrain_steps);
    if (!h->L.build(st, h->kt, (desc->flags & HIPNLP_FLAG_DETECT_SIMPLE_BOUNDS) != 0, (desc->flags & HIPNLP_FLAG_JAC_VARYING_FIRST) != 0))
        return fail(HIPNLP_E_INVALID, h->L.error);
    h->skip_const = h->L.vary_first;   // (a CCS handle stores every entry unless hipnlp_set_constant_jacobian(h, 1) asks for the scheme: the contract on the destination is the caller's to accept)
    if (const char* cj = diag_env("HIPNLP_CONST_JAC")) h->skip_const = std::atoi(cj) != 0;   // diagnostic override of hipnlp_set_constant_jacobian's default
    if (const char* vc = diag_env("HIPNLP_VARY_CHECK")) h->vary_check = std::atoi(vc) != 0;
    if (const char* es = diag_env("HIPNLP_EARLY_STORE")) { h->early_store = std::atoi(es) != 0; h->early_store_always = std::atoi(es) == 2; }   // diagnostic override (A/B in one process)
    if (const char* he = diag_env("HIPNLP_HESS_EARLY_RUN")) h->hess_early_mode = std::atoi(he);   // diagnostic: -1 / 0 / 1, as hipnlp_set_hessian_early_run
    h->np = ParamOffsets(st.horizon).np();
    if (h->L.jperm_glob.size() > 16) return fail(HIPNLP_E_INVALID, "internal: too many global-column entries");

    // ---- device -------------------------------------------------------------------------------------
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(HIPNLP_E_NODEVICE, "no HIP device available (the engine has no CPU fallback)");
    if (desc->device < 0 || desc->device >= ndev) return fail(HIPNLP_E_INVALID, "bad device ordinal");
    h->dev = desc->device;
#define CREATE_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(HIPNLP_E_NODEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
#define DEV_TRY(call) do { if (!front) CREATE_TRY(call); } while (0)   /* (a front handle owns no stream and no device memory) */
    // pinned blocks of a front handle: read and written by the kernels of EVERY shard's device
    const unsigned pin_flags = front ? (hipHostMallocPortable | hipHostMallocMapped) : hipHostMallocDefault;
    CREATE_TRY(hipSetDevice(h->dev));
    DEV_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    DEV_TRY(hipEventCreate(&h->ev0));
    DEV_TRY(hipEventCreate(&h->ev1));
    const size_t B = size_t(h->batch), n = size_t(h->L.n), m = size_t(h->L.m), nnz = size_t(h->L.nnz), N = size_t(st.horizon);
    DEV_TRY(hipMalloc(&h->d_tb, sizeof(DeviceTables)));
    DEV_TRY(hipMalloc(&h->d_x, B * n * sizeof(double)));
    DEV_TRY(hipMalloc(&h->d_pk, B * N * PK_STRIDE * sizeof(double)));
    DEV_TRY(hipMalloc(&h->d_gp, B * sizeof(GParams)));
    // every output of the host-buffer path in ONE device block and ONE pinned mirror: [f | cost terms | grad | g | jac | flag],
    // copied back with a single asynchronous copy (six separate copies cost ~8 us each in launch overhead alone)
    const size_t out_doubles = B * (1 + NCT + n + m + nnz);
    h->out_bytes = out_doubles * sizeof(double) + ((B * sizeof(int32_t) + 7) / 8) * 8;
    DEV_TRY(hipMalloc(&h->d_out, h->out_bytes));
    CREATE_TRY(hipHostMalloc(&h->h_out, h->out_bytes, pin_flags));
    auto carve = [&](char* base) {
        double* p = reinterpret_cast<double*>(base);
        double* f_ = p; p += B;
        double* ct_ = p; p += B * NCT;
        double* gr_ = p; p += B * n;
        double* g_ = p; p += B * m;
        double* j_ = p; p += B * nnz;
        return std::make_tuple(f_, ct_, gr_, g_, j_, reinterpret_cast<int32_t*>(p));
    };
    if (!front) std::tie(h->d_f, h->d_cost_terms, h->d_grad, h->d_g, h->d_jac, h->d_flag) = carve(static_cast<char*>(h->d_out));
    std::tie(h->h_f, h->h_cost_terms, h->h_grad, h->h_g, h->h_jac, h->h_flag) = carve(static_cast<char*>(h->h_out));
    {
        void* od = nullptr;
        CREATE_TRY(hipHostGetDevicePointer(&od, h->h_out, 0));
        std::tie(h->hd_f, h->hd_cost_terms, h->hd_grad, h->hd_g, h->hd_jac, h->hd_flag) = carve(static_cast<char*>(od));
        std::memset(h->h_out, 0, h->out_bytes);
    }
    DEV_TRY(hipMalloc(&h->d_cost_knot, B * size_t(h->nk) * NCT * sizeof(double)));
    // (twice: the SPLIT launches publish the model-free workgroups' partials behind those of the kinematic ones)
    DEV_TRY(hipMalloc(&h->d_cost_pub, 2 * B * size_t(h->nk) * NCT * 2 * sizeof(unsigned long long)));
    DEV_TRY(hipMemset(h->d_cost_pub, 0, 2 * B * size_t(h->nk) * NCT * 2 * sizeof(unsigned long long)));
    DEV_TRY(hipMemset(h->d_flag, 0, B * sizeof(int32_t)));
    // (measured: non-coherent pinned memory for this block — cacheable in the GPU's L2 within a launch, so that the halo record would
    //  not cross PCIe twice — changes nothing: 22.3 us of GPU wait per objective call either way)
    CREATE_TRY(hipHostMalloc(&h->h_x, B * n * sizeof(double), pin_flags));
    {
        void* xd = nullptr;
        CREATE_TRY(hipHostGetDevicePointer(&xd, h->h_x, 0));
        h->hd_x = static_cast<double*>(xd);
        // x over PCIe by the kernel itself (the halo record is read twice): wins while x is small (151 KB at 100 knots: ~5 us
        // less than the copy command, profiles/r02_pcie_probe.txt); big batches go through one H2D copy and the L2
        h->x_zero_copy = B * n * sizeof(double) <= (size_t(1) << 20);
        // (the multipliers of hipnlp_eval_hess likewise: 220 KB at 100 knots.  The knot workgroups gather them through the slot -> row map —
        //  runs of 3 - 24 doubles per constraint block — and even as such fragments on the link they arrive sooner than behind a copy
        //  command and its hand-over to the kernel: 67.5 -> 59.7 us per 100-knot Hessian through host buffers, tools/diag/hess_host_ab.py)
        h->lam_zero_copy = B * (1 + m) * sizeof(double) <= (size_t(1) << 20);
        if (const char* lz = diag_env("HIPNLP_HESS_LAM_ZERO_COPY")) h->lam_zero_copy = std::atoi(lz) != 0;   // diagnostic override (A/B in one process)
    }
    DEV_TRY(hipMemset(h->d_g, 0, B * m * sizeof(double)));
    DEV_TRY(hipMemset(h->d_jac, 0, B * nnz * sizeof(double)));
    DEV_TRY(hipMemset(h->d_grad, 0, B * n * sizeof(double)));
    // tables
    DeviceTables* tb = new DeviceTables();
    std::memset(tb, 0, sizeof(DeviceTables));
    tb->head.kt = h->kt;
    tb->head.ks = Layout::make_ksettings(st);
    for (int v = 0; v < 3; ++v) {
        for (int s = 0; s < GS_PAD; ++s) tb->g_a[v][s] = s < gs::COUNT ? h->L.g_a[v][size_t(s)] : G_NONE;
        for (int i = 0; i < JS_PAD; ++i) tb->jperm[v][i] = size_t(i) < h->L.jperm[v].size() ? h->L.jperm[v][size_t(i)] : -1;
        tb->nnz_v[v] = h->L.nnz_v[v];
    }
    for (int s = 0; s < gs::COUNT; ++s) tb->g_b[s] = h->L.g_b[size_t(s)];
    {
        const int wg = h->wide ? 512 : 256, jt = JS_PAD / wg, gt = GS_PAD / wg;
        // position of a slot among the slots a knot of variant v owns (slot order)
        std::vector<int32_t> gpos[3];
        for (int v = 0; v < 3; ++v) {
            gpos[v].assign(GS_PAD, 0);
            int at = 0;
            for (int sl = 0; sl < gs::COUNT; ++sl) if (tb->g_a[v][sl] != G_NONE) gpos[v][size_t(sl)] = at++;
            h->gs_rows_v[v] = at;
            for (int sl = 0; sl < gs::COUNT; ++sl) if (tb->g_b[sl] & ~GB_STRIDE) h->gs_compact_ok = false;   // (a row stride beyond the field: never)
        }
        for (int v = 0; v < 3; ++v)
            for (int t = 0; t < wg; ++t) {
                for (int it = 0; it < jt; ++it) {
                    const int32_t slot = tb->jperm[v][t + it * wg];
                    tb->jperm_t[v][t * jt + it] = slot >= 0 && h->L.jslot_phase[size_t(slot)] <= 1 ? (slot | COPY_EARLY) : slot;
                }
                for (int it = 0; it < gt; ++it) {
                    const int gslot = t + it * wg;
                    tb->gab_t[v][2 * (t * gt + it)] = tb->g_a[v][gslot];
                    tb->gab_t[v][2 * (t * gt + it) + 1] = tb->g_b[gslot] | (gpos[v][gslot] << GB_POS_SHIFT);   // (the rows of g stay with the copy-out at the end: their runs are 24 - 192 B, and written through the L2 one by one they cost more link time than the early start saves — measured: g alone 11.9 -> 21 us)
                }
            }
    }
    tb->n_glob = int(h->L.jperm_glob.size());
    for (int i = 0; i < tb->n_glob; ++i) tb->jperm_glob[i] = h->L.jperm_glob[size_t(i)];
    tb->jac_glob_base = h->L.jac_glob_base;
    {
        // the VARY instantiations (varying-first order): the varying run of a block, thread-major, -1 behind it
        const int wg = h->wide ? 512 : 256, cap = vary_cap(st.terrain, wg), jt = cap / wg;
        h->vary_ok = h->L.vary_partition_ok;   // (a varying slot outside the window the trimmed scratch stages: every entry is stored, by the kernels on the full staging)
        for (int v = 0; v < 3; ++v) {
            if (h->L.nnz_v[v] > JP_POS + 1) h->vary_ok = false;
            h->cpos[v].clear();
            for (size_t i = 0; i < h->L.jconst_pos[v].size(); ++i) if (h->L.jconst_pos[v][i]) h->cpos[v].push_back(int32_t(i));
            tb->nvary_v[v] = h->L.nvary_v[v];
            for (int i = 0; i < VCAP_MAX; ++i) tb->jpermv_t[v][i] = -1;
            if (h->L.nvary_v[v] > cap) h->vary_ok = false;   // (a settings combination with more varying entries than the kernels' trip count: every entry is stored)
        }
        if (h->vary_ok)
            for (int v = 0; v < 3; ++v)
                for (int pos = 0, i = 0; pos < h->L.nnz_v[v]; ++pos) {   // i: index among the block's varying entries, in the order of the block
                    if (h->L.jconst_pos[v][size_t(pos)]) continue;
                    const int32_t slot = h->L.jperm[v][size_t(pos)];
                    tb->jpermv_t[v][(i % wg) * jt + i / wg] = slot | (pos << JP_POS_SHIFT) | (h->L.jslot_phase[size_t(slot)] <= 1 ? COPY_EARLY : 0);
                    ++i;
                }
    }
    {
        // SPLIT instantiations: the thread-major tables once per half of the program, the other half's entries blanked (recorded owners:
        // Layout::jslot_owner / gslot_owner)
        const int wg = h->wide ? 512 : 256, jt = JS_PAD / wg, gt = GS_PAD / wg;
        for (int hw = 0; hw < 2; ++hw) {
            for (int v = 0; v < 3; ++v) {
                for (int i = 0; i < wg * jt; ++i) {
                    const int32_t w = tb->jperm_t[v][i];
                    tb->split_jperm_t[hw][v][i] = (w >= 0 && h->L.jslot_owner[size_t(w & COPY_SLOT & JP_SLOT)] == hw) ? w : -1;
                }
                for (int i = wg * jt; i < JS_PAD; ++i) tb->split_jperm_t[hw][v][i] = -1;
                for (int t = 0; t < wg; ++t)
                    for (int it = 0; it < gt; ++it) {
                        const int gslot = t + it * wg, at = 2 * (t * gt + it);
                        const bool mine = gslot < gs::COUNT && h->L.gslot_owner[size_t(gslot)] == hw;
                        tb->split_gab_t[hw][v][at] = mine ? tb->gab_t[v][at] : G_NONE;
                        tb->split_gab_t[hw][v][at + 1] = tb->gab_t[v][at + 1];
                    }
                for (int i = 0; i < VCAP_MAX; ++i) {
                    const int32_t w = tb->jpermv_t[v][i];
                    tb->split_jpermv_t[hw][v][i] = (w >= 0 && h->L.jslot_owner[size_t(jp_slot(w))] == hw) ? w : -1;
                }
            }
            for (int i = 0; i < 16; ++i)
                tb->split_jperm_glob[hw][i] = (i < tb->n_glob && h->L.jslot_owner[size_t(tb->jperm_glob[i])] == hw) ? tb->jperm_glob[i] : -1;
        }
    }
    hipError_t ce = front ? hipSuccess : hipnlp_internal_memcpy(h->d_tb, tb, sizeof(DeviceTables), hipMemcpyHostToDevice);
    if (!front && ce == hipSuccess) ce = hipMalloc(&h->d_healed, sizeof(int32_t));
    if (!front && ce == hipSuccess) ce = hipMemset(h->d_healed, 0, sizeof(int32_t));
    delete tb;
    if (ce != hipSuccess) return fail(HIPNLP_E_NODEVICE, std::string("hipMemcpy tables: ") + hipGetErrorString(ce));
    {
        // first / last entry of [batch][nnz] that depends on x (where a store that skips the constants leaves its sentinel words)
        const Layout& L = h->L;
        h->jac_first_vary = h->jac_last_vary = 0;
        bool found = false;
        for (int k = h->kb; k < h->ke && !found; ++k) {
            const int v = L.variant_of(k);
            for (size_t i = 0; i < L.jconst_pos[v].size(); ++i) if (!L.jconst_pos[v][i]) { h->jac_first_vary = size_t(L.jac_base(k)) + i; found = true; break; }
        }
        found = false;
        for (int k = h->ke - 1; k >= h->kb && !found; --k) {
            const int v = L.variant_of(k);
            for (size_t i = L.jconst_pos[v].size(); i-- > 0;) if (!L.jconst_pos[v][i]) { h->jac_last_vary = (B - 1) * nnz + size_t(L.jac_base(k)) + i; found = true; break; }
        }
    }
#undef DEV_TRY
#undef CREATE_TRY
    dims_fill(h, &h->dims);
    {
        const Layout& L = h->L;
        const size_t Bm1 = size_t(h->batch) - 1;
        h->out_first[0] = size_t(h->dims.shard_grad_off);
        h->out_last[0] = Bm1 * size_t(L.n) + size_t(h->dims.shard_grad_off) + size_t(h->dims.shard_grad) - 1;
        long rmin = LONG_MAX, rmax = -1;
        for (int k = h->kb; k < h->ke; ++k) {
            const int v = L.variant_of(k);
            for (int sl = 0; sl < gs::COUNT; ++sl) {
                const int a = L.g_a[v][size_t(sl)];
                if (a == G_NONE) continue;
                const long r = long(a) + long(L.g_b[size_t(sl)]) * k;
                rmin = std::min(rmin, r); rmax = std::max(rmax, r);
            }
        }
        h->out_first[1] = size_t(rmax >= 0 ? rmin : 0);
        h->out_last[1] = Bm1 * size_t(L.m) + size_t(rmax >= 0 ? rmax : 0);
        h->out_first[2] = size_t(h->dims.shard_jac_off);
        h->out_last[2] = Bm1 * size_t(L.nnz) + size_t(h->dims.shard_jac_off) + size_t(h->dims.shard_nnz) - 1;
    }
    *out = h;
    return HIPNLP_OK;
}
int hipnlp_create(const hipnlp_desc* desc, hipnlp_handle** out) { return create_handle(desc, out, false); }

static void multi_destroy(hipnlp_handle* h);
void hipnlp_destroy(hipnlp_handle* h) {
    if (h) {
        (void)hipSetDevice(h->dev);
        auto_unregister_all(h);
        if (h->multi) multi_destroy(h);
    }
    free_all(h);
}

// (computed once per handle: IPOPT's callbacks check their sizes against it at every call, and the row count below is a pass over
//  every native slot of every knot — 3 us at 100 knots, four times per iterate until round 4)
static void dims_fill(const hipnlp_handle* h, hipnlp_dims* o);
int hipnlp_get_dims(const hipnlp_handle* h, hipnlp_dims* o) {
    if (!h || !o) return HIPNLP_E_INVALID;
    *o = h->dims;
    return HIPNLP_OK;
}
static void dims_fill(const hipnlp_handle* h, hipnlp_dims* o) {
    const Layout& L = h->L;
    o->n = L.n; o->m = L.m; o->nnz = L.nnz; o->np = h->np;
    o->nnz_knot = L.N >= 3 ? L.nnz_v[VAR_INTERIOR] : 0;
    int mk = 0;
    for (const RowBlock& b : L.blocks) if (b.nk > 1) mk += b.rows;
    o->m_knot = mk;
    o->shard_grad = NXK * h->nk + (h->ke == L.N ? NXG : 0);
    o->shard_grad_off = NXK * h->kb;
    o->shard_jac_off = int(L.jac_base(h->kb));
    o->shard_nnz = int((h->ke == L.N ? long(L.nnz) : L.jac_base(h->ke)) - L.jac_base(h->kb));
    int rows = 0;
    for (int k = h->kb; k < h->ke; ++k) {
        const int v = L.variant_of(k);
        for (int s = 0; s < gs::COUNT; ++s) rows += L.g_a[v][size_t(s)] != G_NONE;
    }
    o->shard_g_rows = rows;
    o->m_full = L.m_full;
    o->n_lifted = L.n_lifted;
}

// ---- the constant entries of jac g ---------------------------------------------------------------------------------------------------
// Their values under the parameters just set: one pass of the knot program on the host per DISTINCT GParamsLite of the batch (dt and
// the mass are what the constants hold; trajectories of a batch usually share them), kept as templates by block variant.
static void constants_prepare(hipnlp_handle* h, const std::vector<GParams>& gp) {
    const Layout& L = h->L;
    h->param_gen++;
    h->ctpl.clear();
    h->ctpl_of_b.assign(size_t(h->batch), 0);
    h->ctpl_off[0] = 0;
    for (int v = 0; v < 3; ++v) h->ctpl_off[v + 1] = h->ctpl_off[v] + L.nnz_v[v];
    std::vector<size_t> rep;   // trajectory each template was computed from
    std::vector<double> cval(js::COUNT);
    for (size_t b = 0; b < size_t(h->batch); ++b) {
        int found = -1;
        for (size_t t = 0; t < rep.size() && found < 0; ++t)
            if (std::memcmp(static_cast<const GParamsLite*>(&gp[rep[t]]), static_cast<const GParamsLite*>(&gp[b]), sizeof(GParamsLite)) == 0) found = int(t);
        if (found < 0) {
            Layout::constant_values(h->d.settings, h->kt, gp[b], cval.data());
            double vary_mark;
            { const unsigned long long w = CTPL_VARYING; std::memcpy(&vary_mark, &w, sizeof w); }
            std::vector<double> t(size_t(h->ctpl_off[3]) + L.jperm_glob.size(), vary_mark);
            for (int v = 0; v < 3; ++v)
                for (int32_t i : h->cpos[v]) t[size_t(h->ctpl_off[v] + i)] = cval[size_t(L.jperm[v][size_t(i)])];
            for (size_t i = 0; i < L.jperm_glob.size(); ++i) t[size_t(h->ctpl_off[3]) + i] = cval[size_t(L.jperm_glob[i])];
            found = int(h->ctpl.size());
            h->ctpl.push_back(std::move(t));
            rep.push_back(b);
        }
        h->ctpl_of_b[b] = found;
    }
    // the spot check of a caller array: the first and the last constant entry of the handle's knots and a few in between
    h->csample.clear();
    if (L.nconst_total > 0) {
        const size_t nnz = size_t(L.nnz);
        auto add = [&](size_t b, int k, size_t j) {
            const int v = L.variant_of(k);
            if (h->cpos[v].empty()) return;
            const int32_t i = h->cpos[v][j % h->cpos[v].size()];
            h->csample.push_back({b * nnz + size_t(L.jac_base(k)) + size_t(i), h->ctpl[size_t(h->ctpl_of_b[b])][size_t(h->ctpl_off[v] + i)]});
        };
        const size_t B = size_t(h->batch);
        add(0, h->kb, 0);
        for (int q = 1; q <= 14; ++q) add((B * size_t(q)) / 16, h->kb + int((long(h->nk) * q) / 16), size_t(37 * q));
        add(B - 1, h->ke - 1, h->cpos[L.variant_of(h->ke - 1)].empty() ? 0 : h->cpos[L.variant_of(h->ke - 1)].size() - 1);
    }
    // (the pinned block, the library's own, is filled when a launch that skips the constants first stores into it: host_evaluate)
    // device copy for the VARY kernels' check and for the fill of device destinations (hipnlp_eval_device)
    h->ctpl_len = h->ctpl.empty() ? 0 : int(h->ctpl[0].size());
    if (h->vary_ok && h->ctpl_len > 0 && !h->is_front) {   // (a front handle fills HOST destinations only: its shards hold the device copies)
        if (h->d_ctpl) { (void)hipFree(h->d_ctpl); h->d_ctpl = nullptr; }
        if (!h->d_ctpl_of_b && hipMalloc(&h->d_ctpl_of_b, size_t(h->batch) * sizeof(int32_t)) != hipSuccess) { h->d_ctpl_of_b = nullptr; h->vary_ok = false; }
        if (h->vary_ok && hipMalloc(&h->d_ctpl, h->ctpl.size() * size_t(h->ctpl_len) * sizeof(double)) != hipSuccess) { h->d_ctpl = nullptr; h->vary_ok = false; }
        if (h->vary_ok) {
            for (size_t t = 0; t < h->ctpl.size(); ++t)
                (void)hipnlp_internal_memcpy(h->d_ctpl + t * size_t(h->ctpl_len), h->ctpl[t].data(), size_t(h->ctpl_len) * sizeof(double), hipMemcpyHostToDevice);
            std::vector<int32_t> of(h->ctpl_of_b.begin(), h->ctpl_of_b.end());
            (void)hipnlp_internal_memcpy(h->d_ctpl_of_b, of.data(), of.size() * sizeof(int32_t), hipMemcpyHostToDevice);
        }
    }
}
// the constant entries of this handle's knots into a host array jac [batch][nnz]
static void constants_fill(const hipnlp_handle* h, double* jac) {
    const Layout& L = h->L;
    const size_t nnz = size_t(L.nnz);
    for (size_t b = 0; b < size_t(h->batch); ++b) {
        const std::vector<double>& t = h->ctpl[size_t(h->ctpl_of_b[b])];
        double* out = jac + b * nnz;
        for (int k = h->kb; k < h->ke; ++k) {
            const int v = L.variant_of(k);
            double* blk = out + L.jac_base(k);
            const double* tv = t.data() + h->ctpl_off[v];
            if (L.vary_first) std::memcpy(blk + L.nvary_v[v], tv + L.nvary_v[v], size_t(L.nnz_v[v] - L.nvary_v[v]) * sizeof(double));   // (one run behind the varying entries)
            else for (int32_t i : h->cpos[v]) blk[i] = tv[i];
        }
        if (h->ke == L.N) for (size_t i = 0; i < L.jperm_glob.size(); ++i) out[size_t(L.jac_glob_base) + i] = t[size_t(h->ctpl_off[3]) + i];
    }
}
// A host destination of the Jacobian that is not the library's own, about to receive a store that skips the constants: they must be
// in place.  Known array of the current parameter set: spot check (a caller that wrote over its array gets it re-filled); anything
// else: filled now.
static void constants_ensure(hipnlp_handle* h, double* jac_host) {
    int slot = -1;
    for (int i = 0; i < 8; ++i) if (h->cfilled[i].host == jac_host) slot = i;
    const unsigned long long reg = registration_of(jac_host, size_t(h->batch) * size_t(h->L.nnz) * sizeof(double));
    bool ok = slot >= 0 && h->cfilled[slot].gen == h->param_gen && reg != 0 && h->cfilled[slot].reg == reg;
    if (ok)
        for (const auto& sv : h->csample)
            if (std::memcmp(&jac_host[sv.first], &sv.second, sizeof(double)) != 0) { ok = false; h->const_refills++; break; }
    if (ok) return;
    constants_fill(h, jac_host);
    h->const_fills++;
    if (slot < 0) { slot = h->cfilled_next; h->cfilled_next = (h->cfilled_next + 1) % 8; }
    h->cfilled[slot] = {jac_host, h->param_gen, reg};
}

int hipnlp_set_params(hipnlp_handle* h, const double* p) {
    if (!h || !p) return HIPNLP_E_INVALID;
    HIP_TRY(h, hipSetDevice(h->dev));
    const size_t B = size_t(h->batch), N = size_t(h->L.N);
    h->p.assign(p, p + B * size_t(h->np));
    std::vector<double> pk(B * N * PK_STRIDE);
    std::vector<GParams> gp(B);
    for (size_t b = 0; b < B; ++b) pack_params(p + b * size_t(h->np), int(N), pk.data() + b * N * PK_STRIDE, gp[b]);
    if (h->multi) {   // every shard takes the whole parameter array (its kernels index the knot records by global knot number)
        const int rc = multi_set_params(h, p);
        if (rc != HIPNLP_OK) return rc;
    } else {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        HIP_TRY(h, hipnlp_internal_memcpy(h->d_pk, pk.data(), pk.size() * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(h, hipnlp_internal_memcpy(h->d_gp, gp.data(), gp.size() * sizeof(GParams), hipMemcpyHostToDevice));
    }
    h->params_set = true;
    h->have_result = false;
    constants_prepare(h, gp);
    return HIPNLP_OK;
}

int hipnlp_bounds(const hipnlp_handle* h, double* lbx, double* ubx, double* lbg, double* ubg) {
    if (!h) return HIPNLP_E_INVALID;
    if (!h->params_set) return HIPNLP_E_PARAMS;
    h->L.bounds(h->p.data(), lbx, ubx, lbg, ubg);
    return HIPNLP_OK;
}

int hipnlp_lift_map(const hipnlp_handle* h, int32_t* kept_row, double* lb_full, double* ub_full) {
    if (!h) return HIPNLP_E_INVALID;
    if ((lb_full || ub_full) && !h->params_set) return HIPNLP_E_PARAMS;
    if (kept_row) h->L.kept_rows(kept_row);
    if (lb_full || ub_full) {
        std::vector<double> lo(size_t(h->L.m_full)), hi(size_t(h->L.m_full));
        h->L.bounds_full(h->p.data(), lo.data(), hi.data());
        if (lb_full) std::memcpy(lb_full, lo.data(), lo.size() * sizeof(double));
        if (ub_full) std::memcpy(ub_full, hi.data(), hi.size() * sizeof(double));
    }
    return HIPNLP_OK;
}

int hipnlp_simple_rows(const hipnlp_handle* h, int32_t* is_simple, int32_t* var_index) {
    if (!h || !is_simple || !var_index) return HIPNLP_E_INVALID;
    h->L.simple_rows(is_simple, var_index);
    return HIPNLP_OK;
}

int hipnlp_sparsity(const hipnlp_handle* h, int32_t* irow, int32_t* jcol) {
    if (!h || !irow || !jcol) return HIPNLP_E_INVALID;
    std::memcpy(irow, h->L.irow.data(), size_t(h->L.nnz) * sizeof(int32_t));
    std::memcpy(jcol, h->L.jcol.data(), size_t(h->L.nnz) * sizeof(int32_t));
    return HIPNLP_OK;
}

// Timing: an event record drains the stream around the kernel and costs several microseconds, more than a third of a
// 100-knot callback.  The host-buffer path (hipnlp_eval, PCIe bound anyway) is always timed; the device path is timed only for
// the launches an armed profile selects (every stride-th launch), so that measuring does not change what is measured.
// entries in the varying runs of the knots [0, k): where knot k's run starts in a compact destination
static int64_t vary_base(const Layout& L, int k) {
    if (k <= 0) return 0;
    int64_t at = L.nvary_v[VAR_FIRST] + int64_t(std::min(k, L.N - 1) - 1) * L.nvary_v[VAR_INTERIOR];
    if (k >= L.N) at += L.nvary_v[VAR_LAST];
    return at;
}
// rows of g owned by the knots [0, k): where knot k's rows start in a compact staging
static int64_t gs_base(const hipnlp_handle* h, int k) {
    const int N = h->L.N;
    if (k <= 0) return 0;
    int64_t at = h->gs_rows_v[VAR_FIRST] + int64_t(std::min(k, N - 1) - 1) * h->gs_rows_v[VAR_INTERIOR];
    if (k >= N) at += h->gs_rows_v[VAR_LAST];
    return at;
}
static int launch(hipnlp_handle* h, const double* x_dev, double* f_dev, double* grad_dev, double* g_dev, double* jac_dev, hipStream_t s,
                  double* g_stage = nullptr, bool shard_local = false, bool always_timed = false, bool host_block = false,
                  double* const* peer_out = nullptr, int npeer = 0, int peer_rank = 0, bool vary_only = false, bool compact = false, bool no_check = false,
                  double* cost_dst = nullptr, int cc_parity = -1) {
    // cc_parity: which pair of the four sample positions of a block the constants self-check of this launch looks at (0 / 1), counted per
    // DESTINATION by the caller; -1: by the handle's launch number (callers that alternate two jac buffers with the same parity would
    // otherwise show every buffer one pair only)
    // cost_dst (shard launches of a multi-device handle): [batch][nk][NCT], device-visible HOST memory — the knot workgroups store their
    // cost partials there and nothing on the device sums them (no reducer workgroup, no reduction kernel: f_dev is not written); the one
    // caller sums the partials of all shards in the order of the device reduction (multi_reduce_costs)
    // vary_only: the constant entries of jac g are already at jac_dev (constants_ensure): the copy-out leaves them alone
    // compact (with vary_only): jac_dev is a COMPACT destination — the varying runs of the knot blocks behind one another, no room for constants
    // no_check: the destination is not local device memory (a registered host range, a peer's buffer): the launch does not read it back
    KArgs a;
    a.peer_out = peer_out; a.npeer = npeer; a.peer_rank = peer_rank;
    a.tb = h->d_tb; a.x = x_dev; a.pk = h->d_pk; a.gp = h->d_gp;
    {   // the check of the constants a VARY launch finds in a DEVICE destination (a host destination was spot-checked by the host: no reads over PCIe)
        ConstCheck& c = a.cc;
        const bool on = vary_only && !host_block && !compact && !no_check && !peer_out && h->vary_check && jac_dev && h->d_ctpl;
        c.jac = on ? jac_dev : nullptr; c.ctpl = on ? h->d_ctpl : nullptr; c.ctpl_of_b = h->d_ctpl_of_b; c.healed = h->d_healed;
        c.jac_stride = h->L.nnz; c.jac_off = 0; c.ctpl_len = h->ctpl_len;
        for (int v = 0; v < 4; ++v) c.ctpl_off[v] = h->ctpl_off[v];
        for (int v = 0; v < 3; ++v) {
            c.nnz_v[v] = h->L.nnz_v[v]; c.first_const[v] = h->cpos[v].empty() ? -1 : h->cpos[v][0];
            const size_t nc = h->cpos[v].size();
            const size_t at[4] = {0, nc ? nc - 1 : 0, nc ? (nc - 1) / 3 : 0, nc ? (2 * (nc - 1)) / 3 : 0};   // {first, last, two in between}
            for (size_t q = 0; q < 4; ++q) c.samp[v][q] = nc ? h->cpos[v][at[q]] : -1;
        }
        c.kb = h->kb; c.nk = h->nk; c.N = h->L.N; c.n_glob = int(h->L.jperm_glob.size()); c.jac_glob_base = h->L.jac_glob_base;
    }
    a.g = g_dev; a.jac = jac_dev; a.grad = grad_dev; a.g_stage = g_stage;
    a.jb_first = compact ? h->L.nvary_v[VAR_FIRST] : h->L.nnz_v[VAR_FIRST];
    a.jb_interior = compact ? h->L.nvary_v[VAR_INTERIOR] : h->L.nnz_v[VAR_INTERIOR];
    a.gsb_first = a.gsb_interior = 0; a.gs_stride = a.gs_off = 0;
    if (compact && g_stage && h->gs_compact_ok) {   // the rows of g in a compact staging: as many per knot as the knot owns
        a.gsb_first = h->gs_rows_v[VAR_FIRST]; a.gsb_interior = h->gs_rows_v[VAR_INTERIOR];
        a.gs_off = gs_base(h, h->kb); a.gs_stride = gs_base(h, h->ke) - a.gs_off;
    }
    if (shard_local) {
        hipnlp_dims dd;
        hipnlp_get_dims(h, &dd);
        a.jac_stride = dd.shard_nnz; a.jac_off = dd.shard_jac_off; a.grad_stride = dd.shard_grad; a.grad_off = dd.shard_grad_off;
        if (compact) { a.jac_stride = vary_base(h->L, h->ke) - vary_base(h->L, h->kb); a.jac_off = vary_base(h->L, h->kb); }
    } else {
        a.jac_stride = compact ? vary_base(h->L, h->L.N) : int64_t(h->L.nnz); a.jac_off = 0; a.grad_stride = h->L.n; a.grad_off = 0;
    }
    const bool fused = h->fused && !cost_dst;
    a.cost_knot = cost_dst ? cost_dst : h->d_cost_knot; a.f = f_dev; a.cost_pub = fused ? h->d_cost_pub : nullptr; a.flag = h->d_flag;
    // host-buffer path: per-term costs (96 B per trajectory) and the non-finite flag go straight to the pinned block
    a.cost_terms = host_block ? h->hd_cost_terms : h->d_cost_terms;
    a.flag_host = host_block ? h->hd_flag : nullptr;
    // (only launches that store the varying RUN of every block: written through the L2, the early entries of a CCS-ordered block are isolated
    //  doubles on the link — measured 49 -> 258 us per 100-knot call)
    a.early = ((host_block && h->early_store) || h->early_store_always) && vary_only ? 1 : 0;
    if (h->seq == INT32_MAX) {
        // the launch number tags the cost partials and is the generation of the non-finite flags: before it would wrap (2^31 launches:
        // hours of back-to-back 100-knot callbacks) everything that carries one starts over
        HIP_TRY(h, hipStreamSynchronize(s));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        HIP_TRY(h, hipMemset(h->d_flag, 0, size_t(h->batch) * sizeof(int32_t)));
        std::memset(h->h_flag, 0, size_t(h->batch) * sizeof(int32_t));
        if (h->d_cost_pub) HIP_TRY(h, hipMemset(h->d_cost_pub, 0, 2 * size_t(h->batch) * size_t(h->nk) * NCT * 2 * sizeof(unsigned long long)));
        HIP_TRY(h, hipDeviceSynchronize());
        h->seq = 0;
        h->seq_result = -1;
        h->have_result = false;
    }
    a.seq = ++h->seq;
    a.cc.parity = cc_parity >= 0 ? (cc_parity & 1) : (a.seq & 1);
    a.N = h->L.N; a.n = h->L.n; a.m = h->L.m; a.nnz = h->L.nnz; a.knot_begin = h->kb; a.nk = h->nk;
    bool prof = false, run_first = false, run_last = false;
    if (h->prof_cap > 0 && h->prof_run > 0) {        // runs of consecutive launches: one event before the first, one after the last
        const int pos = h->prof_seen % h->prof_run;
        run_first = !h->prof_open && h->prof_n < h->prof_cap && pos == 0;
        run_last = (h->prof_open || run_first) && pos == h->prof_run - 1;
        h->prof_seen++;
    } else if (h->prof_cap > 0) { prof = h->prof_n < h->prof_cap && (h->prof_seen % h->prof_stride) == 0; h->prof_seen++; }
    const bool timed = prof || always_timed;
    hipEvent_t e0 = prof ? h->prof_ev[size_t(3 * h->prof_n)] : h->ev0;
    hipEvent_t e2 = prof ? h->prof_ev[size_t(3 * h->prof_n + 2)] : h->ev1;
    if (run_first) { HIP_TRY(h, hipEventRecord(h->prof_ev[size_t(3 * h->prof_n)], s)); h->prof_open = true; }
#ifdef HIPNLP_STAMPS
    if (!h->d_stamps) {   // (SPLIT launches: two workgroups per knot)
        HIP_TRY(h, hipMalloc(&h->d_stamps, 2 * size_t(h->nk) * size_t(h->batch) * 1024 * sizeof(unsigned long long)));
        HIP_TRY(h, hipMemset(h->d_stamps, 0, 2 * size_t(h->nk) * size_t(h->batch) * 1024 * sizeof(unsigned long long)));
    }
    a.stamps = h->d_stamps;
#endif
    if (timed) HIP_TRY(h, hipEventRecord(e0, s));
    // two workgroups per knot: device destinations of the whole-array kind only (no staging of g for an exchange, no peers, no early stores)
    const bool split = h->split && fused && !peer_out && !g_stage && !host_block && !a.early && !compact;
    const dim3 grid((split ? 2u : 1u) * unsigned(h->nk) + (fused ? 1u : 0u), unsigned(h->batch));   // (+ the row's cost reducer)
    const bool planar = h->d.settings.terrain == HIPNLP_TERRAIN_PLANAR;
    if (split) {
        if (vary_only) {
            if (planar) hipLaunchKernelGGL((hipnlp_knot_kernel<HIPNLP_TERRAIN_PLANAR, 8, false, true, false, true>), grid, dim3(512), 0, s, a.tb, a.x, a.pk, a.gp, a.N, a.n, a.knot_begin, a.nk, a);
            else hipLaunchKernelGGL((hipnlp_knot_kernel<HIPNLP_TERRAIN_SMOOTH_STEPS, 8, false, true, false, true>), grid, dim3(512), 0, s, a.tb, a.x, a.pk, a.gp, a.N, a.n, a.knot_begin, a.nk, a);
        } else {
            if (planar) hipLaunchKernelGGL((hipnlp_knot_kernel<HIPNLP_TERRAIN_PLANAR, 8, false, false, false, true>), grid, dim3(512), 0, s, a.tb, a.x, a.pk, a.gp, a.N, a.n, a.knot_begin, a.nk, a);
            else hipLaunchKernelGGL((hipnlp_knot_kernel<HIPNLP_TERRAIN_SMOOTH_STEPS, 8, false, false, false, true>), grid, dim3(512), 0, s, a.tb, a.x, a.pk, a.gp, a.N, a.n, a.knot_begin, a.nk, a);
        }
    } else
    if (peer_out && vary_only) {   // every rank's buffer holds the constant entries: the varying run of every block, at its place in the pattern, once per rank
        if (h->wide) {
            if (planar) hipLaunchKernelGGL((hipnlp_knot_kernel<HIPNLP_TERRAIN_PLANAR, 8, true, true>), grid, dim3(512), 0, s, a.tb, a.x, a.pk, a.gp, a.N, a.n, a.knot_begin, a.nk, a);
            else hipLaunchKernelGGL((hipnlp_knot_kernel<HIPNLP_TERRAIN_SMOOTH_STEPS, 8, true, true>), grid, dim3(512), 0, s, a.tb, a.x, a.pk, a.gp, a.N, a.n, a.knot_begin, a.nk, a);
        } else {
            if (planar) hipLaunchKernelGGL((hipnlp_knot_kernel<HIPNLP_TERRAIN_PLANAR, 4, true, true>), grid, dim3(256), 0, s, a.tb, a.x, a.pk, a.gp, a.N, a.n, a.knot_begin, a.nk, a);
            else hipLaunchKernelGGL((hipnlp_knot_kernel<HIPNLP_TERRAIN_SMOOTH_STEPS, 4, true, true>), grid, dim3(256), 0, s, a.tb, a.x, a.pk, a.gp, a.N, a.n, a.knot_begin, a.nk, a);
        }
    } else if (peer_out) {
        if (h->wide) {
            if (planar) hipLaunchKernelGGL((hipnlp_knot_kernel<HIPNLP_TERRAIN_PLANAR, 8, true>), grid, dim3(512), 0, s, a.tb, a.x, a.pk, a.gp, a.N, a.n, a.knot_begin, a.nk, a);
            else hipLaunchKernelGGL((hipnlp_knot_kernel<HIPNLP_TERRAIN_SMOOTH_STEPS, 8, true>), grid, dim3(512), 0, s, a.tb, a.x, a.pk, a.gp, a.N, a.n, a.knot_begin, a.nk, a);
        } else {
            if (planar) hipLaunchKernelGGL((hipnlp_knot_kernel<HIPNLP_TERRAIN_PLANAR, 4, true>), grid, dim3(256), 0, s, a.tb, a.x, a.pk, a.gp, a.N, a.n, a.knot_begin, a.nk, a);
            else hipLaunchKernelGGL((hipnlp_knot_kernel<HIPNLP_TERRAIN_SMOOTH_STEPS, 4, true>), grid, dim3(256), 0, s, a.tb, a.x, a.pk, a.gp, a.N, a.n, a.knot_begin, a.nk, a);
        }
    } else if (vary_only) {   // the destination of jac g holds its constant entries: the varying run of every block only
        if (h->wide && a.early) {   // (launches into host memory that send their early entries ahead)
            if (planar) hipLaunchKernelGGL((hipnlp_knot_kernel<HIPNLP_TERRAIN_PLANAR, 8, false, true, true>), grid, dim3(512), 0, s, a.tb, a.x, a.pk, a.gp, a.N, a.n, a.knot_begin, a.nk, a);
            else hipLaunchKernelGGL((hipnlp_knot_kernel<HIPNLP_TERRAIN_SMOOTH_STEPS, 8, false, true, true>), grid, dim3(512), 0, s, a.tb, a.x, a.pk, a.gp, a.N, a.n, a.knot_begin, a.nk, a);
        } else if (h->wide) {
            if (planar) hipLaunchKernelGGL((hipnlp_knot_kernel<HIPNLP_TERRAIN_PLANAR, 8, false, true>), grid, dim3(512), 0, s, a.tb, a.x, a.pk, a.gp, a.N, a.n, a.knot_begin, a.nk, a);
            else hipLaunchKernelGGL((hipnlp_knot_kernel<HIPNLP_TERRAIN_SMOOTH_STEPS, 8, false, true>), grid, dim3(512), 0, s, a.tb, a.x, a.pk, a.gp, a.N, a.n, a.knot_begin, a.nk, a);
        } else {
            if (planar) hipLaunchKernelGGL((hipnlp_knot_kernel<HIPNLP_TERRAIN_PLANAR, 4, false, true>), grid, dim3(256), 0, s, a.tb, a.x, a.pk, a.gp, a.N, a.n, a.knot_begin, a.nk, a);
            else hipLaunchKernelGGL((hipnlp_knot_kernel<HIPNLP_TERRAIN_SMOOTH_STEPS, 4, false, true>), grid, dim3(256), 0, s, a.tb, a.x, a.pk, a.gp, a.N, a.n, a.knot_begin, a.nk, a);
        }
    } else if (h->wide) {   // the whole launch resident at once: eight waves per knot
        if (planar) hipLaunchKernelGGL((hipnlp_knot_kernel<HIPNLP_TERRAIN_PLANAR, 8>), grid, dim3(512), 0, s, a.tb, a.x, a.pk, a.gp, a.N, a.n, a.knot_begin, a.nk, a);
        else hipLaunchKernelGGL((hipnlp_knot_kernel<HIPNLP_TERRAIN_SMOOTH_STEPS, 8>), grid, dim3(512), 0, s, a.tb, a.x, a.pk, a.gp, a.N, a.n, a.knot_begin, a.nk, a);
    } else {
        if (planar) hipLaunchKernelGGL((hipnlp_knot_kernel<HIPNLP_TERRAIN_PLANAR, 4>), grid, dim3(256), 0, s, a.tb, a.x, a.pk, a.gp, a.N, a.n, a.knot_begin, a.nk, a);
        else hipLaunchKernelGGL((hipnlp_knot_kernel<HIPNLP_TERRAIN_SMOOTH_STEPS, 4>), grid, dim3(256), 0, s, a.tb, a.x, a.pk, a.gp, a.N, a.n, a.knot_begin, a.nk, a);
    }
    if (prof) HIP_TRY(h, hipEventRecord(h->prof_ev[size_t(3 * h->prof_n + 1)], s));
    if (!fused && !cost_dst)
        hipLaunchKernelGGL(hipnlp_reduce_kernel, dim3(unsigned(h->batch)), dim3(RWG), 0, s, (const double*)h->d_cost_knot, h->nk, f_dev, a.cost_terms, a.cc);
    if (timed) HIP_TRY(h, hipEventRecord(e2, s));
    if (run_last) {   // (ONE event behind the run: kernel end and launch end coincide for a run)
        HIP_TRY(h, hipEventRecord(h->prof_ev[size_t(3 * h->prof_n + 2)], s));
        h->prof_open = false;
        h->prof_n++;
    }
    HIP_TRY(h, hipGetLastError());
    if (prof) h->prof_n++;
    if (timed) { h->last_e0 = e0; h->last_e2 = e2; h->timing_valid = true; }
    return HIPNLP_OK;
}


// =====================================================================================================================
// One caller, several devices (hipnlp_multi_create).  The reference has ONE caller of the callbacks — IPOPT, inside the process that runs
// self._solver.solve() (/root/reference/src/hippopt/base/opti_solver.py:479) — and that caller holds x and wants f, grad f, g, jac g in
// arrays of its own.  A multi-device handle serves it behind the host-buffer entry points of a plain handle: the horizon is cut into
// contiguous knot ranges (the defects of interval k -> k + 1 belong to the owner of knot k + 1, the naming of
// base/multiple_shooting_solver.py:713-742: a shard reads the record of the knot in front of its first one — the halo — besides its own,
// the six horizon-global variables, and the record of the other horizon end where it owns knot 0 or N - 1), one shard handle per device;
// a call stages x ONCE in a pinned block every device reads over its own link, launches every shard's kernel — which stores the
// shard's entries of grad f / g / jac g at their final places in the caller's registered arrays (or the front's pinned block), over
// that device's own link — and waits for all of them.  Nothing crosses between the devices; nothing is reassembled.
// The total cost is summed by the caller: every knot workgroup stores its twelve cost partials into a pinned block and the host adds
// them in the order of the device reduction (hipnlp_reduce_kernel) — f and the per-term costs are, bit for bit, those of ONE handle over
// the whole horizon whatever the number of shards.
// =====================================================================================================================
struct MultiState {
    std::vector<hipnlp_handle*> shards;
    double* h_cost = nullptr;             // pinned: shard s's block [batch][nk_s][NCT] at cost_off[s]
    std::vector<double*> hd_cost;         // ... and its device-visible address
    std::vector<size_t> cost_off;
    std::vector<double> shard_us;         // [shards][2] of the last evaluation: enqueue, completion seen (host clock, from the start of the enqueue loop)
    // ---- one launching thread per shard (hipnlp_multi_set_threads) -------------------------------------------------------------------
    // Enqueueing a kernel costs the host 4 - 6 us and waiting for a stream a few more: done by the caller's thread one shard after the
    // other, the last of eight shards would start 40 us after the first — longer than the whole callback of one device.  With threads on,
    // shard i > 0 has a worker bound to its device: the caller publishes a job (a ticket number), runs shard 0's part itself and waits
    // for the others; every worker stages / launches / waits for ITS shard.  A worker polls for the next ticket for `spin_us` after its
    // last job (IPOPT's callbacks come in bursts, microseconds apart) and sleeps on a condition variable after that (between bursts
    // IPOPT factorises for milliseconds: no core is burnt meanwhile; the first callback of a burst pays one wake-up).
    std::vector<std::thread> workers;
    std::function<int(size_t)> job;       // what every shard does for the ticket in flight; rc per shard in job_rc
    std::vector<int> job_rc;
    std::atomic<unsigned long long> ticket{0};
    std::atomic<int> done{0};
    std::atomic<int> sleepers{0};
    std::atomic<bool> quit{false};
    std::mutex mx;
    std::condition_variable cv;
    double spin_us = 200.0;
    bool threads = false;
};

static void multi_worker(MultiState* M, size_t i) {
    (void)hipSetDevice(M->shards[i]->dev);   // (the current device is a property of the thread: set once)
    unsigned long long seen = 0;
    auto last = std::chrono::steady_clock::now();
    for (;;) {
        unsigned long long t = M->ticket.load(std::memory_order_acquire);
        if (t == seen) {
            if (M->quit.load(std::memory_order_relaxed)) return;
            if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - last).count() < M->spin_us) { __builtin_ia32_pause(); continue; }
            std::unique_lock<std::mutex> lock(M->mx);
            M->sleepers.fetch_add(1, std::memory_order_seq_cst);
            M->cv.wait(lock, [&] { return M->ticket.load(std::memory_order_acquire) != seen || M->quit.load(std::memory_order_relaxed); });
            M->sleepers.fetch_sub(1, std::memory_order_seq_cst);
            continue;
        }
        seen = t;
        M->job_rc[i] = M->job(i);
        M->done.fetch_add(1, std::memory_order_release);
        last = std::chrono::steady_clock::now();
    }
}
// every shard runs job(i): by its worker (threads on), shard 0 by the caller; or one after the other by the caller.  First failure wins.
static int multi_failed(hipnlp_handle* h, size_t i, int rc);
static int multi_run(hipnlp_handle* h, std::function<int(size_t)> job) {
    MultiState& M = *h->multi;
    const size_t S = M.shards.size();
    if (!M.threads || S == 1) {
        for (size_t i = 0; i < S; ++i) {
            HIP_TRY(h, hipSetDevice(M.shards[i]->dev));
            const int rc = job(i);
            if (rc != HIPNLP_OK) return multi_failed(h, i, rc);
        }
        return HIPNLP_OK;
    }
    M.job = std::move(job);
    M.done.store(0, std::memory_order_relaxed);
    M.ticket.fetch_add(1, std::memory_order_seq_cst);
    if (M.sleepers.load(std::memory_order_seq_cst) > 0) { std::lock_guard<std::mutex> lock(M.mx); M.cv.notify_all(); }
    M.job_rc[0] = M.job(0);
    while (M.done.load(std::memory_order_acquire) < int(S) - 1) __builtin_ia32_pause();
    for (size_t i = 0; i < S; ++i) if (M.job_rc[i] != HIPNLP_OK) return multi_failed(h, i, M.job_rc[i]);
    return HIPNLP_OK;
}
static void multi_stop_workers(MultiState* M) {
    if (M->workers.empty()) return;
    {
        std::lock_guard<std::mutex> lock(M->mx);
        M->quit.store(true, std::memory_order_seq_cst);
    }
    M->cv.notify_all();
    for (std::thread& t : M->workers) t.join();
    M->workers.clear();
    M->quit.store(false, std::memory_order_seq_cst);
    M->threads = false;
}

static void multi_destroy(hipnlp_handle* h) {
    MultiState* M = h->multi;
    if (!M) return;
    multi_stop_workers(M);
    for (hipnlp_handle* s : M->shards) hipnlp_destroy(s);
    if (M->h_cost) (void)hipHostFree(M->h_cost);
    delete M;
    h->multi = nullptr;
}

static int multi_set_params(hipnlp_handle* h, const double* p) {
    for (hipnlp_handle* s : h->multi->shards) {
        const int rc = hipnlp_set_params(s, p);
        if (rc != HIPNLP_OK) { h->err = "shard on device " + std::to_string(s->dev) + ": " + s->err; return rc; }
    }
    return HIPNLP_OK;
}

// The cut, and what of x a shard reads (pure arithmetic: no device) — hipnlp_multi_create and multi_stage_x follow it; the CPU tests
// evaluate the knot program on x poisoned everywhere else (tests/test_multi_plan.py).
//   knots  contiguous, balanced, never empty: the first (horizon mod n_shards) shards own one knot more (hippopt_amd/sharded.py knot_range)
//   x      [0] the shard's own records and the HALO record in front of them (the trapezoid defect of interval k - 1 -> k is a row of knot k:
//              base/multiple_shooting_solver.py:713-742),  [1] the six horizon-global variables,  [2] the record of the LAST knot for the
//              owner of knot 0,  [3] the record of knot 0 for the owner of the last knot (the periodicity rows couple the two ends:
//              turnkey_planners/humanoid_kinodynamic/planner.py:897-930) — {offset, count} in doubles, count 0: nothing more to read
int hipnlp_multi_plan(int horizon, int n_shards, int shard, int32_t* knot_begin, int32_t* knot_end, int64_t* x_ranges) {
    if (horizon < 2 || n_shards < 1 || n_shards > horizon || shard < 0 || shard >= n_shards) return HIPNLP_E_INVALID;
    const int q = horizon / n_shards, r = horizon % n_shards;
    const int kb = shard * q + std::min(shard, r), ke = kb + q + (shard < r ? 1 : 0);
    if (knot_begin) *knot_begin = kb;
    if (knot_end) *knot_end = ke;
    if (x_ranges) {
        const int64_t N = horizon, k0 = kb > 0 ? kb - 1 : 0;
        const int64_t ranges[4][2] = {{NXK * k0, NXK * (ke - k0)}, {NXK * N, NXG},
                                      {NXK * (N - 1), (kb == 0 && ke < N) ? NXK : 0}, {0, (ke == N && kb > 1) ? NXK : 0}};
        std::memcpy(x_ranges, ranges, sizeof ranges);
    }
    return HIPNLP_OK;
}

// a range of every trajectory's row of a [batch][pitch] array between host and device (one copy command whatever the batch)
static hipError_t copy_rows(void* dst, const void* src, size_t pitch_doubles, size_t off, size_t count, size_t B, hipMemcpyKind kind, hipStream_t st) {
    char* d = static_cast<char*>(dst) + off * sizeof(double);
    const char* s = static_cast<const char*>(src) + off * sizeof(double);
    if (count == 0) return hipSuccess;
    if (B == 1) return hipMemcpyAsync(d, s, count * sizeof(double), kind, st);
    return hipMemcpy2DAsync(d, pitch_doubles * sizeof(double), s, pitch_doubles * sizeof(double), count * sizeof(double), B, kind, st);
}
// x for a shard whose launches read it from HBM (big batches): its own knots' records and the halo record in front of them, the
// horizon-global variables, and the record of the other horizon end for the owner of knot 0 / N - 1 (the periodicity rows) — out of
// the front's pinned copy into the SAME places of the shard's device array, on the shard's stream
// (errors are left in the SHARD's handle: this runs on the shard's launching thread when there is one)
static int multi_stage_x(const hipnlp_handle* h, hipnlp_handle* s, size_t shard) {
    const size_t B = size_t(h->batch), n = size_t(h->L.n);
    int64_t rg[4][2];
    if (hipnlp_multi_plan(h->L.N, int(h->multi->shards.size()), int(shard), nullptr, nullptr, &rg[0][0]) != HIPNLP_OK) { s->err = "internal: hipnlp_multi_plan"; return HIPNLP_E_INVALID; }
    for (int i = 0; i < 4; ++i)
        if (rg[i][1] > 0) HIP_TRY(s, copy_rows(s->d_x, h->h_x, n, size_t(rg[i][0]), size_t(rg[i][1]), B, hipMemcpyHostToDevice, s->stream));
    return HIPNLP_OK;
}
// the message of the shard that failed, into the front
static int multi_failed(hipnlp_handle* h, size_t i, int rc) {
    const hipnlp_handle* s = h->multi->shards[i];
    h->err = "shard " + std::to_string(i) + " on device " + std::to_string(s->dev) + ": " + s->err;
    return rc;
}
static int multi_sync(hipnlp_handle* h) {
    for (hipnlp_handle* s : h->multi->shards) HIP_TRY(h, hipStreamSynchronize(s->stream));
    return HIPNLP_OK;
}
static int multi_run(hipnlp_handle* h, std::function<int(size_t)> job);
// f[b] and cost_terms[b][.] from the knots' partials, in the order of hipnlp_reduce_kernel / the in-launch reducer: knot k belongs to
// group k mod 16, a group is summed in ascending k from +0 (with the same padding zeros), four groups of a quarter as (g0 + g1) + (g2 + g3),
// the quarters as ((p0 + p1) + p2) + p3, the twelve terms in order
static void multi_reduce_costs(hipnlp_handle* h) {
    const MultiState& M = *h->multi;
    const int N = h->L.N;
    std::vector<const double*> row(static_cast<size_t>(N), nullptr);   // partials of knot k: [NCT]
    for (int b = 0; b < h->batch; ++b) {
        for (size_t i = 0; i < M.shards.size(); ++i) {
            const hipnlp_handle* s = M.shards[i];
            for (int kk = 0; kk < s->nk; ++kk) row[size_t(s->kb + kk)] = M.h_cost + M.cost_off[i] + (size_t(b) * size_t(s->nk) + size_t(kk)) * NCT;
        }
        double tot = 0.0;
        for (int c = 0; c < NCT; ++c) {
            double G[RGRP];
            for (int g = 0; g < RGRP; ++g) {
                double acc = 0.0;
                for (int k0 = g; k0 < N; k0 += RGRP * RUNR)
                    for (int u = 0; u < RUNR; ++u) { const int k = k0 + u * RGRP; acc += k < N ? row[size_t(k)][c] : 0.0; }
                G[g] = acc;
            }
            double P[4];
            for (int w = 0; w < 4; ++w) P[w] = (G[4 * w] + G[4 * w + 1]) + (G[4 * w + 2] + G[4 * w + 3]);
            const double term = ((P[0] + P[1]) + P[2]) + P[3];
            h->h_cost_terms[size_t(b) * NCT + size_t(c)] = term;
            tot += term;
        }
        h->h_f[b] = tot;
    }
}
static_assert(RGRP == 16 && RWG / 64 == 4, "multi_reduce_costs restates the tree of hipnlp_reduce_kernel");

// One evaluation by all shards.  o[q] (grad, g, jac): a device-visible HOST address every shard stores its entries into — the caller's
// registered array or the front's pinned block — or null: the output stays in the HBM of each shard (its own d_grad / d_g / d_jac).
static int multi_evaluate(hipnlp_handle* h, double* const o[3], bool vary_only, std::chrono::steady_clock::time_point* t_enqueued) {
    MultiState& M = *h->multi;
    const auto t0 = std::chrono::steady_clock::now();
    double* const og = o[0]; double* const ogg = o[1]; double* const oj = o[2];
    // the part of shard i: stage (big batches), launch, wait — on its own stream, from whichever thread runs it
    auto enqueue = [&M, h, og, ogg, oj, vary_only, t0](size_t i) -> int {
        hipnlp_handle* s = M.shards[i];
        const double* xsrc = h->hd_x;
        if (!h->x_zero_copy) {
            const int rc = multi_stage_x(h, s, i);
            if (rc != HIPNLP_OK) return rc;
            xsrc = s->d_x;
        }
        s->early_store = h->early_store;
        const int rc = launch(s, xsrc, s->d_f, og ? og : s->d_grad, ogg ? ogg : s->d_g, oj ? oj : s->d_jac, s->stream, nullptr, false, false, true,
                              nullptr, 0, 0, vary_only && oj != nullptr, false, false, M.hd_cost[i]);
        M.shard_us[2 * i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        return rc;
    };
    auto wait = [&M, t0](size_t i) -> int {
        const hipError_t e = hipStreamSynchronize(M.shards[i]->stream);
        M.shard_us[2 * i + 1] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (e != hipSuccess) { M.shards[i]->err = std::string("hipStreamSynchronize: ") + hipGetErrorString(e); (void)hipGetLastError(); return HIPNLP_E_NODEVICE; }
        return HIPNLP_OK;
    };
    int rc;
    if (M.threads && M.shards.size() > 1) {
        rc = multi_run(h, [&](size_t i) -> int { const int r = enqueue(i); return r != HIPNLP_OK ? r : wait(i); });
        if (t_enqueued) *t_enqueued = t0 + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double, std::micro>(M.shard_us[0]));
    } else {   // the caller's thread: every launch first, then every wait
        rc = multi_run(h, enqueue);
        if (t_enqueued) *t_enqueued = std::chrono::steady_clock::now();
        if (rc == HIPNLP_OK) rc = multi_run(h, wait);
    }
    if (rc != HIPNLP_OK) return rc;
    if (h->seq == INT32_MAX) { std::memset(h->h_flag, 0, size_t(h->batch) * sizeof(int32_t)); h->seq = 0; }   // (the generation of the front's non-finite flags)
    ++h->seq;
    for (const hipnlp_handle* s : M.shards)
        for (int b = 0; b < h->batch; ++b) if (s->h_flag[b] == s->seq) h->h_flag[b] = h->seq;
    multi_reduce_costs(h);
    return HIPNLP_OK;
}
// A cached output that is still in the shards' HBM, asked for now (new_x = 0): every shard sends its part — q = 0: its knots' entries
// of grad f; q = 2: its knot blocks of jac g (vary_run: the varying run of every block, by a small kernel, into a destination that holds
// the constants) — to dst (host address: a registered caller array or the pinned block; dst_dev: its device-visible address).
// Enqueued on the shards' streams; multi_sync afterwards.  (g is never fetched this way: its rows are scattered over the constraint
// blocks; an evaluation that left g in HBM marks it `gone` and a later request evaluates again.)
static int multi_fetch(hipnlp_handle* h, int q, double* dst, double* dst_dev, bool vary_run) {
    const size_t B = size_t(h->batch);
    return multi_run(h, [=](size_t i) -> int {
        hipnlp_handle* s = h->multi->shards[i];
        const Layout& L = s->L;
        if (q == 0) HIP_TRY(s, copy_rows(dst, s->d_grad, size_t(L.n), size_t(s->dims.shard_grad_off), size_t(s->dims.shard_grad), B, hipMemcpyDeviceToHost, s->stream));
        else if (vary_run) {
            hipLaunchKernelGGL(hipnlp_fetch_vary_kernel, dim3(unsigned(s->nk), unsigned(s->batch)), dim3(256), 0, s->stream, (const double*)s->d_jac, dst_dev, s->kb, L.N,
                               int64_t(L.nnz), L.nnz_v[VAR_FIRST], L.nnz_v[VAR_INTERIOR], L.nvary_v[VAR_FIRST], L.nvary_v[VAR_INTERIOR], L.nvary_v[VAR_LAST]);
            HIP_TRY(s, hipGetLastError());
        } else HIP_TRY(s, copy_rows(dst, s->d_jac, size_t(L.nnz), size_t(s->dims.shard_jac_off), size_t(s->dims.shard_nnz), B, hipMemcpyDeviceToHost, s->stream));
        if (h->multi->threads) HIP_TRY(s, hipStreamSynchronize(s->stream));   // (its own thread waits for it; the caller's multi_sync then finds the streams idle)
        return HIPNLP_OK;
    });
}

int hipnlp_multi_set_threads(hipnlp_handle* h, int on, double spin_us);
int hipnlp_multi_create(const hipnlp_desc* desc, const int32_t* devices, int n_devices, hipnlp_handle** out) {
    if (!desc || !out || !devices) { g_create_error = "null argument"; return HIPNLP_E_INVALID; }
    *out = nullptr;
    if (n_devices < 1 || n_devices > 64) { g_create_error = "hipnlp_multi_create: 1 .. 64 devices"; return HIPNLP_E_INVALID; }
    const int N = desc->settings.horizon;
    if (!((desc->knot_begin == 0 && desc->knot_end == 0) || (desc->knot_begin == 0 && desc->knot_end == N))) {
        g_create_error = "hipnlp_multi_create: the descriptor names the whole horizon (knot_begin = knot_end = 0); the library cuts it";
        return HIPNLP_E_INVALID;
    }
    if (n_devices > N) { g_create_error = "hipnlp_multi_create: more devices than knots (every shard owns at least one knot)"; return HIPNLP_E_INVALID; }
    hipnlp_desc fd = *desc;
    fd.knot_begin = fd.knot_end = 0;
    fd.device = devices[0];
    hipnlp_handle* h = nullptr;
    int rc = create_handle(&fd, &h, true);
    if (rc != HIPNLP_OK) return rc;
    MultiState* M = new (std::nothrow) MultiState();
    if (!M) { g_create_error = "out of memory"; hipnlp_destroy(h); return HIPNLP_E_ALLOC; }
    h->multi = M;
    auto fail = [&](int code, const std::string& msg) { g_create_error = msg; hipnlp_destroy(h); return code; };
    size_t cost_doubles = 0;
    for (int i = 0; i < n_devices; ++i) {
        hipnlp_desc sd = *desc;
        sd.device = devices[i];
        (void)hipnlp_multi_plan(N, n_devices, i, &sd.knot_begin, &sd.knot_end, nullptr);   // (the cut)
        hipnlp_handle* s = nullptr;
        rc = hipnlp_create(&sd, &s);
        if (rc != HIPNLP_OK) return fail(rc, "shard " + std::to_string(i) + " on device " + std::to_string(devices[i]) + ": " + g_create_error);
        M->shards.push_back(s);
        // (the shard handles serve the front and nobody else: they register nothing by themselves)
        s->auto_reg = false;
        M->cost_off.push_back(cost_doubles);
        cost_doubles += size_t(s->batch) * size_t(s->nk) * NCT;
    }
    M->shard_us.assign(size_t(2 * n_devices), 0.0);
    if (hipSetDevice(h->dev) != hipSuccess || hipHostMalloc(&M->h_cost, cost_doubles * sizeof(double), hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
        (void)hipGetLastError();
        M->h_cost = nullptr;
        return fail(HIPNLP_E_ALLOC, "hipnlp_multi_create: pinned block of the cost partials");
    }
    std::memset(M->h_cost, 0, cost_doubles * sizeof(double));
    // the pinned blocks are addressed by the kernels of every device through ONE device-visible address (as the registered arrays of the
    // caller are: the table of ranges holds one address per range): a runtime that maps them elsewhere on some device is refused here
    for (int i = 0; i < n_devices; ++i) {
        void *cd = nullptr, *xd = nullptr, *od = nullptr;
        if (hipSetDevice(devices[i]) != hipSuccess || hipHostGetDevicePointer(&cd, M->h_cost, 0) != hipSuccess || hipHostGetDevicePointer(&xd, h->h_x, 0) != hipSuccess ||
            hipHostGetDevicePointer(&od, h->h_out, 0) != hipSuccess) {
            (void)hipGetLastError();
            return fail(HIPNLP_E_NODEVICE, "hipnlp_multi_create: device " + std::to_string(devices[i]) + " cannot address the pinned staging blocks");
        }
        if (xd != static_cast<void*>(h->hd_x) || od != static_cast<void*>(h->hd_f))
            return fail(HIPNLP_E_UNSUPPORTED, "hipnlp_multi_create: device " + std::to_string(devices[i]) + " maps pinned host memory at an address of its own");
        M->hd_cost.push_back(static_cast<double*>(cd) + M->cost_off[size_t(i)]);
    }
    (void)hipSetDevice(h->dev);
    *out = h;
    // one launching thread per shard by default: from the caller's thread alone the waits for the shards' streams come one after the other,
    // ~9 us each (measured with shards of one card: 47 / 64 / 114 us per 100-knot callback with 2 / 4 / 8 shards against 45 / 56 / 72 with
    // threads, profiles/r06_multi_probe.txt); hipnlp_multi_set_threads(h, 0, -1) turns them off
    if (n_devices > 1 && hipnlp_multi_set_threads(h, 1, -1.0) != HIPNLP_OK) h->err.clear();   // (no threads: the caller's thread does it all)
    return HIPNLP_OK;
}

int hipnlp_multi_info(const hipnlp_handle* h, int32_t* n_shards, int32_t* devices, int32_t* knot_begin, int32_t* knot_end, int32_t* waves) {
    if (!h || !n_shards) return HIPNLP_E_INVALID;
    const int cap = *n_shards;
    *n_shards = h->multi ? int(h->multi->shards.size()) : 0;
    if (!h->multi) return HIPNLP_OK;
    for (int i = 0; i < *n_shards && i < cap; ++i) {
        const hipnlp_handle* s = h->multi->shards[size_t(i)];
        if (devices) devices[i] = s->dev;
        if (knot_begin) knot_begin[i] = s->kb;
        if (knot_end) knot_end[i] = s->ke;
        if (waves) waves[i] = s->wide ? 8 : 4;
    }
    return HIPNLP_OK;
}
int hipnlp_multi_set_threads(hipnlp_handle* h, int on, double spin_us) {
    if (!h || !h->multi) return HIPNLP_E_INVALID;
    MultiState* M = h->multi;
    if (spin_us >= 0.0) M->spin_us = spin_us;
    if (!on) { multi_stop_workers(M); return HIPNLP_OK; }
    if (M->threads) return HIPNLP_OK;
    M->job_rc.assign(M->shards.size(), HIPNLP_OK);
    try {
        for (size_t i = 1; i < M->shards.size(); ++i) M->workers.emplace_back(multi_worker, M, i);
    } catch (...) {
        multi_stop_workers(M);
        h->err = "hipnlp_multi_set_threads: could not start the launching threads";
        return HIPNLP_E_ALLOC;
    }
    M->threads = true;
    return HIPNLP_OK;
}
int hipnlp_multi_breakdown(const hipnlp_handle* h, double* us) {
    if (!h || !h->multi || !us) return HIPNLP_E_INVALID;
    std::memcpy(us, h->multi->shard_us.data(), h->multi->shard_us.size() * sizeof(double));
    return HIPNLP_OK;
}

// ---- exact Hessian of the Lagrangian (IPOPT eval_h) ----------------------------------------------------------------------------
static int hess_prepare(hipnlp_handle* h) {
    if (h->hess_state == 1) return HIPNLP_OK;
    if (h->hess_state == 0 && !h->hess_layout_built) {
        h->hess_layout_ok = h->HL.build(h->d.settings, h->kt);
        h->hess_layout_built = true;
    }
    if (!h->hess_layout_ok) { h->hess_state = -1; h->err = h->HL.error; return HIPNLP_E_UNSUPPORTED; }
    if (h->multi) {   // a front handle keeps the layout (structure, counts); the tables and flags are the shards'
        for (hipnlp_handle* s : h->multi->shards) {
            const int rc = hess_prepare(s);
            if (rc != HIPNLP_OK) { h->err = "shard on device " + std::to_string(s->dev) + ": " + s->err; return rc; }
        }
        h->hess_state = 1;
        return HIPNLP_OK;
    }
    // device side: every piece is allocated at most once; a failure leaves the pieces that exist for the next attempt
    HIP_TRY(h, hipSetDevice(h->dev));
    if (!h->d_ht) HIP_TRY(h, hipMalloc(&h->d_ht, sizeof(HessTables)));
    if (!h->d_hflag) {
        HIP_TRY(h, hipMalloc(&h->d_hflag, size_t(h->batch) * sizeof(int32_t)));
        HIP_TRY(h, hipMemset(h->d_hflag, 0, size_t(h->batch) * sizeof(int32_t)));
    }
    if (!h->h_hflag) {
        HIP_TRY(h, hipHostMalloc(&h->h_hflag, size_t(h->batch) * sizeof(int32_t)));
        std::memset(h->h_hflag, 0, size_t(h->batch) * sizeof(int32_t));
        void* fd = nullptr;
        HIP_TRY(h, hipHostGetDevicePointer(&fd, h->h_hflag, 0));
        h->hd_hflag = static_cast<int32_t*>(fd);
    }
    HessTables* t = new HessTables();
    std::memset(t, 0, sizeof(HessTables));
    for (int i = 0; i < hk::COUNT; ++i) t->perm[i] = i < h->HL.nnz_knot ? h->HL.perm[size_t(i)] : -1;
    for (int i = 0; i < 84; ++i) t->perm_couple[i] = i < h->HL.n_couple ? h->HL.perm_couple[size_t(i)] : -1;
    t->nnz_knot = h->HL.nnz_knot; t->n_couple = h->HL.n_couple;
    for (int i = 0; i < HK_PAD; ++i) t->inv[i] = -1;
    for (int i = 0; i < h->HL.nnz_knot; ++i) t->inv[h->HL.perm[size_t(i)]] = int16_t(i);
    for (int i = 0; i < h->HL.n_couple; ++i) t->inv[h->HL.perm_couple[size_t(i)]] = int16_t(h->HL.nnz_knot + i);
    kh_fill_far_lists(h->kt, t->far);
    const hipError_t e = hipnlp_internal_memcpy(h->d_ht, t, sizeof(HessTables), hipMemcpyHostToDevice);
    delete t;
    if (e != hipSuccess) { h->err = std::string("Hessian tables: ") + hipGetErrorString(e); return HIPNLP_E_NODEVICE; }
    h->hess_state = 1;
    return HIPNLP_OK;
}
// entries of THIS handle's knots (a shard handle owns the blocks of its knots; the coupling entries belong to the last knot)
static int64_t hess_count(const hipnlp_handle* h) { return int64_t(h->HL.nnz_knot) * h->nk + (h->ke == h->L.N ? h->HL.n_couple : 0); }

int hipnlp_hess_nnz(hipnlp_handle* h, int64_t* nnz_h) {
    if (!h || !nnz_h) return HIPNLP_E_INVALID;
    const int rc = hess_prepare(h);
    if (rc != HIPNLP_OK) return rc;
    *nnz_h = hess_count(h);
    return HIPNLP_OK;
}
int hipnlp_hess_sparsity(hipnlp_handle* h, int32_t* irow, int32_t* jcol) {
    if (!h || !irow || !jcol) return HIPNLP_E_INVALID;
    const int rc = hess_prepare(h);
    if (rc != HIPNLP_OK) return rc;
    std::vector<int32_t> ir(size_t(h->HL.nnz)), jc(size_t(h->HL.nnz));
    h->HL.pattern(ir.data(), jc.data());
    const int64_t off = h->HL.knot_base(h->kb), cnt = hess_count(h);
    std::memcpy(irow, ir.data() + off, size_t(cnt) * sizeof(int32_t));
    std::memcpy(jcol, jc.data() + off, size_t(cnt) * sizeof(int32_t));
    return HIPNLP_OK;
}
static int hess_launch(hipnlp_handle* h, const double* x_dev, const double* sigma_dev, const double* lambda_dev, double* hess_dev, hipStream_t s,
                       bool host_block = false, bool early = false, int64_t whole_stride = 0) {
    // whole_stride > 0 (shard launches of a multi-device handle): hess_dev is the value array of the WHOLE horizon [batch][whole_stride]; the
    // shard's knot blocks go to their places in it
    HArgs a;
    a.tb = h->d_tb; a.ht = h->d_ht; a.x = x_dev; a.pk = h->d_pk; a.gp = h->d_gp;
    a.sigma = sigma_dev; a.lambda = lambda_dev; a.hess = hess_dev; a.flag = h->d_hflag;
    a.flag_host = host_block ? h->hd_hflag : nullptr;
    a.N = h->L.N; a.n = h->L.n; a.m = h->L.m; a.knot_begin = h->kb;
    a.hstride = hess_count(h); a.hoff = h->HL.knot_base(h->kb);
    if (whole_stride > 0) { a.hstride = whole_stride; a.hoff = 0; }
    if (h->hseq == INT32_MAX) {   // (generation of the Hessian kernel's non-finite flags: start over before it wraps)
        HIP_TRY(h, hipDeviceSynchronize());
        HIP_TRY(h, hipMemset(h->d_hflag, 0, size_t(h->batch) * sizeof(int32_t)));
        std::memset(h->h_hflag, 0, size_t(h->batch) * sizeof(int32_t));
        HIP_TRY(h, hipDeviceSynchronize());
        h->hseq = 0;
    }
    a.seq = ++h->hseq; a.nnz_knot = h->HL.nnz_knot;
    a.early_run = (host_block && early && !h->hess_compact && h->HL.early_run <= 3 * 256 && h->HL.early_phase == hess_early_phase(h->d.settings.terrain)) ? h->HL.early_run : 0;
#ifdef HIPNLP_STAMPS
    if (!h->d_stamps) HIP_TRY(h, hipMalloc(&h->d_stamps, 2 * size_t(h->nk) * size_t(h->batch) * 1024 * sizeof(unsigned long long)));   // (SPLIT launches: two workgroups per knot)
    a.stamps = h->d_stamps;
#endif
    const dim3 hgrid(unsigned(h->nk), unsigned(h->batch));
    const bool smooth = h->d.settings.terrain != HIPNLP_TERRAIN_PLANAR;
    a.trash = nullptr;
    if (h->hess_compact && !smooth && h->hess_direct && !host_block) {   // (the direct emitter's word per workgroup)
        if (!h->d_htrash) HIP_TRY(h, hipMalloc(&h->d_htrash, size_t(h->nk) * size_t(h->batch) * sizeof(double)));
        a.trash = h->d_htrash;
    }
    if (h->hess_compact) {   // (launches longer than the 512 workgroup slots of the full layout: three workgroups per CU)
        if (smooth) hipLaunchKernelGGL((hipnlp_knot_hess_kernel<HIPNLP_TERRAIN_SMOOTH_STEPS, LAYOUT_COMPACT>), hgrid, dim3(256), 0, s, a);
        // planar terrain, DEVICE destination: the entries go straight to their places (no LDS staging: four workgroups per CU); a host
        // destination keeps the staged kernel, whose copy-out leaves as one run per knot (scattered 8-byte stores do not belong on PCIe)
        else if (h->hess_direct && !host_block) hipLaunchKernelGGL((hipnlp_knot_hess_kernel<HIPNLP_TERRAIN_PLANAR, LAYOUT_COMPACT, true>), hgrid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((hipnlp_knot_hess_kernel<HIPNLP_TERRAIN_PLANAR, LAYOUT_COMPACT>), hgrid, dim3(256), 0, s, a);
    } else {
        if (smooth) hipLaunchKernelGGL((hipnlp_knot_hess_kernel<HIPNLP_TERRAIN_SMOOTH_STEPS, LAYOUT_FULL>), hgrid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((hipnlp_knot_hess_kernel<HIPNLP_TERRAIN_PLANAR, LAYOUT_FULL>), hgrid, dim3(256), 0, s, a);
    }
    HIP_TRY(h, hipGetLastError());
    return HIPNLP_OK;
}
int hipnlp_eval_hess_device(hipnlp_handle* h, const double* x_dev, const double* obj_factor_dev, const double* lambda_dev, double* hess_dev, void* stream) {
    if (!h || !x_dev || !obj_factor_dev || !lambda_dev || !hess_dev) return HIPNLP_E_INVALID;
    NOT_FRONT(h, "hipnlp_eval_hess_device");
    if (!h->params_set) { h->err = "parameters not set (hipnlp_set_params)"; return HIPNLP_E_PARAMS; }
    const int rc = hess_prepare(h);
    if (rc != HIPNLP_OK) return rc;
    HIP_TRY(h, hipSetDevice(h->dev));
    return hess_launch(h, x_dev, obj_factor_dev, lambda_dev, hess_dev, stream ? hipStream_t(stream) : h->stream);
}
// how a handle in auto mode (hipnlp_set_hessian_early_run(h, -1)) came to its choice
enum { HESS_WHY_UNDECIDED = 0, HESS_WHY_FORCED = 1, HESS_WHY_NOT_ARMABLE = 2, HESS_WHY_NUMA_SAME = 3, HESS_WHY_NUMA_OTHER = 4, HESS_WHY_MEASURED = 5 };
// NUMA node of the CPU the calling thread runs on (Linux sysfs: the node whose cpulist holds it); -1: not known / one node only says nothing
static int numa_node_of_calling_thread() {
    const int cpu = sched_getcpu();
    if (cpu < 0) return -1;
    for (int node = 0; node < 64; ++node) {
        FILE* f = std::fopen(("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist").c_str(), "r");
        if (!f) { if (node == 0) return -1; break; }
        char text[4096] = {0};
        const size_t got = std::fread(text, 1, sizeof text - 1, f);
        std::fclose(f);
        text[got] = 0;
        for (const char* c = text; *c;) {
            if (!std::isdigit(static_cast<unsigned char>(*c))) { ++c; continue; }
            char* end = nullptr;
            long lo = std::strtol(c, &end, 10), hi = lo;
            if (*end == '-') hi = std::strtol(end + 1, &end, 10);
            if (cpu >= lo && cpu <= hi) return node;
            c = end;
        }
    }
    return -1;
}
int hipnlp_eval_hess(hipnlp_handle* h, const double* x, const double* obj_factor, const double* lambda, double* hess) {
    return hipnlp_eval_hess_at(h, x, 1, obj_factor, lambda, hess);
}
int hipnlp_eval_hess_at(hipnlp_handle* h, const double* x, int new_x, const double* obj_factor, const double* lambda, double* hess) {
    if (!h || !x || !obj_factor || !lambda || !hess) return HIPNLP_E_INVALID;
    if (!h->params_set) { h->err = "parameters not set (hipnlp_set_params)"; return HIPNLP_E_PARAMS; }
    int rc = hess_prepare(h);
    if (rc != HIPNLP_OK) return rc;
    const size_t B = size_t(h->batch), n = size_t(h->L.n), m = size_t(h->L.m), hn = size_t(hess_count(h));
    HIP_TRY(h, hipSetDevice(h->dev));
    // inputs [sigma | lambda] in one device block with a pinned mirror: one H2D copy (the kernel gathers the multipliers through the
    // slot -> row map: scattered 8-byte reads belong in HBM, not on PCIe); every piece allocated at most once
    const unsigned pin_flags = h->is_front ? (hipHostMallocPortable | hipHostMallocMapped) : hipHostMallocDefault;   // (a front's blocks: every shard's device)
    if (!h->d_sigma && !h->is_front) { HIP_TRY(h, hipMalloc(&h->d_sigma, B * (1 + m) * sizeof(double))); h->d_lambda = h->d_sigma + B; }
    if (!h->h_sl) HIP_TRY(h, hipHostMalloc(&h->h_sl, B * (1 + m) * sizeof(double), pin_flags));
    if (!h->d_hess && !h->is_front) HIP_TRY(h, hipMalloc(&h->d_hess, B * hn * sizeof(double)));
    if (!h->h_hess) {
        HIP_TRY(h, hipHostMalloc(&h->h_hess, B * hn * sizeof(double), pin_flags));
        void* hd = nullptr;
        HIP_TRY(h, hipHostGetDevicePointer(&hd, h->h_hess, 0));
        h->hd_hess = static_cast<double*>(hd);
    }
    // x: the staging copy hipnlp_eval uses.  new_x = 0 (IPOPT's flag: eval_h at the x of the callbacks before it — every accepted
    // iterate): the copy of the previous call IS this x — no 151 KB host copy in front of the launch (1.7 - 2.2 us of a 100-knot call, tools/diag/hess_new_x_ab.py); new_x < 0:
    // unknown, compared; nothing staged yet: copied whatever the flag says
    bool stage_x = new_x != 0 || !h->x_staged;
    if (new_x < 0 && h->x_staged) stage_x = x != h->h_x && std::memcmp(x, h->h_x, B * n * sizeof(double)) != 0;
    if (new_x == 0 && h->x_staged && x != h->h_x) {
        // "the x of the callbacks before it" is the CALLER's claim about THIS handle's staged copy: a binding that evaluates the callbacks
        // on one handle and the Hessian on another, or that passes IPOPT's flag on after an evaluation elsewhere, would get the Hessian at
        // a stale x without any error.  Sixty-four words spread over x are compared (nanoseconds): a differing sample stages x as new_x = 1
        // would.  A sample, not a proof — the contract stays "same handle, same x" (include/hipnlp.h) — but the silent case is the one
        // where x differs nearly everywhere (another iterate).
        const size_t total = B * n, stride = std::max<size_t>(1, total / 64);
        for (size_t i = 0; i < total && !stage_x; i += stride) stage_x = std::memcmp(&x[i], &h->h_x[i], sizeof(double)) != 0;
        if (!stage_x) stage_x = std::memcmp(&x[total - 1], &h->h_x[total - 1], sizeof(double)) != 0;
    }
    if (stage_x) {
        h->have_result = false;   // (the staging copy of x is shared with hipnlp_eval)
        h->x_staged = false;
        std::memcpy(h->h_x, x, B * n * sizeof(double));
    }
    std::memcpy(h->h_sl, obj_factor, B * sizeof(double));
    std::memcpy(h->h_sl + B, lambda, B * m * sizeof(double));
    const double* xsrc = h->hd_x;   // x: read by the kernel straight from the pinned staging copy, as in hipnlp_eval
    if (!h->x_zero_copy && !h->multi) {
        if (stage_x) HIP_TRY(h, hipMemcpyAsync(h->d_x, h->h_x, B * n * sizeof(double), hipMemcpyHostToDevice, h->stream));
        xsrc = h->d_x;
    }
    h->x_staged = true;
    const double* sl_dev = h->d_sigma;
    if (h->lam_zero_copy) {   // the kernel gathers the multipliers straight out of the pinned block, no copy command in front of it
        void* hd = nullptr;
        HIP_TRY(h, hipHostGetDevicePointer(&hd, h->h_sl, 0));
        sl_dev = static_cast<const double*>(hd);
    } else if (!h->multi) {
        HIP_TRY(h, hipMemcpyAsync(h->d_sigma, h->h_sl, B * (1 + m) * sizeof(double), hipMemcpyHostToDevice, h->stream));
    }
    // one Hessian by this handle's device — or by every shard's: its knot blocks into their places of the whole value array `dest`, x and
    // the multipliers out of the front's pinned blocks (big batches: out of the shard's own device copies, made on its stream first)
    auto run = [&](double* dest, bool early_run) -> int {
        if (!h->multi) {
            const int rl = hess_launch(h, xsrc, sl_dev, sl_dev + B, dest, h->stream, true, early_run);
            if (rl != HIPNLP_OK) return rl;
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            return HIPNLP_OK;
        }
        const int rm = multi_run(h, [&](size_t i) -> int {   // (shard i: on its launching thread when there is one; errors in the shard's handle)
            hipnlp_handle* s = h->multi->shards[i];
            const double *xs = xsrc, *sl = sl_dev;
            if (!h->x_zero_copy) {
                const int rs = multi_stage_x(h, s, i);
                if (rs != HIPNLP_OK) return rs;
                xs = s->d_x;
            }
            if (!h->lam_zero_copy) {
                if (!s->d_sigma) { HIP_TRY(s, hipMalloc(&s->d_sigma, B * (1 + m) * sizeof(double))); s->d_lambda = s->d_sigma + B; }
                HIP_TRY(s, hipMemcpyAsync(s->d_sigma, h->h_sl, B * (1 + m) * sizeof(double), hipMemcpyHostToDevice, s->stream));
                sl = s->d_sigma;
            }
            s->early_store = h->early_store;
            const int rl = hess_launch(s, xs, sl, sl + B, dest, s->stream, true, early_run, int64_t(hn));
            if (rl != HIPNLP_OK) return rl;
            if (!h->multi->threads) return HIPNLP_OK;   // (the caller's thread: every launch first, the waits below)
            HIP_TRY(s, hipStreamSynchronize(s->stream));
            return HIPNLP_OK;
        });
        if (rm != HIPNLP_OK) return rm;
        return h->multi->threads ? HIPNLP_OK : multi_sync(h);
    };
    // the values leave the kernel as PCIe stores into the pinned block — or straight into the CALLER'S array when that lies in a
    // registered range (no 1.2 MB host copy behind the launch): registered by the caller, or by the handle itself at the array's second
    // consecutive sight (IPOPT evaluates the Hessian into the value array of its own matrix); a store into an array the handle
    // registered is verified with the sentinel words of the callback path, and served through the pinned block if it did not arrive
    const size_t hbytes = B * hn * sizeof(double);
    h->in_call[0] = h->in_call[1] = h->in_call[2] = nullptr; h->in_call[3] = hess;
    double* direct = caller_array_address(h, 3, hess, hbytes);
    h->in_call[3] = nullptr;
    u64 sentinel = 0;
    if (direct && auto_owns(h, hess)) {
        sentinel = 0x7FF8C0DE00000000ull | (++h->sentinel_salt & 0xFFFFFFFFull);
        u64* w = reinterpret_cast<u64*>(hess);
        __atomic_store_n(&w[0], sentinel, __ATOMIC_RELAXED);
        __atomic_store_n(&w[hbytes / 8 - 1], sentinel, __ATOMIC_RELAXED);
    }
    // early run: forced on / off, or — the default — decided ONCE per handle, by what the profiles say decides it (profiles/
    // r05_early_stores_by_box.txt): a gain when the calling thread sits on the card's NUMA node, a loss across the socket interconnect.
    //   1. a launch that cannot arm the early run at all (compact Hessian layout: long launches; a pattern whose early run is not the one
    //      the kernel was built for) has nothing to decide: off, no sampling;
    //   2. both NUMA nodes known: on if they are the same node, off otherwise — no clock involved: the same choice in every run;
    //   3. otherwise (one-node hosts, containers that hide the topology): measured on the handle's own first calls — three to warm up, then
    //      nine of each kind alternating, launch to completion on the host's clock, the lower MEDIAN wins (one scheduler hiccup among
    //      six samples and a minimum used to flip this).
    // hipnlp_get_hessian_early_run / hipnlp_hessian_early_run_reason say what was decided and why.
    constexpr int TUNE_WARM = 3, TUNE_SAMPLES = 18;
    bool early = h->early_store && h->hess_early_mode != 0;
    bool sample = false;
    if (h->hess_early_mode >= 0) h->hess_early_why = HESS_WHY_FORCED;
    if (h->early_store && h->hess_early_mode < 0) {
        if (h->hess_early_choice < 0) {
            const hipnlp_handle* s0 = h->multi ? h->multi->shards[0] : h;   // (the shards of a multi-device handle share the layout decisions of their sizes)
            const bool armable = !s0->hess_compact && h->HL.early_run <= 3 * 256 && h->HL.early_run > 0 && h->HL.early_phase == hess_early_phase(h->d.settings.terrain);
            if (!armable) { h->hess_early_choice = 0; h->hess_early_why = HESS_WHY_NOT_ARMABLE; }
            else if (h->hess_tune_calls == 0) {
                h->hess_caller_node = numa_node_of_calling_thread();
                if (hipnlp_device_numa_node(h->dev, &h->hess_card_node) != HIPNLP_OK) h->hess_card_node = -1;
                if (h->hess_caller_node >= 0 && h->hess_card_node >= 0) {
                    h->hess_early_choice = h->hess_caller_node == h->hess_card_node ? 1 : 0;
                    h->hess_early_why = h->hess_early_choice ? HESS_WHY_NUMA_SAME : HESS_WHY_NUMA_OTHER;
                }
            }
        }
        if (h->hess_early_choice >= 0) early = h->hess_early_choice != 0;
        else {
            const int c = h->hess_tune_calls++;
            sample = c >= TUNE_WARM;
            early = sample ? ((c - TUNE_WARM) & 1) != 0 : true;
        }
    }
    const auto t_launch = std::chrono::steady_clock::now();
    rc = run(direct ? direct : h->hd_hess, early);
    if (rc != HIPNLP_OK) return rc;
    if (sample) {
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_launch).count();
        h->hess_tune_us[early ? 1 : 0].push_back(us);
        if (h->hess_tune_calls >= TUNE_WARM + TUNE_SAMPLES) {
            for (int q = 0; q < 2; ++q) {   // (medians: reported through hipnlp_get_hessian_early_run)
                std::vector<double>& v = h->hess_tune_us[q];
                std::sort(v.begin(), v.end());
                h->hess_tune_best[q] = v.empty() ? 1e30 : v[v.size() / 2];
            }
            h->hess_early_choice = h->hess_tune_best[1] <= h->hess_tune_best[0] ? 1 : 0;
            h->hess_early_why = HESS_WHY_MEASURED;
        }
    }
    if (sentinel) {
        const u64* w = reinterpret_cast<const u64*>(hess);
        if (__atomic_load_n(&w[0], __ATOMIC_RELAXED) == sentinel || __atomic_load_n(&w[hbytes / 8 - 1], __ATOMIC_RELAXED) == sentinel) {
            (void)drop_stale_range(h, hess);
            h->no_auto[3] = hess;
            h->auto_fallbacks++;
            rc = run(h->hd_hess, early);
            if (rc != HIPNLP_OK) return rc;
            direct = nullptr;
        }
    }
    if (!direct) std::memcpy(hess, h->h_hess, hbytes);
    if (h->multi) {
        for (const hipnlp_handle* s : h->multi->shards)
            for (size_t b = 0; b < B; ++b)
                if (s->h_hflag[b] == s->hseq) { h->err = "non-finite value produced by the evaluation"; return HIPNLP_E_NUMERIC; }
        return HIPNLP_OK;
    }
    for (size_t b = 0; b < B; ++b)
        if (h->h_hflag[b] == h->hseq) { h->err = "non-finite value produced by the evaluation"; return HIPNLP_E_NUMERIC; }
    return HIPNLP_OK;
}

// Is p memory of the handle's own device?  (A VARY launch samples the constants it finds in its jac destination — reads that belong in
// HBM, not on PCIe or xGMI: a destination inside a registered host range (a shared host sink) or in a peer's buffer (opened through IPC) is
// filled like any other and never read back.)  Asked once per destination, when it is first seen.
static bool is_local_device_memory(const hipnlp_handle* h, const void* p) {
    {
        std::lock_guard<std::mutex> lock(g_ranges_mutex);
        const char* q = static_cast<const char*>(p);
        for (const HostRange& r : g_ranges) if (q >= r.dev && q < r.dev + r.bytes) return false;
    }
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return true; }   // (unknown to the runtime: treated as the caller says — device memory)
    return at.type == hipMemoryTypeDevice && at.device == h->dev;
}
int hipnlp_eval_device(hipnlp_handle* h, const double* x_dev, double* f_dev, double* grad_dev, double* g_dev, double* jac_dev, void* stream) {
    if (!h || !x_dev) return HIPNLP_E_INVALID;
    NOT_FRONT(h, "hipnlp_eval_device");
    if (!h->params_set) { h->err = "parameters not set (hipnlp_set_params)"; return HIPNLP_E_PARAMS; }
    HIP_TRY(h, hipSetDevice(h->dev));
    h->have_result = false;
    hipStream_t s = stream ? hipStream_t(stream) : h->stream;
    // A jac buffer is filled with the constant entries at its first sight (and again after hipnlp_set_params), on the caller's stream in
    // front of the evaluation; from then on the launches store the varying entries of every block only — one run in the varying-first
    // order, scattered 8-byte stores in CasADi's CCS order (they meet in the L2 before they leave for HBM) — and check the constants
    // they find (a buffer the caller wrote over is repaired by the kernel itself).
    bool vary = false, remote = false;
    int parity = -1;
    if (jac_dev && h->skip_const && h->vary_ok && h->d_ctpl && h->L.nconst_total > 0) {
        int slot = -1;
        for (int i = 0; i < 8; ++i) if (h->dfilled[i].dev == jac_dev) slot = i;
        if (slot < 0 || h->dfilled[slot].gen != h->param_gen) {
            const Layout& L = h->L;
            auto fc = [&](int v) { return h->cpos[v].empty() ? L.nnz_v[v] : int(h->cpos[v][0]); };   // first constant position of a block (none: its end)
            hipLaunchKernelGGL(hipnlp_fill_const_kernel, dim3(unsigned(h->nk), unsigned(h->batch)), dim3(256), 0, s, jac_dev, (const double*)h->d_ctpl,
                               (const int32_t*)h->d_ctpl_of_b, h->ctpl_len, h->ctpl_off[0], h->ctpl_off[1], h->ctpl_off[2], h->ctpl_off[3], h->kb, L.N, int64_t(L.nnz),
                               L.nnz_v[VAR_FIRST], L.nnz_v[VAR_INTERIOR], L.nnz_v[VAR_LAST], fc(VAR_FIRST), fc(VAR_INTERIOR), fc(VAR_LAST),
                               h->ke == L.N ? int(L.jperm_glob.size()) : 0);
            HIP_TRY(h, hipGetLastError());
            if (slot < 0) { slot = h->dfilled_next; h->dfilled_next = (h->dfilled_next + 1) % 8; }
            h->dfilled[slot] = {jac_dev, h->param_gen, !is_local_device_memory(h, jac_dev), 0u};
            h->dev_const_fills++;
        }
        vary = true;
        remote = h->dfilled[slot].remote;
        parity = int(h->dfilled[slot].launches++ & 1u);   // (per destination: every buffer sees all four sample positions of a block in two launches INTO IT)
    }
    return launch(h, x_dev, f_dev ? f_dev : h->d_f, grad_dev, g_dev, jac_dev, s, nullptr, false, false, false, nullptr, 0, 0, vary, false, remote, nullptr, parity);
}

int hipnlp_eval_device_shard(hipnlp_handle* h, const double* x_dev, double* f_dev, double* grad_shard, double* g_stage, double* jac_shard, void* stream) {
    if (!h || !x_dev) return HIPNLP_E_INVALID;
    NOT_FRONT(h, "hipnlp_eval_device_shard");
    if (!h->params_set) { h->err = "parameters not set (hipnlp_set_params)"; return HIPNLP_E_PARAMS; }
    HIP_TRY(h, hipSetDevice(h->dev));
    h->have_result = false;
    return launch(h, x_dev, f_dev ? f_dev : h->d_f, grad_shard, nullptr, jac_shard, stream ? hipStream_t(stream) : h->stream, g_stage, true);
}

int hipnlp_eval_device_peers(hipnlp_handle* h, const double* x_dev, double* const* peer_out_dev, int world, int rank, void* stream) {
    if (!h || !x_dev || !peer_out_dev || world < 1 || rank < 0 || rank >= 64) return HIPNLP_E_INVALID;   // (world: buffers stored to; rank: this shard's cost slot)
    NOT_FRONT(h, "hipnlp_eval_device_peers");
    if (!h->params_set) { h->err = "parameters not set (hipnlp_set_params)"; return HIPNLP_E_PARAMS; }
    if (h->batch != 1) { h->err = "hipnlp_eval_device_peers: one trajectory per handle (batch 1)"; return HIPNLP_E_INVALID; }
    if (!h->fused) { h->err = "hipnlp_eval_device_peers: shards of at most 256 knots (the shard's cost is summed inside the launch)"; return HIPNLP_E_INVALID; }
    HIP_TRY(h, hipSetDevice(h->dev));
    h->have_result = false;
    return launch(h, x_dev, h->d_f, nullptr, nullptr, nullptr, stream ? hipStream_t(stream) : h->stream, nullptr, false, false, false, peer_out_dev, world, rank);
}

// ---- what an exchange between GPUs moves: the varying runs only (include/hipnlp.h, "Exchanges without the constants of jac g") ------------
static int vary_ready(hipnlp_handle* h, const char* who) {
    if (!h->params_set) { h->err = "parameters not set (hipnlp_set_params)"; return HIPNLP_E_PARAMS; }
    if (!h->L.vary_first) { h->err = std::string(who) + ": the handle must list the varying entries of a knot block first (HIPNLP_FLAG_JAC_VARYING_FIRST)"; return HIPNLP_E_UNSUPPORTED; }
    if (!h->vary_ok || !h->d_ctpl) { h->err = std::string(who) + ": this settings combination has more varying entries per block than the kernels' trip count"; return HIPNLP_E_UNSUPPORTED; }
    return HIPNLP_OK;
}
int hipnlp_jac_vary_layout(const hipnlp_handle* h, int64_t* out) {
    if (!h || !out) return HIPNLP_E_INVALID;
    const Layout& L = h->L;
    out[0] = L.nvary_v[VAR_FIRST]; out[1] = L.N >= 3 ? L.nvary_v[VAR_INTERIOR] : 0; out[2] = L.nvary_v[VAR_LAST];
    out[3] = vary_base(L, L.N); out[4] = vary_base(L, h->kb); out[5] = vary_base(L, h->ke) - vary_base(L, h->kb);
    return HIPNLP_OK;
}
int hipnlp_fill_jac_constants(hipnlp_handle* h, double* jac_dev, int whole_horizon, void* stream) {
    if (!h || !jac_dev) return HIPNLP_E_INVALID;
    NOT_FRONT(h, "hipnlp_fill_jac_constants");
    if (!h->params_set) { h->err = "parameters not set (hipnlp_set_params)"; return HIPNLP_E_PARAMS; }
    if (!h->d_ctpl || !h->d_ctpl_of_b) { h->err = "hipnlp_fill_jac_constants: no constant templates for this handle"; return HIPNLP_E_UNSUPPORTED; }
    HIP_TRY(h, hipSetDevice(h->dev));
    const Layout& L = h->L;
    const int kb = whole_horizon ? 0 : h->kb, ke = whole_horizon ? L.N : h->ke;
    auto fc = [&](int v) { return h->cpos[v].empty() ? L.nnz_v[v] : int(h->cpos[v][0]); };
    hipLaunchKernelGGL(hipnlp_fill_const_kernel, dim3(unsigned(ke - kb), unsigned(h->batch)), dim3(256), 0, stream ? hipStream_t(stream) : h->stream, jac_dev,
                       (const double*)h->d_ctpl, (const int32_t*)h->d_ctpl_of_b, h->ctpl_len, h->ctpl_off[0], h->ctpl_off[1], h->ctpl_off[2], h->ctpl_off[3], kb, L.N,
                       int64_t(L.nnz), L.nnz_v[VAR_FIRST], L.nnz_v[VAR_INTERIOR], L.nnz_v[VAR_LAST], fc(VAR_FIRST), fc(VAR_INTERIOR), fc(VAR_LAST),
                       ke == L.N ? int(L.jperm_glob.size()) : 0);
    HIP_TRY(h, hipGetLastError());
    return HIPNLP_OK;
}
int hipnlp_eval_device_vary(hipnlp_handle* h, const double* x_dev, double* f_dev, double* grad_dev, double* g_dev, double* jac_vary_dev, void* stream) {
    if (!h || !x_dev) return HIPNLP_E_INVALID;
    NOT_FRONT(h, "hipnlp_eval_device_vary");
    const int rc = vary_ready(h, "hipnlp_eval_device_vary");
    if (rc != HIPNLP_OK) return rc;
    HIP_TRY(h, hipSetDevice(h->dev));
    h->have_result = false;
    return launch(h, x_dev, f_dev ? f_dev : h->d_f, grad_dev, g_dev, jac_vary_dev, stream ? hipStream_t(stream) : h->stream, nullptr, false, false, false, nullptr, 0, 0, true, true);
}
int hipnlp_eval_device_shard_vary(hipnlp_handle* h, const double* x_dev, double* f_dev, double* grad_shard, double* g_stage, double* jac_vary_shard, void* stream) {
    if (!h || !x_dev) return HIPNLP_E_INVALID;
    NOT_FRONT(h, "hipnlp_eval_device_shard_vary");
    const int rc = vary_ready(h, "hipnlp_eval_device_shard_vary");
    if (rc != HIPNLP_OK) return rc;
    if (g_stage && !h->gs_compact_ok) { h->err = "hipnlp_eval_device_shard_vary: a row stride of this layout does not fit the copy-out word"; return HIPNLP_E_UNSUPPORTED; }
    HIP_TRY(h, hipSetDevice(h->dev));
    h->have_result = false;
    return launch(h, x_dev, f_dev ? f_dev : h->d_f, grad_shard, nullptr, jac_vary_shard, stream ? hipStream_t(stream) : h->stream, g_stage, true, false, false, nullptr, 0, 0, true, true);
}
int hipnlp_eval_device_peers_vary(hipnlp_handle* h, const double* x_dev, double* const* peer_out_dev, int world, int rank, void* stream) {
    if (!h || !x_dev || !peer_out_dev || world < 1 || rank < 0 || rank >= 64) return HIPNLP_E_INVALID;
    NOT_FRONT(h, "hipnlp_eval_device_peers_vary");
    const int rc = vary_ready(h, "hipnlp_eval_device_peers_vary");
    if (rc != HIPNLP_OK) return rc;
    if (h->batch != 1) { h->err = "hipnlp_eval_device_peers_vary: one trajectory per handle (batch 1)"; return HIPNLP_E_INVALID; }
    if (!h->fused) { h->err = "hipnlp_eval_device_peers_vary: shards of at most 256 knots (the shard's cost is summed inside the launch)"; return HIPNLP_E_INVALID; }
    HIP_TRY(h, hipSetDevice(h->dev));
    h->have_result = false;
    return launch(h, x_dev, h->d_f, nullptr, nullptr, nullptr, stream ? hipStream_t(stream) : h->stream, nullptr, false, false, false, peer_out_dev, world, rank, true);
}

int hipnlp_stage_rows(const hipnlp_handle* h, int k, int32_t* rows) {
    if (!h || !rows || k < 0 || k >= h->L.N) return HIPNLP_E_INVALID;
    const int v = h->L.variant_of(k);
    for (int s = 0; s < gs::COUNT; ++s) {
        const int a = h->L.g_a[v][size_t(s)];
        rows[s] = a != G_NONE ? a + h->L.g_b[size_t(s)] * k : -1;
    }
    return HIPNLP_OK;
}

// ---- auto-registration of caller arrays ----------------------------------------------------------------------------------------------
// (the ranges a handle registered by itself are the entries of g_ranges it owns; an explicit hipnlp_host_register of the same array
//  takes the entry over, an explicit hipnlp_host_unregister removes it whoever owns it)
static void auto_unregister_all(hipnlp_handle* h) {
    std::vector<char*> mine;
    {
        std::lock_guard<std::mutex> lock(g_ranges_mutex);
        for (const HostRange& r : g_ranges) if (r.owner == h) mine.push_back(r.host);
    }
    for (char* p : mine) (void)hipnlp_host_unregister(p);
}
static size_t auto_count(const hipnlp_handle* h, char** oldest = nullptr) {
    std::lock_guard<std::mutex> lock(g_ranges_mutex);
    size_t n = 0;
    for (const HostRange& r : g_ranges) if (r.owner == h) { if (n == 0 && oldest) *oldest = r.host; ++n; }
    return n;
}
// Is p inside a range that SOME handle registered by itself?  Such a mapping may have gone stale behind the library's back (the array
// freed, its address reused), whichever handle uses it and wherever in the range p lies: every direct store into it is verified.
// (The table is process wide: a second handle, or a pointer into the middle of a range, finds the same device address.)
static bool auto_owns(const hipnlp_handle*, const void* p) {
    std::lock_guard<std::mutex> lock(g_ranges_mutex);
    const char* q = static_cast<const char*>(p);
    for (const HostRange& r : g_ranges) if (r.owner != nullptr && q >= r.host && q < r.host + r.bytes) return true;
    return false;
}
// start of the registered range that contains p (what hipnlp_host_unregister wants), or null
static void* range_base_of(const void* p) {
    std::lock_guard<std::mutex> lock(g_ranges_mutex);
    const char* q = static_cast<const char*>(p);
    for (const HostRange& r : g_ranges) if (q >= r.host && q < r.host + r.bytes) return r.host;
    return nullptr;
}
// a range is gone (stale mapping, eviction): nothing of the handle may point into it any more
static void forget_range(hipnlp_handle* h, const char* base, size_t bytes) {
    for (int q = 0; q < 3; ++q) {
        const char* sh = reinterpret_cast<const char*>(h->seen_host[q]);
        if (sh && sh >= base && sh < base + bytes) h->seen_host[q] = nullptr;
        const char* eh = reinterpret_cast<const char*>(h->early_host[q]);
        if (eh && eh >= base && eh < base + bytes) { h->early_host[q] = nullptr; h->early_mask &= ~(q == 0 ? HIPNLP_WANT_GRAD : (q == 1 ? HIPNLP_WANT_G : HIPNLP_WANT_JAC)); }
    }
    for (int i = 0; i < 8; ++i) {
        const char* c = static_cast<const char*>(h->cfilled[i].host);
        if (c && c >= base && c < base + bytes) h->cfilled[i] = {nullptr, 0, 0};
    }
}
static int drop_stale_range(hipnlp_handle* h, void* p) {
    void* base = range_base_of(p);
    if (!base) return HIPNLP_E_INVALID;
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> lock(g_ranges_mutex);
        for (const HostRange& r : g_ranges) if (r.host == base) bytes = r.bytes;
    }
    forget_range(h, static_cast<const char*>(base), bytes);
    return hipnlp_host_unregister(base);
}
constexpr size_t AUTO_MIN_BYTES = 64 * 1024;   // smaller arrays are cheaper to copy than to page-lock
constexpr size_t AUTO_MAX = 6;
// device-visible address of the caller's output array q (0 grad, 1 g, 2 jac, 3 Hessian values), registering it at its second consecutive sight
static double* caller_array_address(hipnlp_handle* h, int q, double* p, size_t bytes) {
    if (!p) return nullptr;   // (IPOPT passes one array per callback: "consecutive" means consecutive calls that pass this output)
    double* dev = device_address_of(p, bytes);
    if (!dev && h->auto_reg && bytes >= AUTO_MIN_BYTES && h->last_seen[q] == p && h->no_auto[q] != p) {
        if (auto_count(h) >= AUTO_MAX) {
            // the oldest range of this handle that THIS call does not use goes (a range whose device address the call has already
            // taken — grad before g before jac — must outlive the launch)
            char* victim = nullptr;
            size_t vbytes = 0;
            {
                std::lock_guard<std::mutex> lock(g_ranges_mutex);
                for (const HostRange& r : g_ranges) {
                    if (r.owner != h) continue;
                    bool used = false;
                    for (int i = 0; i < 4; ++i) {
                        const char* c = static_cast<const char*>(h->in_call[i]);
                        used |= c && c >= r.host && c < r.host + r.bytes;
                    }
                    if (!used) { victim = r.host; vbytes = r.bytes; break; }
                }
            }
            if (!victim) { h->last_seen[q] = p; return nullptr; }   // (every range is in use by this call: the array goes through the pinned block)
            forget_range(h, victim, vbytes);
            (void)hipnlp_host_unregister(victim);
        }
        void* d = nullptr;
        if (hipnlp_host_register(p, bytes, &d) == HIPNLP_OK) {
            // (diagnostic, tests only: HIPNLP_DEBUG_MISDIRECT_AUTO maps the array to the pinned block's copy of the output instead —
            //  stores that do not arrive in the caller's pages, the situation the sentinel check exists for, without unmapping anything)
            if (diag_env("HIPNLP_DEBUG_MISDIRECT_AUTO")) d = q == 0 ? (void*)h->hd_grad : (q == 1 ? (void*)h->hd_g : (q == 2 ? (void*)h->hd_jac : (void*)h->hd_hess));
            {
                std::lock_guard<std::mutex> lock(g_ranges_mutex);
                for (HostRange& r : g_ranges) if (r.host == reinterpret_cast<char*>(p)) { r.owner = h; r.dev = static_cast<char*>(d); }
            }
            h->auto_registered++;
            dev = static_cast<double*>(d);
        } else {
            h->no_auto[q] = p;
        }
    }
    h->last_seen[q] = p;
    return dev;
}

// Host-buffer path.  Makes the outputs named by `want` of the evaluation at x present on the host: in the pinned block or — `direct` —
// in the caller's own array.
// A NEW evaluation: x is copied into the pinned staging block (the kernel reads it from there), ONE evaluation computes all four
// outputs; each goes to one of three places: the caller's registered array (wanted by this call, or — early outputs — seen in an
// earlier call), the pinned block (wanted or in the prefetch set), or device memory.  A cached evaluation (new_x = 0): outputs not on
// the host yet are fetched from the device copies with one asynchronous copy each.  (Measured on MI355X, profiles/r02_pcie_probe.txt,
// r03_doorbell_probe.txt: launch + synchronise 13 us; kernel stores to pinned memory 56 GB/s; the same bytes through device memory +
// hipMemcpyAsync: + 10 us.  A resident kernel behind a doorbell in place of launch + synchronise was built, measured slower
// once x is read fresh and the outputs are made visible to the host, and removed: profiles/r03_resident_experiment.txt.)
struct HostDest {   // registered caller arrays: device-visible addresses (or null) and the host addresses they belong to
    double *grad = nullptr, *g = nullptr, *jac = nullptr;
    double *grad_host = nullptr, *g_host = nullptr, *jac_host = nullptr;
};
static int host_evaluate(hipnlp_handle* h, const double* x, int new_x, unsigned want, const HostDest& dst = HostDest(), unsigned* direct = nullptr) {
    const size_t B = size_t(h->batch), n = size_t(h->L.n), m = size_t(h->L.m), nnz = size_t(h->L.nnz);
    const unsigned bit[3] = {HIPNLP_WANT_GRAD, HIPNLP_WANT_G, HIPNLP_WANT_JAC};
    const size_t bytes[3] = {B * n * sizeof(double), B * m * sizeof(double), B * nnz * sizeof(double)};
    if (direct) *direct = 0;
    if (new_x < 0 && h->have_result) new_x = (x != h->h_x && std::memcmp(x, h->h_x, B * n * sizeof(double)) != 0) ? 1 : 0;   // unknown: compare
    if (new_x || !h->have_result) {
        HIP_TRY(h, hipSetDevice(h->dev));
        const unsigned to_host = (want | h->prefetch) & HIPNLP_WANT_ALL;
        const auto t0 = std::chrono::steady_clock::now();
        h->x_staged = false;
        if (x != h->h_x) std::memcpy(h->h_x, x, B * n * sizeof(double));
        const auto t1 = std::chrono::steady_clock::now();
        h->have_result = false;
        // destination of every output: 2 = the caller's registered array (then NOT in the pinned block: a later new_x = 0 request for
        // it from another array is re-evaluated — IPOPT never asks twice for the same output at one x), 1 = pinned block, 0 = HBM
        double* const dev_of_dst[3] = {dst.grad, dst.g, dst.jac};
        double* const host_of_dst[3] = {dst.grad_host, dst.g_host, dst.jac_host};
        double* caller_dev[3] = {nullptr, nullptr, nullptr};
        double* caller_host[3] = {nullptr, nullptr, nullptr};
        unsigned sel[3], dmask = 0;
        h->early_mask = 0;
        for (int q = 0; q < 3; ++q) {
            sel[q] = (to_host & bit[q]) ? 1u : 0u;
            if ((want & bit[q]) && dev_of_dst[q]) { sel[q] = 2u; caller_dev[q] = dev_of_dst[q]; caller_host[q] = host_of_dst[q]; dmask |= bit[q]; }
            else if (h->early && (q != 0 || h->early_grad) && !(want & bit[q]) && h->seen_host[q]) {
                // early outputs (opt-in): what this call does NOT ask for goes straight into the registered caller array an earlier call
                // passed for it — IPOPT's eval_g / eval_grad_f / eval_jac_g at this x then find their values in place
                double* dev = device_address_of(h->seen_host[q], bytes[q]);
                if (!dev) { h->seen_host[q] = nullptr; continue; }   // (unregistered since)
                sel[q] = 2u; caller_dev[q] = dev; caller_host[q] = h->seen_host[q];
                h->early_host[q] = h->seen_host[q];
                h->early_mask |= bit[q];
            }
        }
        // arrays this handle registered by itself are verified at every use: a word of the call's own is written into the first and
        // the last entry; the kernel overwrites both (every entry of every output is written by every evaluation) — unless the mapping
        // went stale (the caller freed the array and the address now belongs to other pages): then the words are still there
        // The Jacobian into a HOST destination: its constant entries (43 % of them at 100 knots) are put there once — the pinned block
        // here, a caller array at its first use, spot-checked at every later one — and the launch stores the entries that depend on x only.
        // (Only in the varying-first order of a block, where the varying entries are ONE run per knot: skipping the constants of a
        //  CCS-ordered block leaves fragments of one to four doubles on the link — measured SLOWER than storing everything, 59.7
        //  against 55.0 us per 100-knot call, profiles/r04_host_path.txt.)
        bool vary_only = false;
        if (h->skip_const && h->vary_ok && h->L.vary_first && h->L.nconst_total > 0 && sel[2] != 0u) {   // (host destinations: the varying RUN of a varying-first block only)
            if (sel[2] == 1u) {
                if (h->pinned_const_gen != h->param_gen) { constants_fill(h, h->h_jac); h->pinned_const_gen = h->param_gen; }
            } else constants_ensure(h, caller_host[2]);
            vary_only = true;
        }
        u64 sentinel = 0;
        bool checked[3] = {false, false, false};
        size_t w_first[3] = {h->out_first[0], h->out_first[1], h->out_first[2]}, w_last[3] = {h->out_last[0], h->out_last[1], h->out_last[2]};   // (words the launch writes)
        if (vary_only) { w_first[2] = h->jac_first_vary; w_last[2] = h->jac_last_vary; }
        for (int q = 0; q < 3; ++q)
            if (sel[q] == 2u && auto_owns(h, caller_host[q])) {
                if (!sentinel) sentinel = 0x7FF8C0DE00000000ull | (++h->sentinel_salt & 0xFFFFFFFFull);   // (a quiet NaN no evaluation produces)
                u64* w = reinterpret_cast<u64*>(caller_host[q]);
                __atomic_store_n(&w[w_first[q]], sentinel, __ATOMIC_RELAXED);
                __atomic_store_n(&w[w_last[q]], sentinel, __ATOMIC_RELAXED);
                checked[q] = true;
            }
        const bool f_host = (to_host & HIPNLP_WANT_F) != 0;
        int rc;
        std::chrono::steady_clock::time_point t2, t3;
        if (h->multi) {   // every shard's launch, every shard's stores into the same arrays, one wait for all (multi_evaluate)
            double* const pinned[3] = {h->hd_grad, h->hd_g, h->hd_jac};
            double* o[3];
            for (int q = 0; q < 3; ++q) o[q] = sel[q] == 2u ? caller_dev[q] : (sel[q] == 1u ? pinned[q] : nullptr);
            t2 = t1;
            rc = multi_evaluate(h, o, vary_only, &t2);
            if (rc != HIPNLP_OK) return rc;
            h->x_staged = true;
            t3 = std::chrono::steady_clock::now();
            h->host_us[1] = std::chrono::duration<double, std::micro>(t2 - t1).count();
            h->host_us[2] = std::chrono::duration<double, std::micro>(t3 - t2).count();
        } else {   // launch + synchronise
            const double* xsrc = h->hd_x;
            if (!h->x_zero_copy) {
                HIP_TRY(h, hipMemcpyAsync(h->d_x, h->h_x, B * n * sizeof(double), hipMemcpyHostToDevice, h->stream));
                xsrc = h->d_x;
            }
            double* const pinned[3] = {h->hd_grad, h->hd_g, h->hd_jac};
            double* const hbm[3] = {h->d_grad, h->d_g, h->d_jac};
            double* o[3];
            for (int q = 0; q < 3; ++q) o[q] = sel[q] == 2u ? caller_dev[q] : (sel[q] == 1u ? pinned[q] : hbm[q]);
            rc = launch(h, xsrc, f_host ? h->hd_f : h->d_f, o[0], o[1], o[2], h->stream, nullptr, false, h->time_host, true, nullptr, 0, 0, vary_only);
            if (rc != HIPNLP_OK) return rc;
            h->x_staged = true;   // (the pinned copy — and the device copy, where the launch reads that — hold this x: hipnlp_eval_hess_at, new_x = 0)
            t2 = std::chrono::steady_clock::now();
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            t3 = std::chrono::steady_clock::now();
            h->host_us[1] = std::chrono::duration<double, std::micro>(t2 - t1).count();
            h->host_us[2] = std::chrono::duration<double, std::micro>(t3 - t2).count();
        }
        h->host_us[0] = std::chrono::duration<double, std::micro>(t1 - t0).count();
        // verification of the auto-registered arrays
        bool stale = false;
        for (int q = 0; q < 3; ++q)
            if (checked[q]) {
                const u64* w = reinterpret_cast<const u64*>(caller_host[q]);
                if (__atomic_load_n(&w[w_first[q]], __ATOMIC_RELAXED) == sentinel || __atomic_load_n(&w[w_last[q]], __ATOMIC_RELAXED) == sentinel) {
                    (void)drop_stale_range(h, caller_host[q]);
                    h->no_auto[q] = caller_host[q];
                    if (h->seen_host[q] == caller_host[q]) h->seen_host[q] = nullptr;
                    h->auto_fallbacks++;
                    stale = true;
                }
            }
        if (stale) {   // evaluate again; the dropped arrays now go through the pinned block (the caller's pointers are looked up afresh)
            HostDest again = dst;
            if (!device_address_of(dst.grad_host, bytes[0])) { again.grad = nullptr; again.grad_host = nullptr; }
            if (!device_address_of(dst.g_host, bytes[1])) { again.g = nullptr; again.g_host = nullptr; }
            if (!device_address_of(dst.jac_host, bytes[2])) { again.jac = nullptr; again.jac_host = nullptr; }
            return host_evaluate(h, h->h_x, 1, want, again, direct);
        }
        h->have_result = true;
        h->seq_result = h->seq;
        h->on_host = (f_host || h->multi ? HIPNLP_WANT_F : 0u);   // (a multi-device handle sums the cost on the host: always there)
        for (int q = 0; q < 3; ++q) if (sel[q] == 1u) h->on_host |= bit[q];
        h->gone = dmask | h->early_mask;
        if (h->multi && sel[1] == 0u) h->gone |= HIPNLP_WANT_G;   // (left in the shards' HBM, scattered over the constraint blocks: asked for later, evaluated again)
        if (direct) *direct = dmask;
    }
    if (h->early_mask && !(new_x)) {   // a cached request for an output that already sits in the caller's own array
        double* const asked[3] = {dst.grad_host, dst.g_host, dst.jac_host};
        for (int q = 0; q < 3; ++q)
            if ((want & bit[q]) && (h->early_mask & bit[q]) && asked[q] == h->early_host[q] && direct) *direct |= bit[q];
    }
    unsigned missing = want & HIPNLP_WANT_ALL & ~h->on_host & ~(direct ? *direct : 0u);
    if (missing & h->gone) {   // stored into a caller array by the evaluation and now asked for again: evaluate again (rare)
        h->have_result = false;
        return host_evaluate(h, h->h_x, 1, want, dst, direct);
    }
    if (missing) {
        // Still in HBM: one transfer each — straight into the caller's array when that is registered (page-locked), else into the
        // pinned block.  The Jacobian into a registered array whose blocks list the varying entries first: a small kernel stores the
        // varying run of every knot block (the constants are in place: constants_ensure), 628 KB instead of 1.1 MB at 100 knots.
        // A copy into an array some handle registered by itself is verified with the sentinel words, like the direct stores of a new
        // evaluation: a stale mapping is dropped and the output served through the pinned block.
        HIP_TRY(h, hipSetDevice(h->dev));
        unsigned to_caller = 0;
        if ((missing & HIPNLP_WANT_F) && !h->multi) HIP_TRY(h, hipMemcpyAsync(h->h_f, h->d_f, B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        double* const caller[3] = {dst.grad_host, dst.g_host, dst.jac_host};
        double* const caller_dev[3] = {dst.grad, dst.g, dst.jac};
        double* const pinned[3] = {h->h_grad, h->h_g, h->h_jac};
        const double* const hbm[3] = {h->d_grad, h->d_g, h->d_jac};
        u64 sentinel = 0;
        bool checked[3] = {false, false, false};
        size_t w_first[3] = {0, 0, 0}, w_last[3] = {bytes[0] / 8 - 1, bytes[1] / 8 - 1, bytes[2] / 8 - 1};   // (whole-array copies)
        for (int q = 0; q < 3; ++q) {
            if (!(missing & bit[q])) continue;
            const bool vary_run = q == 2 && caller[q] && caller_dev[q] && h->skip_const && h->vary_ok && h->L.vary_first && h->L.nconst_total > 0;
            if (vary_run) { constants_ensure(h, caller[q]); w_first[q] = h->jac_first_vary; w_last[q] = h->jac_last_vary; }
            if (caller[q]) {
                to_caller |= bit[q];
                if (auto_owns(h, caller[q])) {
                    if (!sentinel) sentinel = 0x7FF8C0DE00000000ull | (++h->sentinel_salt & 0xFFFFFFFFull);
                    u64* w = reinterpret_cast<u64*>(caller[q]);
                    __atomic_store_n(&w[w_first[q]], sentinel, __ATOMIC_RELAXED);
                    __atomic_store_n(&w[w_last[q]], sentinel, __ATOMIC_RELAXED);
                    checked[q] = true;
                }
            }
            if (h->multi) {
                const int rc = multi_fetch(h, q, caller[q] ? caller[q] : pinned[q], caller_dev[q], vary_run);
                if (rc != HIPNLP_OK) return rc;
            } else if (vary_run) {
                const Layout& L = h->L;
                hipLaunchKernelGGL(hipnlp_fetch_vary_kernel, dim3(unsigned(h->nk), unsigned(h->batch)), dim3(256), 0, h->stream, hbm[q], caller_dev[q], h->kb, L.N,
                                   int64_t(L.nnz), L.nnz_v[VAR_FIRST], L.nnz_v[VAR_INTERIOR], L.nvary_v[VAR_FIRST], L.nvary_v[VAR_INTERIOR], L.nvary_v[VAR_LAST]);
                HIP_TRY(h, hipGetLastError());
            } else HIP_TRY(h, hipMemcpyAsync(caller[q] ? caller[q] : pinned[q], hbm[q], bytes[q], hipMemcpyDeviceToHost, h->stream));
        }
        if (h->multi) { const int rc = multi_sync(h); if (rc != HIPNLP_OK) return rc; }
        else HIP_TRY(h, hipStreamSynchronize(h->stream));
        for (int q = 0; q < 3; ++q)
            if (checked[q]) {
                const u64* w = reinterpret_cast<const u64*>(caller[q]);
                if (__atomic_load_n(&w[w_first[q]], __ATOMIC_RELAXED) == sentinel || __atomic_load_n(&w[w_last[q]], __ATOMIC_RELAXED) == sentinel) {
                    (void)drop_stale_range(h, caller[q]);
                    h->no_auto[q] = caller[q];
                    h->auto_fallbacks++;
                    if (h->multi) {
                        int rc = multi_fetch(h, q, pinned[q], nullptr, false);
                        if (rc == HIPNLP_OK) rc = multi_sync(h);
                        if (rc != HIPNLP_OK) return rc;
                    } else HIP_TRY(h, hipnlp_internal_memcpy(pinned[q], hbm[q], bytes[q], hipMemcpyDeviceToHost));
                    to_caller &= ~bit[q];   // (hipnlp_eval copies it out of the pinned block)
                }
            }
        h->on_host |= missing & ~to_caller;   // (an output copied into a caller array stays in HBM: a later request copies again)
        if (direct) *direct |= to_caller;
    }
    return HIPNLP_OK;
}
static int host_numeric_status(hipnlp_handle* h) {
    for (int b = 0; b < h->batch; ++b)
        if (h->h_flag[b] == h->seq_result) { h->err = "non-finite value produced by the evaluation"; return HIPNLP_E_NUMERIC; }
    return HIPNLP_OK;
}

int hipnlp_eval(hipnlp_handle* h, const double* x, int new_x, double* f, double* grad_f, double* g, double* jac) {
    if (!h || !x) return HIPNLP_E_INVALID;
    if (!h->params_set) { h->err = "parameters not set (hipnlp_set_params)"; return HIPNLP_E_PARAMS; }
    const size_t B = size_t(h->batch), n = size_t(h->L.n), m = size_t(h->L.m), nnz = size_t(h->L.nnz);
    const unsigned want = (f ? HIPNLP_WANT_F : 0u) | (grad_f ? HIPNLP_WANT_GRAD : 0u) | (g ? HIPNLP_WANT_G : 0u) | (jac ? HIPNLP_WANT_JAC : 0u);
    HostDest dst;   // caller arrays inside a registered range (hipnlp_host_register, or registered by the handle itself): direct kernel outputs
    h->in_call[0] = grad_f; h->in_call[1] = g; h->in_call[2] = jac; h->in_call[3] = nullptr;
    dst.grad = caller_array_address(h, 0, grad_f, B * n * sizeof(double));
    dst.g = caller_array_address(h, 1, g, B * m * sizeof(double));
    dst.jac = caller_array_address(h, 2, jac, B * nnz * sizeof(double));
    dst.grad_host = dst.grad ? grad_f : nullptr;
    dst.g_host = dst.g ? g : nullptr;
    dst.jac_host = dst.jac ? jac : nullptr;
    unsigned direct = 0;
    const int rc = host_evaluate(h, x, new_x, want, dst, &direct);
    h->in_call[0] = h->in_call[1] = h->in_call[2] = nullptr;
    if (rc != HIPNLP_OK) return rc;
    // (looked up again: the evaluation may have dropped an array whose mapping had gone stale)
    if (dst.grad_host && device_address_of(grad_f, B * n * sizeof(double))) h->seen_host[0] = dst.grad_host;
    if (dst.g_host && device_address_of(g, B * m * sizeof(double))) h->seen_host[1] = dst.g_host;
    if (dst.jac_host && device_address_of(jac, B * nnz * sizeof(double))) h->seen_host[2] = dst.jac_host;
    const auto t0 = std::chrono::steady_clock::now();
    if (f) std::memcpy(f, h->h_f, B * sizeof(double));
    if (grad_f && !(direct & HIPNLP_WANT_GRAD)) std::memcpy(grad_f, h->h_grad, B * n * sizeof(double));
    if (g && !(direct & HIPNLP_WANT_G)) std::memcpy(g, h->h_g, B * m * sizeof(double));
    if (jac && !(direct & HIPNLP_WANT_JAC)) std::memcpy(jac, h->h_jac, B * nnz * sizeof(double));
    h->host_us[3] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    return host_numeric_status(h);   // (the outputs are copied first: the caller may want to look at the NaNs, as IPOPT does)
}

int hipnlp_eval_pinned(hipnlp_handle* h, const double* x, int new_x, unsigned want, const double** f, const double** grad_f, const double** g,
                       const double** jac) {
    if (!h || !x) return HIPNLP_E_INVALID;
    if (!h->params_set) { h->err = "parameters not set (hipnlp_set_params)"; return HIPNLP_E_PARAMS; }
    const int rc = host_evaluate(h, x, new_x, want);
    if (rc != HIPNLP_OK) return rc;
    if (f) *f = (want & HIPNLP_WANT_F) ? h->h_f : nullptr;
    if (grad_f) *grad_f = (want & HIPNLP_WANT_GRAD) ? h->h_grad : nullptr;
    if (g) *g = (want & HIPNLP_WANT_G) ? h->h_g : nullptr;
    if (jac) *jac = (want & HIPNLP_WANT_JAC) ? h->h_jac : nullptr;
    return host_numeric_status(h);
}

int hipnlp_set_auto_register(hipnlp_handle* h, int on) {
    if (!h) return HIPNLP_E_INVALID;
    h->auto_reg = on != 0;
    if (!h->auto_reg) {
        auto_unregister_all(h);
        for (int q = 0; q < 4; ++q) { h->last_seen[q] = nullptr; h->no_auto[q] = nullptr; }
    }
    return HIPNLP_OK;
}

int hipnlp_host_stats(const hipnlp_handle* h, long* out /*[8]*/) {
    if (!h || !out) return HIPNLP_E_INVALID;
    out[0] = h->auto_registered; out[1] = h->auto_fallbacks; out[2] = long(auto_count(h));
    out[3] = long(h->seq); out[4] = h->const_fills + h->dev_const_fills; out[5] = h->const_refills; out[6] = long(h->L.nconst_total);
    int32_t healed = 0;   // (wave slices of constants a VARY kernel put back into a device buffer; the copy waits for the device)
    if (h->d_healed && hipSetDevice(h->dev) == hipSuccess && hipnlp_internal_memcpy(&healed, h->d_healed, sizeof healed, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); healed = -1; }
    out[7] = healed;
    return HIPNLP_OK;
}

int hipnlp_set_early_outputs(hipnlp_handle* h, int on) {
    if (!h || on < 0 || on > 2) return HIPNLP_E_INVALID;
    h->early = on != 0;
    h->early_grad = on == 2;   // grad f only on explicit request: IPOPT's adapter hands eval_grad_f the storage of ITS OWN gradient vector
    if (!h->early) { h->early_mask = 0; }
    return HIPNLP_OK;
}

int hipnlp_set_hessian_early_run(hipnlp_handle* h, int mode) {
    if (!h || mode < -1 || mode > 1) return HIPNLP_E_INVALID;
    h->hess_early_mode = mode;
    h->hess_early_choice = -1; h->hess_tune_calls = 0; h->hess_tune_best[0] = h->hess_tune_best[1] = 1e30;   // (auto: decided afresh)
    h->hess_tune_us[0].clear(); h->hess_tune_us[1].clear();
    h->hess_early_why = mode >= 0 ? HESS_WHY_FORCED : HESS_WHY_UNDECIDED;
    return HIPNLP_OK;
}
const char* hipnlp_hessian_early_run_reason(const hipnlp_handle* h) {
    if (!h) return "";
    switch (h->hess_early_mode >= 0 ? HESS_WHY_FORCED : h->hess_early_why) {
        case HESS_WHY_FORCED: return "set by the caller (hipnlp_set_hessian_early_run)";
        case HESS_WHY_NOT_ARMABLE: return "off: this handle's Hessian launches cannot send a run ahead (compact layout of long launches, or no early run in the pattern)";
        case HESS_WHY_NUMA_SAME: return "on: the calling thread runs on the card's NUMA node";
        case HESS_WHY_NUMA_OTHER: return "off: the calling thread runs on another NUMA node than the card's";
        case HESS_WHY_MEASURED: return "measured on the handle's first calls (NUMA nodes not known): the lower median of nine calls of each kind";
        default: return "not decided yet (decided by the first hipnlp_eval_hess call)";
    }
}
int hipnlp_get_hessian_early_run(const hipnlp_handle* h, int* mode, int* chosen, double* us_off, double* us_on) {
    if (!h) return HIPNLP_E_INVALID;
    if (mode) *mode = h->hess_early_mode;
    if (chosen) *chosen = h->hess_early_mode >= 0 ? h->hess_early_mode : h->hess_early_choice;
    if (us_off) *us_off = h->hess_tune_best[0] < 1e29 ? h->hess_tune_best[0] : 0.0;
    if (us_on) *us_on = h->hess_tune_best[1] < 1e29 ? h->hess_tune_best[1] : 0.0;
    return HIPNLP_OK;
}

int hipnlp_set_constant_jacobian(hipnlp_handle* h, int on) {
    if (!h) return HIPNLP_E_INVALID;
    h->skip_const = on != 0;
    return HIPNLP_OK;
}
int hipnlp_forget_jac_destination(hipnlp_handle* h, const void* p) {
    if (!h) return HIPNLP_E_INVALID;
    int n = 0;
    for (int i = 0; i < 8; ++i) {
        if (h->dfilled[i].dev && (!p || h->dfilled[i].dev == p)) { h->dfilled[i] = {nullptr, 0, false, 0u}; ++n; }
        if (h->cfilled[i].host && (!p || h->cfilled[i].host == p)) { h->cfilled[i] = {nullptr, 0, 0}; ++n; }
    }
    return n;
}
int hipnlp_jac_constant_mask(const hipnlp_handle* h, unsigned char* mask) {
    if (!h || !mask) return HIPNLP_E_INVALID;
    h->L.constant_mask(mask);
    return HIPNLP_OK;
}

int hipnlp_set_prefetch(hipnlp_handle* h, unsigned mask) {
    if (!h || (mask & ~HIPNLP_WANT_ALL)) return HIPNLP_E_INVALID;
    h->prefetch = mask;
    return HIPNLP_OK;
}

int hipnlp_host_breakdown(const hipnlp_handle* h, double* us) {
    if (!h || !us) return HIPNLP_E_INVALID;
    for (int i = 0; i < 4; ++i) us[i] = h->host_us[i];
    return HIPNLP_OK;
}

int hipnlp_set_host_timing(hipnlp_handle* h, int on) {
    if (!h) return HIPNLP_E_INVALID;
    h->time_host = on != 0;
    return HIPNLP_OK;
}

int hipnlp_host_register(void* p, size_t bytes, void** dev_ptr) {
    if (!p || !bytes) return HIPNLP_E_INVALID;
    {   // an array a handle registered by itself: the caller takes over — with a FRESH registration: the handle's one may be of pages
        // that were freed since (its stores are verified, the caller's are not), so its device address is not handed on
        bool taken = false;
        {
            std::lock_guard<std::mutex> lock(g_ranges_mutex);
            for (HostRange& r : g_ranges) {
                if (r.host != static_cast<char*>(p)) continue;
                if (r.owner == nullptr && r.bytes >= bytes) { if (dev_ptr) *dev_ptr = r.dev; return HIPNLP_OK; }   // (registered by the caller before: as it is)
                taken = true;
            }
        }
        if (taken) (void)hipnlp_host_unregister(p);
    }
    {   // Ranges that HANDLES registered by themselves and that overlap [p, p + bytes) describe memory that has changed hands — an
        // array was freed and this one took its addresses: they go before the new registration is made (their owners look their
        // arrays up again at every use and simply find nothing).
        std::vector<char*> stale;
        {
            std::lock_guard<std::mutex> lock(g_ranges_mutex);
            const char* q = static_cast<const char*>(p);
            for (const HostRange& r : g_ranges)
                if (r.owner != nullptr && q < r.host + r.bytes && r.host < q + bytes) stale.push_back(r.host);
        }
        for (char* sp : stale) (void)hipnlp_host_unregister(sp);
    }
    if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return HIPNLP_E_NODEVICE; }
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess) { (void)hipHostUnregister(p); (void)hipGetLastError(); return HIPNLP_E_NODEVICE; }
    if (dev_ptr) *dev_ptr = d;
    std::lock_guard<std::mutex> lock(g_ranges_mutex);
    g_ranges.push_back(HostRange{static_cast<char*>(p), bytes, static_cast<char*>(d), nullptr, ++g_range_ids});
    return HIPNLP_OK;
}
int hipnlp_host_release_auto_ranges(void) {
    std::vector<char*> all;
    {
        std::lock_guard<std::mutex> lock(g_ranges_mutex);
        for (const HostRange& r : g_ranges) if (r.owner != nullptr) all.push_back(r.host);
    }
    for (char* q : all) (void)hipnlp_host_unregister(q);
    return int(all.size());
}
// A copy between plain host memory and the device that the runtime refuses with "invalid argument": the host buffer (heap memory of the
// library or of the caller) may lie inside a range some handle registered by itself for an array that has been FREED since — the
// allocator handed its addresses out again, the runtime still believes the old registration.  Those registrations are conveniences
// the handles re-create whenever they see their arrays again: all are released and the copy is tried once more.
extern "C" hipError_t hipnlp_internal_memcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    hipError_t e = hipMemcpy(dst, src, bytes, kind);
    if (e == hipErrorInvalidValue) {
        (void)hipGetLastError();
        if (hipnlp_host_release_auto_ranges() > 0) e = hipMemcpy(dst, src, bytes, kind);
    }
    return e;
}
int hipnlp_host_unregister(void* p) {
    if (!p) return HIPNLP_E_INVALID;
    bool known = false;
    {
        std::lock_guard<std::mutex> lock(g_ranges_mutex);
        for (size_t i = 0; i < g_ranges.size(); ++i)
            if (g_ranges[i].host == static_cast<char*>(p)) { g_ranges.erase(g_ranges.begin() + long(i)); known = true; break; }
    }
    if (!known) return HIPNLP_E_INVALID;
    // (the error of an unregistration the runtime no longer knows about must not stay behind as the process's "last error")
    if (hipHostUnregister(p) != hipSuccess) { (void)hipGetLastError(); return HIPNLP_E_INVALID; }
    return HIPNLP_OK;
}

int hipnlp_cost_terms(hipnlp_handle* h, double* values) {
    if (!h || !values) return HIPNLP_E_INVALID;
    if (h->is_front && !h->have_result) { h->err = "hipnlp_cost_terms: no evaluation yet (hipnlp_eval)"; return HIPNLP_E_INVALID; }   // (summed on the host by every evaluation)
    if (!h->have_result) {   // a device-path evaluation: its per-term costs are in device memory (the host path stores them straight to the pinned block)
        HIP_TRY(h, hipSetDevice(h->dev));
        HIP_TRY(h, hipDeviceSynchronize());   // (the launch may sit on a stream of the caller's: hipnlp_eval_device(..., stream))
        HIP_TRY(h, hipnlp_internal_memcpy(h->h_cost_terms, h->d_cost_terms, size_t(h->batch) * NCT * sizeof(double), hipMemcpyDeviceToHost));
    }
    std::memcpy(values, h->h_cost_terms, size_t(h->batch) * NCT * sizeof(double));
    return HIPNLP_OK;
}

const char* hipnlp_cost_term_name(int i) {
    static const char* names[NCT] = {"swing_height_regularization", "u_v_regularization", "f_dot_regularization", "com_velocity_error",
                                     "frame_quaternion_error", "base_quaternion_error", "base_quaternion_velocity_error", "joint_positions_error",
                                     "contacts_centroid_cost", "f_regularization", "yaw_regularization", "final_and_periodicity"};
    return (i >= 0 && i < NCT) ? names[i] : "";
}

int hipnlp_num_row_blocks(const hipnlp_handle* h) { return h ? int(h->L.blocks.size()) : HIPNLP_E_INVALID; }

int hipnlp_row_block(const hipnlp_handle* h, int i, const char** name, int32_t* first_row, int32_t* rows_per_knot, int32_t* first_knot, int32_t* n_knots) {
    if (!h || i < 0 || i >= int(h->L.blocks.size())) return HIPNLP_E_INVALID;
    const RowBlock& b = h->L.blocks[size_t(i)];
    if (name) *name = b.name.c_str();
    if (first_row) *first_row = b.first_row;
    if (rows_per_knot) *rows_per_knot = b.rows;
    if (first_knot) *first_knot = b.k0;
    if (n_knots) *n_knots = b.nk;
    return HIPNLP_OK;
}

int hipnlp_last_kernel_ms(hipnlp_handle* h, float* ms) {
    if (!h || !ms) return HIPNLP_E_INVALID;
    NOT_FRONT(h, "hipnlp_last_kernel_ms");
    if (!h->timing_valid) { h->err = "no timed evaluation yet (hipnlp_eval, or a launch sampled by hipnlp_profile_begin)"; return HIPNLP_E_INVALID; }
    HIP_TRY(h, hipSetDevice(h->dev));
    HIP_TRY(h, hipEventSynchronize(h->last_e2));
    HIP_TRY(h, hipEventElapsedTime(ms, h->last_e0, h->last_e2));
    return HIPNLP_OK;
}

int hipnlp_profile_begin(hipnlp_handle* h, int max_launches, int stride) {
    if (!h || max_launches < 0 || stride < 1) return HIPNLP_E_INVALID;
    NOT_FRONT(h, "hipnlp_profile_begin");
    HIP_TRY(h, hipSetDevice(h->dev));
    while (int(h->prof_ev.size()) < 3 * max_launches) {
        hipEvent_t e;
        HIP_TRY(h, hipEventCreate(&e));
        h->prof_ev.push_back(e);
    }
    h->prof_cap = max_launches;
    h->prof_n = 0;
    h->prof_stride = stride;
    h->prof_seen = 0;
    h->prof_run = 0;
    h->prof_open = false;
    h->timing_valid = false;
    return HIPNLP_OK;
}
int hipnlp_profile_begin_runs(hipnlp_handle* h, int max_runs, int run_len) {
    if (!h || run_len < 1) return HIPNLP_E_INVALID;
    const int rc = hipnlp_profile_begin(h, max_runs, 1);
    if (rc != HIPNLP_OK) return rc;
    h->prof_run = run_len;
    return HIPNLP_OK;
}

int hipnlp_reassemble(const double* gathered_dev, const int64_t* src_dev, double* out_dev, int64_t count, int world, int64_t shard_len,
                      double* f_out_dev, void* stream) {
    if (!gathered_dev || !src_dev || !out_dev || count < 0 || world < 1) return HIPNLP_E_INVALID;
    const int64_t blocks = (count + 255) / 256;
    const unsigned grid = unsigned(blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks));
    hipLaunchKernelGGL(hipnlp_reassemble_kernel, dim3(grid), dim3(256), 0, hipStream_t(stream), gathered_dev, src_dev, out_dev, count, world, shard_len, f_out_dev);
    return hipGetLastError() == hipSuccess ? HIPNLP_OK : HIPNLP_E_NODEVICE;
}

int hipnlp_reassemble_scatter(const double* gathered_dev, const int64_t* src_dev, const int64_t* dst_dev, double* out_dev, int64_t count, int world, int64_t shard_len,
                              double* f_out_dev, void* stream) {
    if (!gathered_dev || !src_dev || !dst_dev || !out_dev || count < 0 || world < 1) return HIPNLP_E_INVALID;
    const int64_t blocks = (count + 255) / 256;
    const unsigned grid = unsigned(blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks));
    hipLaunchKernelGGL(hipnlp_reassemble_scatter_kernel, dim3(grid), dim3(256), 0, hipStream_t(stream), gathered_dev, src_dev, dst_dev, out_dev, count, world, shard_len, f_out_dev);
    return hipGetLastError() == hipSuccess ? HIPNLP_OK : HIPNLP_E_NODEVICE;
}

// ---- peer exchange without a collective (include/hipnlp.h) ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void hipnlp_peer_push_kernel(const double* __restrict__ shard, const int64_t* __restrict__ dst, int64_t count,
                                                               double* const* __restrict__ peer_out, int world) {
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += stride) {
        const int64_t d = dst[i];
        if (d < 0) continue;
        const double v = shard[i];
        for (int r = 0; r < world; ++r) peer_out[r][d] = v;   // (consecutive lanes: consecutive addresses on every link)
    }
}
// A rank whose own status word is raised (a wait of ITS step gave up: what it pushed may have gone into a buffer that was still being
// read) signals with the poison bit: the receiving wait passes and poisons the step there too.
constexpr unsigned long long PEER_POISON = 1ull << 63;
__global__ void hipnlp_peer_signal_kernel(unsigned long long* const* peer_flags, int world, int rank, unsigned long long seq, const int* status) {
    __threadfence_system();   // (the push kernel ended before this one started: its stores are performed; ordered before the flags system-wide)
    const int r = threadIdx.x;
    const unsigned long long v = (status && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) ? (seq | PEER_POISON) : seq;
    if (r < world) __hip_atomic_store(peer_flags[r] + rank, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// One workgroup of 64 x 4 threads.  A wait that gives up is STICKY: *status is only ever raised here (the host clears it), and every
// output of the step is poisoned — f as before, and grad / jac / g as well (out[0 .. f_off)): a step whose pushes may be partial must
// not look like an ordinary evaluation with a bad cost.
__global__ __launch_bounds__(256) void hipnlp_peer_wait_kernel(const unsigned long long* flags, int world, unsigned long long seq, double* out, int64_t f_off, int* status) {
    __shared__ int s_late;
    const int r = threadIdx.x;
    if (r == 0) s_late = 0;
    __syncthreads();
    if (r < world) {
        int spins = 0;
        unsigned long long fv;
        while ((fv = __hip_atomic_load(flags + r, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM)) < seq) {   // (a poisoned flag is >= any step number)
            if (++spins > (1 << 20)) { s_late = 1; break; }   // (a rank that died must not hang this device)
            __builtin_amdgcn_s_sleep(8);
        }
        if (fv & PEER_POISON) s_late = 1;   // the sender's own step had failed
    }
    __syncthreads();
    const int late = s_late;
    __threadfence_system();
    if (late) for (int64_t i = r; i < f_off; i += blockDim.x) out[i] = __builtin_nan("");
    if (r == 0) {
        double f = 0.0;
        for (int q = 0; q < world; ++q) f += __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(out + f_off + q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        out[f_off + world] = late ? __builtin_nan("") : f;
        if (late) atomicOr(status, 1);
    }
}

int hipnlp_ipc_alloc(size_t bytes, int device, void** dev_ptr, void* handle_out) {
    if (!dev_ptr || !handle_out || bytes == 0) return HIPNLP_E_INVALID;
    static_assert(sizeof(hipIpcMemHandle_t) == HIPNLP_IPC_HANDLE_BYTES, "IPC handle size");
    if (hipSetDevice(device) != hipSuccess) return HIPNLP_E_NODEVICE;
    void* p = nullptr;
    // uncached / fine-grained: a peer's stores must not meet stale lines in this device's L2 (what RCCL does for its own buffers)
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); return HIPNLP_E_NODEVICE; }
    }
    if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(p); return HIPNLP_E_NODEVICE; }
    hipIpcMemHandle_t hnd;
    if (hipIpcGetMemHandle(&hnd, p) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); return HIPNLP_E_NODEVICE; }
    std::memcpy(handle_out, &hnd, sizeof hnd);
    *dev_ptr = p;
    return HIPNLP_OK;
}
int hipnlp_ipc_open(const void* handle, int device, void** dev_ptr) {
    if (!handle || !dev_ptr) return HIPNLP_E_INVALID;
    if (hipSetDevice(device) != hipSuccess) return HIPNLP_E_NODEVICE;
    hipIpcMemHandle_t hnd;
    std::memcpy(&hnd, handle, sizeof hnd);
    void* p = nullptr;
    if (hipIpcOpenMemHandle(&p, hnd, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return HIPNLP_E_NODEVICE; }
    *dev_ptr = p;
    return HIPNLP_OK;
}
int hipnlp_ipc_close(void* dev_ptr) { return dev_ptr && hipIpcCloseMemHandle(dev_ptr) == hipSuccess ? HIPNLP_OK : HIPNLP_E_INVALID; }
int hipnlp_ipc_free(void* dev_ptr) { return dev_ptr && hipFree(dev_ptr) == hipSuccess ? HIPNLP_OK : HIPNLP_E_INVALID; }

int hipnlp_peer_push(const double* shard_dev, const int64_t* dst_dev, int64_t count, double* const* peer_out_dev, int world, void* stream) {
    if (!shard_dev || !dst_dev || !peer_out_dev || count < 0 || world < 1) return HIPNLP_E_INVALID;
    const int64_t blocks = (count + 255) / 256;
    const unsigned grid = unsigned(blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks));
    hipLaunchKernelGGL(hipnlp_peer_push_kernel, dim3(grid), dim3(256), 0, hipStream_t(stream), shard_dev, dst_dev, count, peer_out_dev, world);
    return hipGetLastError() == hipSuccess ? HIPNLP_OK : HIPNLP_E_NODEVICE;
}
int hipnlp_peer_signal_checked(unsigned long long* const* peer_flags_dev, int world, int rank, unsigned long long seq, const int* status_dev, void* stream) {
    // (world: flag arrays written to — every rank's, or one rank's: gather_to_root —; rank: this rank's slot in them)
    if (!peer_flags_dev || world < 1 || world > 64 || rank < 0 || rank >= 64 || (seq & PEER_POISON)) return HIPNLP_E_INVALID;
    hipLaunchKernelGGL(hipnlp_peer_signal_kernel, dim3(1), dim3(64), 0, hipStream_t(stream), peer_flags_dev, world, rank, seq, status_dev);
    return hipGetLastError() == hipSuccess ? HIPNLP_OK : HIPNLP_E_NODEVICE;
}
int hipnlp_peer_signal(unsigned long long* const* peer_flags_dev, int world, int rank, unsigned long long seq, void* stream) {
    return hipnlp_peer_signal_checked(peer_flags_dev, world, rank, seq, nullptr, stream);
}
int hipnlp_peer_wait(const unsigned long long* flags_dev, int world, unsigned long long seq, double* out_dev, int64_t f_off, int* status_dev, void* stream) {
    if (!flags_dev || !out_dev || !status_dev || world < 1 || world > 64) return HIPNLP_E_INVALID;
    hipLaunchKernelGGL(hipnlp_peer_wait_kernel, dim3(1), dim3(256), 0, hipStream_t(stream), flags_dev, world, seq, out_dev, f_off, status_dev);
    return hipGetLastError() == hipSuccess ? HIPNLP_OK : HIPNLP_E_NODEVICE;
}

int hipnlp_kernels_per_eval(const hipnlp_handle* h) { return h ? (h->fused || h->is_front ? 1 : 2) : HIPNLP_E_INVALID; }   // (a multi-device handle: one knot launch per shard, the cost summed on the host)

int hipnlp_profile_end(hipnlp_handle* h, double* mean_knot_kernel_ms, double* mean_launch_ms, int* count) {
    if (!h) return HIPNLP_E_INVALID;
    NOT_FRONT(h, "hipnlp_profile_end");
    HIP_TRY(h, hipSetDevice(h->dev));
    double a = 0.0, b = 0.0;
    for (int i = 0; i < h->prof_n; ++i) {
        float m1 = 0.f, m2 = 0.f;
        HIP_TRY(h, hipEventSynchronize(h->prof_ev[size_t(3 * i + 2)]));
        HIP_TRY(h, hipEventElapsedTime(&m1, h->prof_ev[size_t(3 * i)], h->prof_ev[size_t(3 * i + (h->prof_run > 0 ? 2 : 1))]));
        HIP_TRY(h, hipEventElapsedTime(&m2, h->prof_ev[size_t(3 * i)], h->prof_ev[size_t(3 * i + 2)]));
        a += m1; b += m2;
    }
    const double per = double(h->prof_n) * double(h->prof_run > 0 ? h->prof_run : 1);   // launches behind the sums
    if (count) *count = int(per);
    if (mean_knot_kernel_ms) *mean_knot_kernel_ms = h->prof_n ? a / per : 0.0;
    if (mean_launch_ms) *mean_launch_ms = h->prof_n ? b / per : 0.0;
    h->prof_run = 0;
    h->prof_open = false;
    h->prof_cap = 0;
    h->prof_n = 0;
    h->timing_valid = false;
    return HIPNLP_OK;
}

#ifdef HIPNLP_STAMPS
int hipnlp_debug_stamps(hipnlp_handle* h, unsigned long long* out /*[nk*batch][4][16]*/) {
    if (!h) return HIPNLP_E_INVALID;
    NOT_FRONT(h, "hipnlp_debug_stamps");
    HIP_TRY(h, hipSetDevice(h->dev));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipnlp_internal_memcpy(out, h->d_stamps, 2 * size_t(h->nk) * size_t(h->batch) * 1024 * sizeof(unsigned long long), hipMemcpyDeviceToHost));   // (out: [2 nk batch][8][128])
    return HIPNLP_OK;
}
#endif

}  // extern "C"
