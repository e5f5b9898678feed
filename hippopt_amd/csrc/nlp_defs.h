// nlp_defs.h — shared constants of the hipnlp engine: per-knot variable offsets, the device-side
// parameter records, the kernel's NATIVE output slots and the symbolic row/column ids the
// kernel body attaches to every value it produces.
//
// The body (knot_body.h) is compiled three ways from ONE source:
//   device  : values go to LDS at native slots (ids are dead code)
//   record  : host, values ignored, ids recorded -> layout.cpp derives g rows / CCS positions
//   hostemu : tests only (tests/hostemu), values + ids on the CPU to debug the body without a GPU
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <limits.h>

#include "../../include/hipnlp.h"

#if defined(__HIPCC__)
#define HD __host__ __device__ __forceinline__
#else
#define HD inline
#endif

namespace hipnlp {

// "no row" in the slot -> row tables (g_a).  A valid g_a may be NEGATIVE (row = g_a + g_b k, and a block that starts at knot 1 with
// its first row at 0 — the lifted layout begins with one — has g_a = -g_b), so validity is a value of its own, not a sign.
constexpr int32_t G_NONE = INT32_MIN;


constexpr int NJ = HIPNLP_NJ, NL = HIPNLP_NL, NC = HIPNLP_NC, NXK = HIPNLP_NXK, NXG = HIPNLP_NXG;
constexpr int LEG_PATH = 6;    // joints between root_link and each sole frame (ergoCub topology)
constexpr int CHEST_PATH = 3;  // joints between root_link and the chest frame
constexpr int XPAD = 192;      // LDS stride of one knot record
constexpr int FK_SPLIT = 11;   // forward kinematics: joints [0, FK_SPLIT) on one wave, the rest on another (both parent-closed)
constexpr int MAX_LEAF = 6;    // leaf links of the kinematic tree (ergoCub: 2 hands + 2 feet)

// ---- per-knot variable offsets (reference creation order, tests/golden/kinodyn_structure.json) --
enum : int { V_ = 0, FD_ = 3, P_ = 6, F_ = 9, U_ = 12, PT_ = 15,
             VB_ = 120, QD_ = 123, PB_ = 127, QB_ = 130, SD_ = 134, S_ = 157, COM_ = 180, H_ = 183 };

// ---- per-knot parameter record on the device: [descriptors 24 | references 55 | pad] -----------
constexpr int PK_STRIDE = 88;   // [descriptors 24 | references 55 | sin/cos of the yaw references 8 | pad 1]
enum : int { PK_DESC = 0, PK_REF = 24, PK_YAWSC = 79,
             R_ALPHA_L = 0, R_YAW_L = 4, R_ALPHA_R = 5, R_YAW_R = 9, R_SWING = 10, R_CW = 11, R_CREF = 14,
             R_VREF = 17, R_FQ = 20, R_BQ = 24, R_BQV = 28, R_JREG = 32 };

// ---- horizon-global parameters the kernels need (bounds-only parameters stay on the host) -----
// GParamsLite: what every knot reads; GParams adds the final-state values, which only the LAST knot reads and only in `minimize`
// mode (the compact device layout leaves them in global memory)
struct alignas(16) GParamsLite {   // (16: staged into LDS in 16-byte pieces)
    double dt, kt, kbs, eps, mu, mass;
    double gravity[6];
};
struct alignas(16) GParams : GParamsLite {
    double final_rhs[105];  // final_state values in final-row order (only read in `minimize` mode)
};

// ---- kinematic tree tables (device copy of hipnlp_robot_model + derived topology) ----------------
// KinLite: the tables every phase indexes per lane (LDS on the device).  KinTables adds the blocks that are read ONCE per knot
// (joint frames: phase A only; link inertials: phase C only) or only by the first / last knot (horizon-end row tables): the compact
// device layout does not keep those in LDS for the whole program (knot_body.h, KnotScratchT<LAYOUT_COMPACT>).
struct alignas(16) KinLite {
    int32_t leg_pos[2][NJ];     // position of joint j in the root->sole path (0..5) or -1
    int32_t leg_joint[2][LEG_PATH];
    int32_t chest_pos[NJ];      // position in the root->chest path or -1
    int32_t frame_link[3];
    // tree as ancestor / descendant lists: every lane sums over its own list, no loop-carried dependence
    int32_t par_link[NJ];       // parent link of joint j
    int32_t fk_first[2];        // first useful position of the (front-padded) ancestor lists of joints [0, FK_SPLIT) / [FK_SPLIT, NJ)
    alignas(8) int8_t anc[NJ][8];  // (read as one 8-byte word) joints on the path root -> j (inclusive, in path order, j LAST), FRONT padded with NJ (identity / zero slot)
    alignas(8) int8_t desc[NL][NL];  // (read as 8-byte words) links of the subtree rooted at link i (inclusive), padded with NL (zero slot)
    int16_t ndesc[NL];          // size of that subtree
    int16_t comp_order[NL];     // links by decreasing subtree size (order of the composite tasks)
    int16_t comp_cnt[NL / 4];   // largest subtree in each group of four links of that order
    double frame_R[3][9], frame_o[3][3];
    double inv_total_mass;      // 1 / (sum of the link masses): every use of the total mass on the device is a division by it, and the first one heads the
                                // dependent chain of the derivative-column tasks (an IEEE division is ~35 dependent instructions)
};
// (one record per joint / per link: the lane that needs a joint's frame or a link's inertials forms ONE address — in the compact device
//  layouts a 64-bit global one — and reads the fields at immediate offsets; three arrays side by side cost three address chains)
//  (no padding: the full layouts stage these tables in LDS, and the Hessian kernel's 64 KB are used to the last 128 bytes)
struct JointFixRec { double R_fix[9], o_fix[3], axis[3]; };   // 120 B
struct LinkInertialsRec { double inertia[9], com[3], mass; }; // 104 B
struct JointFix { JointFixRec j[NJ]; };                                      // joint frames (phase A only)
struct LinkInertials { LinkInertialsRec l[NL]; };                            // link inertials in the link frame (phase C only)
struct EndTables {   // horizon-end rows (first / last knot only)
    int16_t fin_var[105], fin_slot[105], fin_desc[105];  // variable / slot among the 81 variable rows / descriptor index (3c+i) or -1
    int16_t per_var[84];
};
// Sloped tops of the terrain steps (SmoothTerrain.step(top_normal_direction=...), utilities/smooth_terrain.py:238-264): the top surface
// pi(q_xy) = height - (n_x q_x + n_y q_y) / n_z of step s, in world coordinates  pi = height + px[s] (p_x - o_x) + py[s] (p_y - o_y).
// Read by the lanes of a bump task only for steps whose TerrainStepK says so (step_sloped): kept OUT of KSettings, whose LDS copy in the
// compact layouts has no byte to spare (the five-per-CU smooth-terrain kernel: 31 984 of 32 000 B), behind the tables that the compact
// layouts read from global memory once per knot.
struct TerrainTops {
    double px[HIPNLP_MAX_TERRAIN_STEPS], py[HIPNLP_MAX_TERRAIN_STEPS];
};
struct KinTables : KinLite {
    EndTables en;
    JointFix jf;
    LinkInertials li;
    TerrainTops tops;
};
constexpr int JFIX_DOUBLES = NJ * 15, LINR_DOUBLES = NL * 13;
static_assert(sizeof(KinLite) % 8 == 0, "KinLite is copied in 8-byte words");
static_assert(sizeof(JointFix) == sizeof(double) * JFIX_DOUBLES && sizeof(LinkInertials) == sizeof(double) * LINR_DOUBLES, "contiguous blocks of doubles");

// ---- graph constants (hipnlp_settings, flattened for the kernel) ------------------------------------
struct TerrainStepK {   // one SmoothTerrain.step bump, pre-digested: a = ax dx + ay dy, b = bx dx + by dy (dx = p_x - ox, ...)
    double ox, oy, oz, height, ax, ay, bx, by;
    int32_t m, r;       // exponents 2*edge_sharpness (bit 16: the top of the step is sloped, TerrainTops) and 2*side_sharpness
};
constexpr int32_t STEP_SLOPED = 1 << 16;
HD constexpr int step_m(const TerrainStepK& t) { return t.m & (STEP_SLOPED - 1); }
HD constexpr bool step_sloped(const TerrainStepK& t) { return (t.m & STEP_SLOPED) != 0; }
struct alignas(16) KSettings {
    int32_t horizon, final_type, periodicity_type, joint_reg_as_coded, hdyn_x0;
    int32_t terrain, n_steps;
    TerrainStepK steps[HIPNLP_MAX_TERRAIN_STEPS];
    int32_t yaw_corner[2][3];
    double final_weight, periodicity_weight;
    double m_centroid, w_comvel[3], m_comvel, m_frameq, m_baseq, m_baseqv, w_jreg[NJ], m_jreg, m_freg, m_yaw, m_swing, m_ureg, m_fdreg;
    // static pose finder only (pose_body.h): expression types of the com / left / right point position terms and their multipliers
    int32_t pose_com_type, pose_left_type, pose_right_type, pose_pad_;
    double m_pcom, m_favg, m_preg;
};

// the read-only tables every workgroup stages into LDS, as ONE 16-byte-aligned block (copied with 16-byte loads, all in flight at once)
struct alignas(16) HeadTables {
    KinTables kt;
    KSettings ks;
};
static_assert(sizeof(HeadTables) % 16 == 0, "HeadTables is copied in 16-byte words");
static_assert(sizeof(GParams) % 8 == 0, "GParams is copied in 8-byte words");

// ---- cost terms (Output.cost_values grouping; order = hipnlp_cost_term_name) -------------------------
enum : int { CT_SWING = 0, CT_UREG, CT_FDREG, CT_COMVEL, CT_FRAMEQ, CT_BASEQ, CT_BASEQV, CT_JREG, CT_CENTROID, CT_FREG, CT_YAW, CT_ENDS, NCT };
static_assert(NCT == HIPNLP_NCOST_TERMS, "cost term count");

// =====================================================================================================
// Row ids: which constraint row a native g slot / jac slot belongs to.
// =====================================================================================================
enum RowKind : int {
    // per contact point (c = 0..7)
    RK_FDYN_IN = 0, RK_FDYN_OUT, RK_FDYN_X0, RK_PDYN_IN, RK_PDYN_OUT, RK_PDYN_X0,
    RK_PLANAR, RK_DCC, RK_HEIGHT, RK_NORMAL, RK_FRICTION, RK_UB, RK_FDB, RK_KINC,
    // global
    RK_PBDYN_IN, RK_PBDYN_OUT, RK_PBDYN_X0, RK_QBDYN_IN, RK_QBDYN_OUT, RK_QBDYN_X0,
    RK_SDYN_IN, RK_SDYN_OUT, RK_SDYN_X0, RK_COMDYN_IN, RK_COMDYN_OUT, RK_COMDYN_X0,
    RK_HDYN_IN, RK_HDYN_OUT, RK_HDYN_X0,
    RK_UNITQ, RK_COMC, RK_CMMC, RK_AMB, RK_COMH, RK_FEETD, RK_JPB, RK_JVB, RK_FEETH,
    RK_FIN, RK_PER0, RK_PERN,
    // static pose finder only (pose_body.h)
    RK_PCOMPL, RK_PBAL, RK_PCOMERR, RK_PPREG, RK_PHAND, RK_COUNT
};
// hand position expressions of the pose finder (pose_body.h; planner.py:596-660): read from global memory by the two hand task
// groups — not part of KSettings, which every kernel stages into LDS
struct PoseHands {
    int32_t type[2];      // HIPNLP_EXPR_* of the left / right hand
    int32_t link[2];      // link the hand frame is attached to
    double R[2][9], o[2][3];   // link_T_frame
    double mult[2];       // regularisation cost multipliers
};
HD constexpr int row_id(int kind, int c, int i) { return (kind << 16) | (c << 8) | i; }
HD constexpr int rid_kind(int id) { return id >> 16; }
HD constexpr int rid_point(int id) { return (id >> 8) & 0xff; }
HD constexpr int rid_index(int id) { return id & 0xff; }
constexpr int COL_GLOBAL = NXK;  // column ids >= NXK address the horizon-global variables

// =====================================================================================================
// Native g slots (per knot)
// =====================================================================================================
namespace gs {
constexpr int FDYN = 0, PDYN = 3, PLANAR = 6, DCC = 9, HEIGHT = 10, NORMAL = 11, FRICTION = 12, UB = 13, FDB = 16, KINC = 19,
              FDYN_X0 = 22, PDYN_X0 = 25, PT_STRIDE = 28;
constexpr int G0 = NC * PT_STRIDE;  // 224
constexpr int PBDYN = G0, QBDYN = PBDYN + 3, SDYN = QBDYN + 4, COMDYN = SDYN + NJ, HDYN = COMDYN + 3,
              UNITQ = HDYN + 6, COMC = UNITQ + 1, CMMC = COMC + 3, AMB = CMMC + 3, COMH = AMB + 3, FEETD = COMH + 1,
              JPB = FEETD + 1, JVB = JPB + NJ, FEETH = JVB + NJ,
              PB_X0 = FEETH + 1, QB_X0 = PB_X0 + 3, S_X0 = QB_X0 + 4, COM_X0 = S_X0 + NJ, H_X0 = COM_X0 + 3,
              FIN = H_X0 + 6, PER = FIN + 105, COUNT = PER + 84;
}  // namespace gs

// =====================================================================================================
// Native jac slots (per knot): every structural entry of the knot's COLUMN block
// =====================================================================================================
namespace js {
// Three regions, so that a kernel whose destination already holds the entries that do not depend on x (VARY instantiations,
// hipnlp.hip) needs LDS for the varying ones only — 809 slots on the planar terrain, 1041 on the smooth steps, instead of 1962:
//   D  [0, V0)       entries that NEVER depend on x (emitted through emit_jc and nothing else): per point (PTC_STRIDE each), then global
//   B  [V0, PV0)     global entries that depend on x
//   P  [PV0, COUNT)  per-point entries that depend on x, stride pt_stride(terrain): the part both terrains use first, then EITHER the
//                    planar terrain's eight entries OR the smooth terrain's 39 (a slot behind offset PT_COMMON means different things on
//                    the two terrains; exactly one set of point tasks runs per handle)
// The static pose finder (pose_body.h) reuses the names for entries of its own; it allocates the whole range.
// ---- D, per point: base ptc(c) ----
constexpr int FDYN = 0;          // IN_X 3, IN_Y 3, OUT_X 3, OUT_Y 3, X0 3
constexpr int PDYN = 15;
constexpr int PLANAR_V = 30;                  // [3]  d row_i / d v_i = 1
constexpr int UB = 33, FDB = 36;
constexpr int KINC_P = 39, KINC_PB = 42;
constexpr int HEIGHT_Z = 45;                  // d h / d p_z = 1 (both terrains)
constexpr int PTC_STRIDE = 46;
constexpr int CG0 = NC * PTC_STRIDE;          // 368
// ---- D, global ----
// trivial dynamics blocks: L*5 slots each: IN_X, IN_Y, OUT_X, OUT_Y, X0
constexpr int PBDYN = CG0, QBDYN = PBDYN + 15, SDYN = QBDYN + 20, COMDYN = SDYN + 5 * NJ;
constexpr int HDYN_SELF_IN = COMDYN + 15, HDYN_SELF_OUT = HDYN_SELF_IN + 6, HDYN_X0 = HDYN_SELF_OUT + 6, HDYN_X0G = HDYN_X0 + 6;
constexpr int HDYN_LIN_F_IN = HDYN_X0G + 6, HDYN_LIN_F_OUT = HDYN_LIN_F_IN + 24;            // [c][i]
constexpr int COMC_COM = HDYN_LIN_F_OUT + 24, COMC_PB = COMC_COM + 3, CMMC_H = COMC_PB + 3, AMB = CMMC_H + 3;
constexpr int COMH_Z = AMB + 3;               // d h(com)/d com_z = 1 (both terrains)
constexpr int JPB = COMH_Z + 1, JVB = JPB + NJ, FEETH = JVB + NJ;
constexpr int FIN = FEETH + NC, PER0 = FIN + 81, PERN = PER0 + 84;
constexpr int V0 = PERN + 84;                 // 921: first slot that may depend on x
// ---- B: global, varying ----
constexpr int HDYN_ANG_P_IN = V0, HDYN_ANG_P_OUT = HDYN_ANG_P_IN + 48;     // [c][6]
constexpr int HDYN_ANG_F_IN = HDYN_ANG_P_OUT + 48, HDYN_ANG_F_OUT = HDYN_ANG_F_IN + 48;
constexpr int HDYN_ANG_COM_IN = HDYN_ANG_F_OUT + 48, HDYN_ANG_COM_OUT = HDYN_ANG_COM_IN + 6;
constexpr int UNITQ = HDYN_ANG_COM_OUT + 6;
constexpr int COMC_QB = UNITQ + 4, COMC_S = COMC_QB + 12;  // S [3][NJ]
constexpr int CMMC_QB = COMC_S + 3 * NJ, CMMC_QD = CMMC_QB + 12, CMMC_S = CMMC_QD + 12, CMMC_SD = CMMC_S + 3 * NJ;
constexpr int FEETD = CMMC_SD + 3 * NJ;       // [2][LEG_PATH]
constexpr int COMH_XY = FEETD + 2 * LEG_PATH; // [2] d h(com)/d com_x, com_y (smooth terrain)
constexpr int PV0 = COMH_XY + 2;              // 1386
// ---- P, per point: base pt(terrain, c); offsets inside a point's block ----
constexpr int FRICTION_F = 0;                 // [3]
constexpr int KINC_QB = 3, KINC_S = 15;       // QB [3][4], S [3][LEG_PATH]
constexpr int PT_COMMON = 15 + 3 * LEG_PATH;  // 33
// planar terrain
constexpr int PL_U = PT_COMMON;               // [2]  d row_i / d u_i, i = x, y (tau-dependent)
constexpr int PL_P = PL_U + 2;                // [2]  d row_i / d p_z
constexpr int PL_DCC_P = PL_P + 2, PL_DCC_F = PL_DCC_P + 1, PL_DCC_V = PL_DCC_F + 1, PL_DCC_FD = PL_DCC_V + 1;   // z components
// (two entries that are constant on the planar terrain only — d(n.f)/d f_z = 1, d row_z / d u_z = -1; the smooth terrain computes them —
//  live here, not in region D: js::COUNT stays what the four-wave kernels' 40 KB of LDS allow, at the price of 16 idle slots in a planar VARY kernel)
constexpr int PL_NORMAL_FZ = PL_DCC_FD + 1, PL_UZ = PL_NORMAL_FZ + 1;
constexpr int PT_STRIDE_PLANAR = PL_UZ + 1;   // 43
// smooth terrain
constexpr int PLANAR_U = PT_COMMON;           // [3][3]    d row_i / d u_j
constexpr int PLANAR_P = PLANAR_U + 9;        // [3][3]    d row_i / d p_j
constexpr int DCC_P = PLANAR_P + 9, DCC_F = DCC_P + 3, DCC_V = DCC_F + 3, DCC_FD = DCC_V + 3;   // [3] each
constexpr int HEIGHT_XY = DCC_FD + 3;         // [2]  d h / d p_x, p_y
constexpr int NORMAL_P = HEIGHT_XY + 2, NORMAL_F = NORMAL_P + 2;   // [2] d(n.f)/d p_x,p_y ; [3] d/d f
constexpr int FRICTION_P = NORMAL_F + 3;      // [2]
constexpr int PT_STRIDE_SMOOTH = FRICTION_P + 2;  // 72
constexpr int COUNT = PV0 + NC * PT_STRIDE_SMOOTH;   // 1962
HD constexpr int pt_stride(bool planar) { return planar ? PT_STRIDE_PLANAR : PT_STRIDE_SMOOTH; }
HD constexpr int ptc(int c) { return PTC_STRIDE * c; }
// slots a kernel of one terrain may touch behind V0 (what a VARY instantiation keeps in LDS)
HD constexpr int vary_slots(bool planar) { return PV0 - V0 + NC * pt_stride(planar); }
}  // namespace js

// variable behind periodicity row i (0..83): per point (u_v, f_dot), then h, v_b, qdot_b, sdot (planner.py:897-922)
constexpr int NPER = 84;
HD constexpr int periodicity_row_var(int i) {
    return i < 48 ? PT_ * (i / 6) + ((i % 6) < 3 ? U_ + (i % 6) : FD_ + ((i % 6) - 3))
                  : (i < 54 ? H_ + (i - 48) : (i < 57 ? VB_ + (i - 54) : (i < 61 ? QD_ + (i - 57) : SD_ + (i - 61))));
}

// the six structural entries of d(a x b)/db = [a]x, in (row, col) order
HD constexpr int cross_row(int e) { return e / 2; }
HD constexpr int cross_col(int e) { return (e % 2 == 0) ? (e / 2 == 0 ? 1 : 0) : (e / 2 == 2 ? 1 : 2); }
// e: 0 (0,1) 1 (0,2) 2 (1,0) 3 (1,2) 4 (2,0) 5 (2,1)

}  // namespace hipnlp
