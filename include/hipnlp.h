/*
 * hipnlp.h — C-ABI of the MI355X-native multiple-shooting NLP-callback engine.
 *
 * Drop-in boundary for ONE hot path of ami-iit/hippopt: the per-knot evaluation of
 * (f, grad f, g, jac g) of the humanoid kinodynamic multiple-shooting NLP that the
 * reference assembles in
 *   src/hippopt/turnkey_planners/humanoid_kinodynamic/planner.py:26-176
 * and evaluates, through CasADi's Opti/nlpsol, once per IPOPT callback in
 *   src/hippopt/base/opti_solver.py:479   (self._solver.solve()).
 *
 * What each entry point replaces on the reference side
 *   hipnlp_create       <- cs.Opti("nlp") + opti.variable()/parameter() creation order
 *                          (base/opti_solver.py:139-175, 251-333) and the planner's
 *                          subject_to/minimize call order (planner.py:124-176)
 *   hipnlp_set_params   <- opti.set_value(parameter, value)     (opti_solver.py:236-249)
 *   hipnlp_bounds       <- Opti's canonical lbg/ubg of every subject_to
 *   hipnlp_sparsity     <- nlp_jac_g sparsity (CCS, column-major)  [CasADi, 3rd party]
 *   hipnlp_eval         <- nlp_f / nlp_grad_f / nlp_g / nlp_jac_g   [CasADi SX VM, 3rd party]
 *                          = IPOPT's eval_f / eval_grad_f / eval_g / eval_jac_g
 *                          (IpStdCInterface.h callback quartet; no eval_h: the reference
 *                          runs hessian_approximation=limited-memory,
 *                          main_periodic_step.py:116)
 *   hipnlp_eval_device  <- same, device-resident in/out (no PCIe), used by bench.py and
 *                          by the multi-GPU path (RCCL all-gather of the output shards)
 *
 * Conventions
 *   - all reals are IEEE fp64; all indices are 0-based int32
 *   - return value 0 = ok, negative = error (see HIPNLP_E_*); no exceptions cross the ABI
 *   - the caller owns every host buffer; the library owns device memory
 *   - one handle is used by one thread at a time (IPOPT is single threaded)
 *   - x, lambda order  = reference creation order: [knot 0 (189) | ... | knot N-1 | initial_state.centroidal_momentum (6)]
 *   - p order          = reference parameter creation order (tests/golden/kinodyn_structure.json)
 *   - g order          = reference subject_to call order (constraint-type-major, knot-minor)
 *   - jac order        = CCS (sorted by column, then row), fixed at create time
 */
#ifndef HIPNLP_H
#define HIPNLP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version of this header.  The caller stores it in hipnlp_desc.abi_version / hipnlp_pose_desc.abi_version; hipnlp_create and
 * hipnlp_pose_create refuse a descriptor built against another version (a C caller compiled against an older header would pass a
 * shorter struct).  History: 1 = rounds 1-2 (no version field); 2 = abi_version + flags in the descriptors, m_full / n_lifted in
 * hipnlp_dims, detect_simple_bounds layout, auto-registration, IPOPT callback quartet (hipnlp_ipopt.h); 3 = round 6:
 * hipnlp_terrain_step.top_normal (sloped step tops: the descriptors grew by 24 bytes per terrain step).                            */
#define HIPNLP_ABI_VERSION 3

#define HIPNLP_NJ 23        /* actuated joints (ergoCub: torso 3, arms 4+4, legs 6+6) */
#define HIPNLP_NL 24        /* links = root + one per joint                            */
#define HIPNLP_NC 8         /* contact points: left[0..3], right[0..3]                 */
#define HIPNLP_NXK 189      /* decision variables per knot                             */
#define HIPNLP_NPK 79       /* parameters per knot (24 descriptors + 55 references)    */
#define HIPNLP_NXG 6        /* horizon-global decision variables                       */
#define HIPNLP_NPG 326      /* horizon-global parameters                               */

/* error codes */
#define HIPNLP_OK 0
#define HIPNLP_E_INVALID (-1)   /* bad argument / inconsistent descriptor */
#define HIPNLP_E_NODEVICE (-2)  /* no HIP device / HIP runtime error      */
#define HIPNLP_E_ALLOC (-3)
#define HIPNLP_E_PARAMS (-4)    /* parameters not set before eval (opti_solver.py:447-450) */
#define HIPNLP_E_NUMERIC (-5)   /* NaN/Inf produced by the kernel (IPOPT callback must return false) */
#define HIPNLP_E_UNSUPPORTED (-6) /* the request is valid but this part is not built (hipnlp_last_error says which) */

/* ExpressionType of base/problem.py:15-19 */
#define HIPNLP_EXPR_SKIP 0
#define HIPNLP_EXPR_SUBJECT_TO 1
#define HIPNLP_EXPR_MINIMIZE 2

/* Terrain kinds.
 *   PLANAR        robot_planning/utilities/planar_terrain.py:7-41       h = p_z, n = e_z, R_t = I
 *   SMOOTH_STEPS  TerrainSum of SmoothTerrain.step bumps (utilities/smooth_terrain.py:201-227,266-336, terrain_sum.py:19-38;
 *                 the stairs of main_walking_on_stairs.py:18-28):  h = p_z - sum_s [ H_s exp(-g_s^(2 side_s)) + o_s,z ],
 *                 g_s = (2 q_x / L_s)^(2 edge_s) + (2 q_y / W_s)^(2 edge_s),  q = Rz(orientation_s)^T (p - o_s);
 *                 a step with a sloped top (hipnlp_terrain_step.top_normal):  H_s  ->  pi_s(q) = H_s - (n_x q_x + n_y q_y) / n_z;
 *                 normal / orientation from the TerrainDescriptor defaults (utilities/terrain_descriptor.py:45-80)      */
#define HIPNLP_TERRAIN_PLANAR 0
#define HIPNLP_TERRAIN_SMOOTH_STEPS 1
#define HIPNLP_MAX_TERRAIN_STEPS 4

typedef struct hipnlp_terrain_step {
    double length, width, height;
    double position[3];
    double orientation;     /* yaw of the bump (rad) */
    int32_t edge_sharpness; /* default 5  (exponent 2*edge on the footprint) */
    int32_t side_sharpness; /* default 10 (exponent 2*side on g)            */
    /* SmoothTerrain.step(top_normal_direction=...) (utilities/smooth_terrain.py:238-264; the ramp of main_walking_on_ramp.py:18-30,
     * 403-409): the top surface of the step is the plane through (0, 0, height) of the step's frame with this normal,
     *     pi(q_x, q_y) = height - (n_x q_x + n_y q_y) / n_z,   n = top_normal / |top_normal|,
     * and the bump is exp(-g^(2 side)) pi(q).  The zero vector (a zeroed struct) = the reference's None: the flat top, pi = height.
     * Refused as the reference refuses them: |top_normal| < 1e-6 (but not zero), |n_z| < 1e-6. */
    double top_normal[3];
} hipnlp_terrain_step;

/* Frame slots of hipnlp_robot_model.frame_* */
#define HIPNLP_FRAME_LEFT_SOLE 0
#define HIPNLP_FRAME_RIGHT_SOLE 1
#define HIPNLP_FRAME_CHEST 2

/*
 * Kinematic tree, adam-robotics conventions:
 *   child link of joint j is link j+1; parent[j] is ANY other link of a tree rooted at link 0 = root_link (the joint order is the
 *   reference's joints_name_list order, which fixes the x layout and need not follow the tree; chains of at most 8 joints).
 *   parent_T_child(s_j) = [ R_fix[j] * Rot(axis[j], s_j) , o_fix[j] ]   (URDF origin xyz/rpy, then axis rotation)
 *   inertials are expressed in the link frame: mass, com, 3x3 inertia about the com (row-major).
 *   frames: rigidly attached to frame_link[f] with link_T_frame = [frame_R[f], frame_o[f]].
 */
typedef struct hipnlp_robot_model {
    int32_t parent[HIPNLP_NJ];
    double R_fix[HIPNLP_NJ][9];
    double o_fix[HIPNLP_NJ][3];
    double axis[HIPNLP_NJ][3];
    double mass[HIPNLP_NL];
    double com[HIPNLP_NL][3];
    double inertia[HIPNLP_NL][9];
    int32_t frame_link[3];
    double frame_R[3][9];
    double frame_o[3][3];
} hipnlp_robot_model;

/*
 * Everything that is a python-side *setting* (a constant baked into the CasADi graph) in the
 * reference, as opposed to an Opti parameter.  Field -> reference:
 *   horizon                       settings.horizon_length                 (settings.py:19)
 *   final_state_type/weight       settings.final_state_expression_*       (planner.py:417-425)
 *   periodicity_type/weight       settings.periodicity_expression_*       (planner.py:923-930)
 *   *_multiplier / *_weights      settings.*                              (planner.py:249-264,433-520,746-895)
 *   yaw_corner[foot][0..2]        bottom-right / top-right / top-left point index per foot,
 *                                 chosen from the descriptor geometry     (planner.py:773-828)
 *   joint_reg_as_coded            1: evaluate joint_positions_error exactly as coded
 *                                 (elementwise * of a 23x23 diagonal with a 23x1 vector,
 *                                 planner.py:505-520);  0: intended sum_i (sdot_i + w_i e_i)^2
 */
typedef struct hipnlp_settings {
    int32_t horizon;
    int32_t terrain;
    int32_t final_state_type;
    int32_t periodicity_type;
    int32_t joint_reg_as_coded;
    int32_t yaw_corner[2][3];
    double final_state_weight;
    double periodicity_weight;
    double contacts_centroid_cost_multiplier;
    double com_linear_velocity_cost_weights[3];
    double com_linear_velocity_cost_multiplier;
    double desired_frame_quaternion_cost_multiplier;
    double base_quaternion_cost_multiplier;
    double base_quaternion_velocity_cost_multiplier;
    double joint_regularization_cost_weights[HIPNLP_NJ];
    double joint_regularization_cost_multiplier;
    double force_regularization_cost_multiplier;
    double foot_yaw_regularization_cost_multiplier;
    double swing_foot_height_cost_multiplier;
    double contact_velocity_control_cost_multiplier;
    double contact_force_control_cost_multiplier;
    int32_t n_terrain_steps;   /* SMOOTH_STEPS: number of bumps (1..HIPNLP_MAX_TERRAIN_STEPS) */
    int32_t reserved_;
    hipnlp_terrain_step terrain_steps[HIPNLP_MAX_TERRAIN_STEPS];
} hipnlp_settings;

typedef struct hipnlp_desc {
    hipnlp_settings settings;
    hipnlp_robot_model model;
    int32_t batch;       /* independent trajectories evaluated per call (>=1); x/p/outputs get a leading batch dim */
    int32_t knot_begin;  /* shard: this handle evaluates knots [knot_begin, knot_end) of every trajectory   */
    int32_t knot_end;    /*        (0, horizon) = whole horizon.  See DESIGN.md §6 (multi-GPU)              */
    int32_t device;      /* HIP device ordinal                                                               */
    int32_t abi_version; /* HIPNLP_ABI_VERSION of the header the caller was compiled against (checked by hipnlp_create) */
    int32_t flags;       /* HIPNLP_FLAG_*                                                                     */
} hipnlp_desc;

/* hipnlp_desc.flags
 *   DETECT_SIMPLE_BOUNDS  the handle IS the NLP nlpsol hands to IPOPT when Opti runs with {"detect_simple_bounds": True}
 *                         (main_periodic_step.py:109-110): every row of g that is exactly one decision variable — the u_v and
 *                         joint position / velocity boxes (planner.py:386-405,699-719; 70 rows per interior knot), the
 *                         `x_0 == initial_state` rows, the final-state rows that hold a variable — leaves g and jac g and becomes a
 *                         bound on that variable: m, nnz, the sparsity pattern, g, jac g, lbg / ubg and the multipliers of
 *                         hipnlp_eval_hess are those of the REDUCED problem, lbx / ubx carry the lifted bounds; the kernels never
 *                         compute or move the lifted rows.  hipnlp_lift_map relates the reduced rows to the reference's full list
 *                         of named constraints (hipnlp_row_block stays in full numbering).                                        */
#define HIPNLP_FLAG_DETECT_SIMPLE_BOUNDS 1
/*   JAC_VARYING_FIRST     order of the non-zeros of jac g INSIDE the column block of a knot (hipnlp_sparsity, every jac value array):
 *                         the entries that depend on x first (in CCS order), the constant ones behind them (in CCS order), instead of
 *                         one CCS run.  43 % of the pattern at 100 knots is constant — the +-1, -dt/2, mass entries of the trapezoid
 *                         defects (integrators/implicit_trapezoid.py:24-39), the x_0 rows (base/multiple_shooting_solver.py:713-742)
 *                         and the single-variable bound rows (planner.py:386-405,699-719) — and the host path stores only the varying
 *                         entries (hipnlp_set_constant_jacobian): in this order they are ONE contiguous run per knot on the PCIe
 *                         link instead of fragments of a few doubles.  IPOPT takes triplets in any order (eval_jac_g's structure
 *                         call), so a C binding sets the flag; a consumer that wants CasADi's CCS value array leaves it unset.
 *                         Knot blocks keep their places and sizes (shards, hipnlp_dims are unchanged).                              */
#define HIPNLP_FLAG_JAC_VARYING_FIRST 2

typedef struct hipnlp_dims {
    int32_t n;        /* decision variables per trajectory = 189*N + 6                */
    int32_t m;        /* constraint rows per trajectory                               */
    int32_t nnz;      /* structural non-zeros of jac g per trajectory                 */
    int32_t np;       /* parameters per trajectory = 79*N + 326                       */
    int32_t nnz_knot; /* non-zeros in the column block of one interior knot           */
    int32_t m_knot;   /* rows owned by one interior knot                              */
    /* shard view (== full view when the handle owns the whole horizon) */
    int32_t shard_g_rows;   /* rows written by this handle          */
    int32_t shard_nnz;      /* jac values written by this handle    */
    int32_t shard_grad;     /* grad entries written by this handle  */
    int32_t shard_jac_off;  /* offset of the handle's jac block in the CCS value array */
    int32_t shard_grad_off; /* offset of the handle's block in grad f                  */
    int32_t m_full;         /* rows of the reference's full subject_to list (== m without DETECT_SIMPLE_BOUNDS)   */
    int32_t n_lifted;       /* rows of that list that are exactly one decision variable (m_full - n_lifted = m when lifted) */
} hipnlp_dims;

typedef struct hipnlp_handle hipnlp_handle;

int hipnlp_abi_version(void);            /* HIPNLP_ABI_VERSION the library was built with */
const char* hipnlp_build_info(void);     /* how the library was built (e.g. "gfx950; ds_read2_b64 split in the assembly" or "gfx950; plain hipcc") */
/* NUMA node of the host that card `device` hangs off (-1: not known).  The host-buffer paths want the calling thread and the arrays it
 * passes on THAT node (its CPUs: /sys/devices/system/node/node<N>/cpulist): on a two-socket host the same call measured 37.6 - 39.0 us
 * with the caller pinned to the card's node and 40.2 - 41.7 us on the other one (hipnlp_eval, all four outputs, 100 knots), the exact
 * Hessian 55 against 60 - 62 us — and its early run (hipnlp_set_hessian_early_run) a gain on the card's node, a loss on the other. */
int hipnlp_device_numa_node(int device, int* node);
/* ... and the calling THREAD onto the CPUs of that node (sched_setaffinity with those of them it is allowed to run on; arrays it
 * allocates from then on follow by first touch).  node_out / cpus_out (may be NULL): the node and the number of CPUs the thread now has —
 * -1 / 0 when nothing was changed (node unknown, none of its CPUs allowed). */
int hipnlp_pin_thread_to_device_numa_node(int device, int* node_out, int* cpus_out);
int hipnlp_create(const hipnlp_desc* desc, hipnlp_handle** out);
void hipnlp_destroy(hipnlp_handle* h);
const char* hipnlp_last_error(const hipnlp_handle* h); /* h may be NULL: last create() error */
int hipnlp_get_dims(const hipnlp_handle* h, hipnlp_dims* out);

/* p: [batch][np] host array in reference parameter order. Computes bounds, uploads device copies. */
int hipnlp_set_params(hipnlp_handle* h, const double* p);

/* Canonical bounds (CasADi Opti canon form), valid after set_params.  Any pointer may be NULL.
 * lbx/ubx: [n] (+-inf: the reference adds no explicit variable bounds; with DETECT_SIMPLE_BOUNDS: the bounds of the lifted rows)
 * lbg/ubg: [m] of trajectory 0 (bounds depend on parameters only)                              */
int hipnlp_bounds(const hipnlp_handle* h, double* lbx, double* ubx, double* lbg, double* ubg);

/* is_simple[m_full]: 1 where row i of the FULL subject_to list is exactly one decision variable (what nlpsol's
 * detect_simple_bounds lifts, main_periodic_step.py:110); var_index[m_full]: that variable or -1. */
int hipnlp_simple_rows(const hipnlp_handle* h, int32_t* is_simple, int32_t* var_index);
/* The handle's rows against the reference's full list of named constraints (hipnlp_row_block): kept_row[m_full] = row of this
 * handle's g behind full row r, or -1 when the row was lifted into a bound (identity without DETECT_SIMPLE_BOUNDS);
 * lb_full / ub_full [m_full] = canonical bounds of every full row (valid after set_params).  What a binding needs to hand
 * IPOPT's multipliers back per named constraint (Output.constraint_multipliers, opti_solver.py:530-537): lambda of a kept row is
 * lambda[kept_row]; the multiplier of a lifted row is the bound multiplier z_U - z_L of its variable (hipnlp_simple_rows), given
 * to the row whose bound is the active one.  Any pointer may be NULL. */
int hipnlp_lift_map(const hipnlp_handle* h, int32_t* kept_row, double* lb_full, double* ub_full);

/* CCS pattern as triplets sorted by (col,row): irow[nnz], jcol[nnz] (IPOPT eval_jac_g, values==NULL call).
 * With HIPNLP_FLAG_JAC_VARYING_FIRST: knot blocks in knot order, inside a block the varying entries (col,row)-sorted, then the constant ones. */
int hipnlp_sparsity(const hipnlp_handle* h, int32_t* irow, int32_t* jcol);
/* mask[nnz] (order of hipnlp_sparsity): 1 where the entry of jac g does not depend on x (its value is a function of the parameters:
 * +-1, -dt/2, the mass, +-1/4), 0 elsewhere. */
int hipnlp_jac_constant_mask(const hipnlp_handle* h, unsigned char* mask);

/* Host-buffer callback quartet (IPOPT eval_f/eval_grad_f/eval_g/eval_jac_g in one fused launch).
 * x: [batch][n]; f: [batch]; grad_f: [batch][n]; g: [batch][m]; jac: [batch][nnz].  Any output may be NULL.
 * new_x = 0 lets the library return cached results of the previous evaluation (IPOPT's flag); new_x < 0 = "unknown": the library
 * compares x with its staging copy of the previous evaluation (what a binding without IPOPT's flag — cyipopt, SciPy — passes).
 *
 * What crosses PCIe (the non-NULL outputs are the call's WANT MASK): a new evaluation copies x into a pinned staging block the
 * kernel reads directly, evaluates all four outputs in ONE launch and lets the kernel store the wanted ones — plus the handle's
 * prefetch set — straight into a pinned output block (no copy command, no second staging copy); the others stay in HBM and are
 * fetched by one asynchronous copy if a later call with new_x = 0 asks for them.  An eval_g at a line-search trial point therefore
 * moves 0.2 MB, not the 1.1 MB of jac g.  Returns HIPNLP_E_NUMERIC when the evaluation produced a NaN/Inf — AFTER filling the
 * outputs (IPOPT's callback returns false; the reference hands CasADi's NaNs to IPOPT, which cuts the step).                    */
int hipnlp_eval(hipnlp_handle* h, const double* x, int new_x,
                double* f, double* grad_f, double* g, double* jac);

#define HIPNLP_WANT_F 1u
#define HIPNLP_WANT_GRAD 2u
#define HIPNLP_WANT_G 4u
#define HIPNLP_WANT_JAC 8u
#define HIPNLP_WANT_ALL 15u
/* Same evaluation, zero-copy: returns pointers INTO the library's pinned output block (valid until the next call on the handle)
 * for the outputs named in `want` (HIPNLP_WANT_*), NULL for the others.  What a binding that must copy anyway (cyipopt copies the
 * callback's return value into IPOPT's array) uses to avoid a second copy.                                                      */
int hipnlp_eval_pinned(hipnlp_handle* h, const double* x, int new_x, unsigned want,
                       const double** f, const double** grad_f, const double** g, const double** jac);
/* Outputs every NEW evaluation brings to the host besides the ones its call asks for.  Default F | GRAD | G (0.37 MB at 100 knots:
 * +7 us): IPOPT asks for f and g at every trial point and for grad f and jac g (new_x = 0) only at accepted ones.
 * HIPNLP_WANT_ALL makes every new evaluation move everything (one launch, no second round trip for jac g).                      */
int hipnlp_set_prefetch(hipnlp_handle* h, unsigned mask);
/* Constant entries of jac g on the host path (ON by default; takes effect on handles created with HIPNLP_FLAG_JAC_VARYING_FIRST only: in
 * CCS order the varying entries of a block are fragments of one to four doubles, and storing fragments over PCIe measured slower than
 * storing everything).  A HOST destination of the Jacobian values — the library's pinned block, or
 * a caller array inside a registered range (hipnlp_host_register / auto-registration) — is filled with the constant entries ONCE per
 * parameter set (the pinned block when a launch first needs it, a caller array at its first use and again after hipnlp_set_params);
 * the kernel then stores the entries that depend on x only: 628 KB instead of 1.1 MB per 100-knot Jacobian on a path that is bound by
 * the bytes crossing PCIe.  What the caller gets is the complete value array, every call.  Contract for a REGISTERED caller array: the
 * caller does not write into it between calls (IPOPT's TNLPAdapter never touches its jac_g_ buffer).  A caller that does anyway is
 * covered by a spot check — sixteen constant entries spread over the array, its first and its last among them, are compared before every
 * such launch and the array is re-filled when one differs (hipnlp_host_stats out[5] counts these) — not by a guarantee: a caller
 * that edits single entries of a registered Jacobian array must turn this off.  Arrays that are NOT registered always receive complete
 * values by a copy out of the pinned block.
 * Device-resident outputs (hipnlp_eval_device): a jac_dev buffer is filled with the constant entries at its first
 * sight and again after hipnlp_set_params (one small launch on the call's stream, in front of the evaluation); the evaluations store the
 * varying entries of every knot block only — one run in the varying-first order, scattered over the block in CasADi's CCS order (on = 1
 * on a handle WITHOUT HIPNLP_FLAG_JAC_VARYING_FIRST asks for exactly this: device destinations only, host destinations of a CCS handle keep
 * receiving every entry) — and no longer stage the constants in LDS at all.  Same contract — the caller does not write into
 * the buffer between calls — and the launch itself samples it: the one workgroup per trajectory that sums the cost compares constant
 * entries of every knot block with the handle's templates — four positions per block (its first, its last, two in between), two of them
 * per launch, alternating with the launches INTO THAT BUFFER — and puts the constants of all the trajectory's blocks back when one differs (hipnlp_host_stats out[7] counts
 * such repairs).  A sample, not a guarantee: constants
 * overwritten elsewhere in a block are returned as they are.  The record "this buffer holds the constants" is keyed by the buffer's
 * ADDRESS: a caller that frees a jac buffer and later hands over other memory at the same address (a caching allocator does that) must
 * call hipnlp_forget_jac_destination first — the next evaluation then fills the buffer again.
 * hipnlp_eval_device_shard / _peers are not affected: every entry is stored.
 * on = 0: every launch stores every entry (the behaviour of ABI 2 libraries before this switch existed).
 * Default: on for handles created with HIPNLP_FLAG_JAC_VARYING_FIRST, off for handles in CCS order. */
int hipnlp_set_constant_jacobian(hipnlp_handle* h, int on);
/* Forget that the jac destination at address p (device buffer of hipnlp_eval_device, or registered host array of hipnlp_eval) holds this
 * handle's constant entries; p = NULL: every destination the handle remembers.  The next evaluation into it fills the constants again.
 * For callers whose allocator may hand out the same address for a new buffer.  Returns the number of records dropped (>= 0), or an
 * error code (< 0).  Host arrays need no such call since ABI-compatible revision 5: their record is tied to the array's REGISTRATION
 * (hipnlp_host_register / auto-registration), and an array that was unregistered — by anybody — is filled again when it comes back. */
int hipnlp_forget_jac_destination(hipnlp_handle* h, const void* p);
/* Early outputs (opt-in, off by default).  on = 1: a NEW evaluation stores g and jac g — when its call does NOT ask for them —
 * straight into the REGISTERED caller arrays (hipnlp_host_register, or registered by the handle itself) that earlier calls passed
 * for them, before the caller asks.  IPOPT's sequence eval_f(new x), eval_g, eval_grad_f, eval_jac_g then costs one launch that moves
 * everything and calls that return at once (grad f comes from the pinned block: one 151 KB host copy), instead of a second 1.1 MB
 * transfer when the Jacobian is asked for; a rejected line-search trial point pays for a Jacobian it never reads.
 * Contract: the contents of such a g / jac array are overwritten by every new_x call on the handle from the moment the array has been
 * passed once.  That is safe for arrays the caller treats as scratch and re-reads right after the callback that fills them — IPOPT's
 * TNLPAdapter evaluates g and the Jacobian values into buffers of its own (full_g_, jac_g_) and copies out of them.  It is NOT safe
 * for grad f under IPOPT: the adapter hands eval_grad_f the storage of IPOPT's own gradient vector, which stays alive (as the
 * gradient at the current iterate) while trial points are evaluated — so grad f is excluded unless the caller asks for it with
 * on = 2, which is for callers whose grad array is scratch of their own too (hippopt_amd's HipNlpSolver callback cache).
 * A cached call (new_x = 0) that passes a DIFFERENT array for such an output re-evaluates. */
int hipnlp_set_early_outputs(hipnlp_handle* h, int on);
/* Bracket host-path launches with HIP events (for hipnlp_last_kernel_ms); off by default: an event pair costs microseconds. */
int hipnlp_set_host_timing(hipnlp_handle* h, int on);
/* Wall clock (us) of the last hipnlp_eval on the host side: us[0] staging copy of x, us[1] enqueue (launch call), us[2] waiting for
 * the GPU (kernel + its PCIe stores), us[3] copies into caller arrays that are not registered.  Diagnostic. */
int hipnlp_host_breakdown(const hipnlp_handle* h, double* us /*[4]*/);
/* Caller-owned host arrays as DIRECT kernel outputs: register the array once (page-locks it; dev_ptr, optional, receives its
 * device-visible address).  From then on a grad_f / g / jac pointer of hipnlp_eval that lies inside a registered range is stored
 * to by the kernel itself — no staging copy at all (IPOPT's TNLP adapter keeps its g and jac-value arrays for the whole solve: the
 * binding registers them at the first callback).  The device-visible address can also be passed to hipnlp_eval_device as
 * grad_dev / g_dev / jac_dev.  Also how several processes share ONE host buffer (a shared-memory segment mapped and registered by
 * each: hippopt_amd/sharded.py HostSink).  Process-wide (no handle); the caller unregisters before freeing the memory.           */
int hipnlp_host_register(void* p, size_t bytes, void** dev_ptr);
int hipnlp_host_unregister(void* p);
/* Auto-registration (ON by default): a grad_f / g / jac array (hipnlp_eval) or Hessian-value array (hipnlp_eval_hess) of at least 64 KB
 * that the call sees at the same address on two
 * CONSECUTIVE calls is registered by the handle itself (as hipnlp_host_register would) and is a direct kernel output from then on —
 * a plain binding that simply passes IPOPT's arrays gets the fast path without knowing about it.  Safety: a registration pins the
 * PAGES behind an address; should the caller free the array and the allocator map other pages at the same address, the kernel's
 * stores would go to the old pages.  Every use of an auto-registered array is therefore verified: the call writes a word of its own
 * into the first and the last entry before the evaluation and finds both overwritten afterwards (every entry of every output is
 * written by every evaluation) — if not, the range is dropped, the address is left alone from then on and the call is served through
 * the pinned block as for any unregistered array.  The check covers every pointer into a range that ANY handle of the process registered
 * by itself (the table of ranges is process wide).  At most six ranges per handle (the oldest one that the call in progress does not use
 * goes first); all are released by hipnlp_destroy or hipnlp_set_auto_register(h, 0).  Arrays registered explicitly with
 * hipnlp_host_register are the caller's responsibility and are not verified; registering an array a handle had registered by itself
 * replaces that registration with a fresh one. */
int hipnlp_set_auto_register(hipnlp_handle* h, int on);
/* Releases every range that handles of this process registered by themselves (returns how many); the handles register their arrays
 * again when they next see them twice in a row.  For a caller about to free arrays it passed to hipnlp_eval: memory that is freed while
 * a registration of it is alive and then handed out again by the allocator confuses the HIP runtime — a hipMemcpy from such memory is
 * refused with "invalid argument" (the library's own copies release the ranges and retry when that happens). */
int hipnlp_host_release_auto_ranges(void);
/* Counters of the host-buffer path: out[0] arrays auto-registered so far, out[1] stale-mapping fallbacks, out[2] ranges currently
 * auto-registered, out[3] evaluations so far (kernel launches), out[4] caller arrays filled with the constant Jacobian entries so far,
 * out[5] of which re-fills after a failed spot check, out[6] constant entries of the handle's pattern, out[7] wave slices of constants
 * the kernels put back into device buffers the caller had written over (read from the device: the call waits for a small copy). */
int hipnlp_host_stats(const hipnlp_handle* h, long* out /*[8]*/);


/* Exact Hessian of the Lagrangian  sigma f(x) + lambda^T g(x)  of the kinodynamic NLP — IPOPT's eval_h (IpStdCInterface.h:
 * Eval_H_CB(n, x, new_x, obj_factor, m, lambda, new_lambda, nele_hess, iRow, jCol, values, ud)); what CasADi's nlp_hess_l
 * computes for the reference when a script does not set `hessian_approximation = limited-memory` (SURVEY 8f rank 1;
 * opti_solver.py:123-125,479).  Lower triangle (irow >= jcol) as triplets with a fixed pattern: the Hessian is block diagonal
 * by knot (the trapezoid defects are sums of one-knot terms), plus the 84 entries coupling the last knot with the first when
 * the periodicity expression is a cost.  Order: knot blocks in knot order, each sorted by (column, row), then the coupling.
 * Both terrains (1539 triplets per knot on the planar terrain, 1878 on the smooth steps, whose contact rows need the bump sum to
 * fourth order: truncated Taylor polynomials, knot_hess_terrain.h).
 *   hipnlp_hess_nnz / _sparsity   structure (the `values == NULL` call of eval_h); a shard handle reports its own knots' blocks
 *   hipnlp_eval_hess              host buffers: x [batch][n], obj_factor [batch], lambda [batch][m] -> values [batch][nnz_h]
 *                                 (a `values` array inside a range registered with hipnlp_host_register is written by the kernel
 *                                 directly: 70 instead of 113 us per 100-knot call)
 *   hipnlp_eval_hess_device       device pointers, enqueued on `stream`, not synchronised                                   */
int hipnlp_hess_nnz(hipnlp_handle* h, int64_t* nnz_h);
int hipnlp_hess_sparsity(hipnlp_handle* h, int32_t* irow, int32_t* jcol);
int hipnlp_eval_hess(hipnlp_handle* h, const double* x, const double* obj_factor, const double* lambda, double* values);
/* The same with IPOPT's new_x flag (Eval_H_CB: FALSE when x is the x of the evaluation callbacks before it — the case at every accepted
 * iterate): new_x = 0 = the x of the previous host-buffer call on this handle (hipnlp_eval*, hipnlp_eval_hess*) — its staged copy is
 * used, no host copy of x in front of the launch; new_x < 0 = unknown (compared); anything else, or nothing staged yet: x is copied.
 * new_x = 0 is a statement about THIS handle: the callbacks at x ran on it.  (Sixty-four words of x are compared with the staged copy
 * all the same and a differing sample stages x: a binding that runs the callbacks on another handle gets the right Hessian, not a stale one.)
 * hipnlp_eval_hess is new_x = 1; hipnlp_ipopt_eval_h passes IPOPT's flag. */
int hipnlp_eval_hess_at(hipnlp_handle* h, const double* x, int new_x, const double* obj_factor, const double* lambda, double* values);
/* hipnlp_eval_hess* into host memory: the run at the start of every knot block (the point columns: a quarter of the values on the planar
 * terrain, two fifths on the smooth steps) is final long before the program ends and can leave then — the same kernel, the same values,
 * bit for bit.  Whether that is FASTER is a property of the host: measured 4 - 7 us sooner per 100-knot Hessian with the calling thread on
 * the card's NUMA node (hipnlp_pin_thread_to_device_numa_node) and 2 - 3 us later from the other socket of the same box.  mode 1 / 0: on / off;
 * -1 (the default): decided once, by the handle's first Hessian call, without a clock wherever the host says enough — off when the
 * handle's launches cannot send a run ahead at all (long launches on the compact layout), on when the calling thread runs on the card's
 * NUMA node (sysfs), off when it runs on another one — the same choice in every run on the same host; only where the topology is not
 * known (one-node hosts, containers) does the handle try both on its own first calls (three to warm up, nine of each kind alternating,
 * launch to completion on the host's clock) and keep the kind with the lower MEDIAN.  Setting a mode starts that over.
 * hipnlp_get_hessian_early_run: the mode, what is in use (-1: not decided yet) and — measured choices only — the two medians in
 * microseconds (0: not measured); hipnlp_hessian_early_run_reason: one sentence on how the choice was made. */
int hipnlp_set_hessian_early_run(hipnlp_handle* h, int mode);
int hipnlp_get_hessian_early_run(const hipnlp_handle* h, int* mode, int* chosen, double* us_off, double* us_on);
const char* hipnlp_hessian_early_run_reason(const hipnlp_handle* h);
int hipnlp_eval_hess_device(hipnlp_handle* h, const double* x_dev, const double* obj_factor_dev, const double* lambda_dev,
                            double* values_dev, void* stream);
/* Device-resident variant: all pointers are device pointers on desc.device (or device-visible addresses of registered host
 * memory, hipnlp_host_register), same shapes.  Work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the handle's
 * own non-blocking stream — NOT the legacy default stream: order other work against it with an explicit stream) and NOT
 * synchronised.  Launches of ONE handle must not overlap on the device (one stream at a time, or ordered streams), and a launch is
 * not replayable from a captured hipGraph: every launch carries its own sequence number, which tags the cost partials the launch's
 * reducer workgroup waits for and the non-finite flag. */
int hipnlp_eval_device(hipnlp_handle* h, const double* x_dev,
                       double* f_dev, double* grad_dev, double* g_dev, double* jac_dev,
                       void* stream);

/* Sharded variant (multi-GPU): the handle owns knots [knot_begin, knot_end).  Outputs are
 * shard-local and contiguous so that one RCCL all-gather reassembles them:
 *   grad_shard [batch][shard_grad]  = grad f entries of the shard's knots (+ the 6 global entries on the last shard)
 *   jac_shard  [batch][shard_nnz]   = the CCS value run of the shard's column blocks
 *   g_stage    [batch][nk][HIPNLP_G_STAGE] knot-major staging of the rows each knot owns; hipnlp_stage_rows()
 *              gives the global row of every staging slot (-1 = unused) for the scatter after the gather
 *   f_dev      [batch] partial cost of the shard's knots                                                     */
#define HIPNLP_G_STAGE 550
int hipnlp_eval_device_shard(hipnlp_handle* h, const double* x_dev, double* f_dev, double* grad_shard,
                             double* g_stage, double* jac_shard, void* stream);
int hipnlp_stage_rows(const hipnlp_handle* h, int k, int32_t* rows /*[HIPNLP_G_STAGE]*/);
/* Reassembly behind the all-gather of the fused shard buffers: out[i] = gathered[src[i]] (i < count: [grad | jac | g] in reference
 * order) and *f_out = sum over ranks, in rank order, of gathered[r * shard_len] (the cost partials); device pointers on the current
 * device, enqueued on `stream`, not synchronised.  Stateless (no handle): `stream` must be the stream the shard evaluation and the
 * all-gather were enqueued on (a NULL stream here is the legacy default stream, which nothing orders against a handle's own
 * stream: hippopt_amd/sharded.py passes one explicit stream to both calls). */
int hipnlp_reassemble(const double* gathered_dev, const int64_t* src_dev, double* out_dev, int64_t count, int world, int64_t shard_len,
                      double* f_out_dev, void* stream);

/* ---- the same exchange WITHOUT a collective: peer stores over xGMI (one process per GPU) ----------------------------
 * xGMI is point to point: instead of a ring all-gather followed by a reassembly pass, every rank pushes its fused shard buffer —
 * entry by entry, already at its position in the reference's order — into the output buffer of EVERY rank (its own included) with
 * one kernel of plain stores, then raises its flag in every rank's flag array; a rank's outputs are complete when all `world` flags
 * carry the step number.  The buffers live in fine-grained device memory shared through HIP IPC handles (64 bytes each, moved
 * between the processes by any channel: hippopt_amd/sharded.py uses torch.distributed's object gather once, at set-up); nothing here
 * depends on torch or RCCL.  All calls are enqueued on `stream` of the CURRENT device and not synchronised.
 *   hipnlp_ipc_alloc   fine-grained (uncached) device memory on `device`, zeroed, + its IPC handle
 *   hipnlp_ipc_open    maps a peer's allocation into this process (enables peer access to its device on demand)
 *   hipnlp_peer_push   peer_out[r][dst[i]] = shard[i] for every rank r and every i < count with dst[i] >= 0 (peer_out: a DEVICE array
 *                      of `world` device pointers)
 *   hipnlp_peer_signal system-scope fence, then peer_flags[r][rank] = seq for every r < world (peer_flags: device array of `world` pointers —
 *                      every rank's flag array, or a subset: gather_to_root signals rank 0 alone; rank < 64: this rank's slot)
 *   hipnlp_peer_signal_checked  the same, but with bit 63 set in the value when *status_dev != 0 at that point of the stream: the step of
 *                      THIS rank has failed (a wait of its own gave up), what it pushed must not pass for an evaluation on the
 *                      receiving side either — hipnlp_peer_wait takes a flag with that bit for "arrived, poisoned" (seq < 2^63)
 *   hipnlp_peer_wait   spins (bounded: ~2^20 polls — about a second) until flags[r] >= seq for all r, then
 *                      out[f_off + world] = sum over r, in rank order, of out[f_off + r] (the cost partials).  A wait that gives up
 *                      is sticky and loud: *status_dev is OR-ed with 1 (never cleared here: the host zeroes it when it sets the
 *                      exchange up) and the whole step is poisoned — out[0 .. f_off) and the cost become NaN — so that pushes that
 *                      may be partial cannot pass for an evaluation; a flag that carries the poison bit does the same              */
#define HIPNLP_IPC_HANDLE_BYTES 64
int hipnlp_ipc_alloc(size_t bytes, int device, void** dev_ptr, void* handle_out /*[HIPNLP_IPC_HANDLE_BYTES]*/);
int hipnlp_ipc_open(const void* handle /*[HIPNLP_IPC_HANDLE_BYTES]*/, int device, void** dev_ptr);
int hipnlp_ipc_close(void* dev_ptr);
int hipnlp_ipc_free(void* dev_ptr);
int hipnlp_peer_push(const double* shard_dev, const int64_t* dst_dev, int64_t count, double* const* peer_out_dev, int world, void* stream);
int hipnlp_peer_signal(unsigned long long* const* peer_flags_dev, int world, int rank, unsigned long long seq, void* stream);
int hipnlp_peer_signal_checked(unsigned long long* const* peer_flags_dev, int world, int rank, unsigned long long seq, const int* status_dev,
                               void* stream);
int hipnlp_peer_wait(const unsigned long long* flags_dev, int world, unsigned long long seq, double* out_dev, int64_t f_off,
                     int* status_dev, void* stream);
/* The push folded into the evaluation: the knot kernel of a shard handle (batch 1, at most 256 knots in the shard) stores the shard's
 * grad f / jac g / g entries — at their FINAL positions — and the shard's cost straight into the buffer of every rank,
 *     peer_out[r] = [grad f (n) | jac g (nnz) | g (m) | cost partial of rank 0 .. world-1 | f]      (n, m, nnz of the whole problem),
 * the layout hipnlp_peer_push fills (f_off = n + nnz + m for hipnlp_peer_wait): the transfers over the links overlap with the knot
 * programs still running, and neither a push nor a reassembly launch follows.  peer_out_dev: DEVICE array of `world` device-visible
 * base addresses (hipnlp_ipc_alloc / hipnlp_ipc_open) — the buffers this call stores into: every rank's, or rank 0's alone
 * (world = 1: gather_to_root, one consumer); rank: the slot of this shard's cost partial (< 64; the buffers hold one slot per rank
 * of the job, whatever `world` is here).  Then hipnlp_peer_signal and hipnlp_peer_wait as above.  Enqueued on `stream`
 * (NULL: the handle's own stream), not synchronised; the non-finite flag of the handle works as with hipnlp_eval_device. */
int hipnlp_eval_device_peers(hipnlp_handle* h, const double* x_dev, double* const* peer_out_dev, int world, int rank, void* stream);

/* ---- Exchanges without the constants of jac g (multi-GPU) ---------------------------------------------------------------
 * 43 % of the Jacobian pattern of the kinodynamic NLP never changes between callbacks: the +-1 / -dt/2 / mass entries of the linear rows the
 * transcription emits (trapezoid defects and x_0 rows: /root/reference/src/hippopt/base/multiple_shooting_solver.py:713-742,
 * integrators/implicit_trapezoid.py:24-39).  An exchange between GPUs — the bound of the knot-sharded path — has no reason to move
 * them every step: on a handle created with HIPNLP_FLAG_JAC_VARYING_FIRST the shards hand over the VARYING RUN of every knot block
 * only, and the consumer's buffer (the reassembled array on every rank, rank 0's gather buffer, a shared host sink) holds the constant
 * entries, put there once per parameter set by hipnlp_fill_jac_constants.  All of these need HIPNLP_FLAG_JAC_VARYING_FIRST
 * (HIPNLP_E_UNSUPPORTED otherwise) and hipnlp_set_params first; device pointers, enqueued on `stream` (NULL: the handle's own), not
 * synchronised.
 *   hipnlp_jac_vary_layout         out[0..2] varying entries of a first / interior / last knot block, out[3] of the whole horizon,
 *                                  out[4] of the knots in front of this handle's first knot, out[5] of this handle's knots: a COMPACT
 *                                  destination lists the varying runs of the knot blocks behind one another (knot k's run at
 *                                  out[0] + (k - 1) out[1] for k >= 1), entry i of a run being entry i of the knot's block in the pattern
 *                                  (hipnlp_sparsity; hipnlp_jac_constant_mask is 0 exactly there)
 *   hipnlp_fill_jac_constants      the constant entries (values under the parameters last set) into jac_dev [batch][nnz], a complete
 *                                  value array in the handle's pattern order: of the handle's own knots, or — whole_horizon != 0 — of every
 *                                  knot and the horizon-global columns (a shard handle knows the whole pattern: the reassembled buffer
 *                                  of a rank is filled by that rank).  Again after every hipnlp_set_params that changes dt or the mass.
 *   hipnlp_eval_device_vary        hipnlp_eval_device with a compact jac destination: jac_vary_dev [batch][out[3]]
 *   hipnlp_eval_device_shard_vary  hipnlp_eval_device_shard with a compact jac shard: jac_vary_shard [batch][out[5]] — and a compact
 *                                  staging of g: g_stage [batch][shard_g_rows] lists, knot behind knot, the rows each knot owns in the
 *                                  order of its staging slots (the entries >= 0 of hipnlp_stage_rows(k), in slot order), instead of
 *                                  HIPNLP_G_STAGE slots per knot of which half are unused
 *   hipnlp_eval_device_peers_vary  hipnlp_eval_device_peers storing the varying runs only, at their places in the pattern, into buffers
 *                                  whose jac part holds the constants (hipnlp_fill_jac_constants(..., whole_horizon = 1) on the buffer's
 *                                  owner)
 *   hipnlp_reassemble_scatter      hipnlp_reassemble with a destination index: out[dst[i]] = gathered[src[i]], i < count — entries of
 *                                  `out` no shard sends (the constants) are left alone                                                 */
int hipnlp_jac_vary_layout(const hipnlp_handle* h, int64_t* out /*[6]*/);
int hipnlp_fill_jac_constants(hipnlp_handle* h, double* jac_dev, int whole_horizon, void* stream);
int hipnlp_eval_device_vary(hipnlp_handle* h, const double* x_dev, double* f_dev, double* grad_dev, double* g_dev, double* jac_vary_dev, void* stream);
int hipnlp_eval_device_shard_vary(hipnlp_handle* h, const double* x_dev, double* f_dev, double* grad_shard, double* g_stage, double* jac_vary_shard,
                                  void* stream);
int hipnlp_eval_device_peers_vary(hipnlp_handle* h, const double* x_dev, double* const* peer_out_dev, int world, int rank, void* stream);
int hipnlp_reassemble_scatter(const double* gathered_dev, const int64_t* src_dev, const int64_t* dst_dev, double* out_dev, int64_t count, int world,
                              int64_t shard_len, double* f_out_dev, void* stream);

/* ---- One caller, several devices ---------------------------------------------------------------------------------------------------------
 * The reference has ONE caller of the callbacks: IPOPT, in the process that runs self._solver.solve()
 * (/root/reference/src/hippopt/base/opti_solver.py:479); x lives in that process's memory and f, grad f, g, jac g are wanted in arrays of
 * its own.  hipnlp_multi_create serves that caller from SEVERAL devices behind the host-buffer entry points above: the handle it returns
 * is used exactly like one of hipnlp_create — hipnlp_set_params, hipnlp_bounds, hipnlp_sparsity, hipnlp_eval, hipnlp_eval_pinned,
 * hipnlp_eval_hess / _at, hipnlp_hess_nnz / _sparsity, hipnlp_cost_terms, the hipnlp_set_* switches of the host path, hipnlp_host_stats,
 * the hipnlp_ipopt_* callbacks, hipnlp_destroy — with the same results:
 *   - the horizon is cut into n_devices contiguous knot ranges, the first (horizon mod n_devices) of them one knot longer; the rows of the
 *     trapezoid defect of interval k -> k + 1 belong to the owner of knot k + 1 (the naming of base/multiple_shooting_solver.py:713-742),
 *     so a shard READS the record of the knot in front of its first one (the halo) besides its own, the six horizon-global variables, and
 *     the record of the other horizon end where it owns knot 0 or knot N - 1 (periodicity rows); one shard handle per entry of devices[]
 *     (a device may be named more than once: its shards then share it — how the path is tested on one card);
 *   - an evaluation copies x ONCE into a pinned block that the kernels of every device read over their own link (big batches: each shard
 *     copies the records it reads into its own HBM first, on its own stream), launches every shard's kernel — which stores the shard's
 *     entries of grad f, g and jac g AT THEIR FINAL PLACES in the caller's arrays (registered: hipnlp_host_register, or by the handle at
 *     their second sight) or in the handle's pinned block, over that device's own link — and waits for all of them.  No data moves
 *     between the devices and nothing is reassembled;
 *   - f and the per-term costs are summed by the caller's thread from the knots' partials (96 B per knot, stored to pinned memory by the
 *     knot workgroups) in the order of the device reduction: bit for bit the f of one handle over the whole horizon, whatever the cut;
 *     grad f, g, jac g and the Hessian values are those of the knot programs, which do not depend on the cut either (a shard small
 *     enough to be resident at once runs the eight-wave instantiation of the same program, hipnlp_multi_info: equal to 1e-13, not bit
 *     for bit, where that differs from the instantiation one handle over the whole horizon would run);
 *   - an output that is NOT brought to the host by its evaluation (hipnlp_set_prefetch) stays in the shards' HBM; grad f and jac g are
 *     fetched from there when a later new_x = 0 call asks (one copy per shard), g — scattered over the constraint blocks — is evaluated again.
 * The device-resident entry points (hipnlp_eval_device*, hipnlp_eval_hess_device, hipnlp_fill_jac_constants, the profiling calls) answer
 * HIPNLP_E_UNSUPPORTED on such a handle: they belong to one device.
 *   hipnlp_multi_create     desc: the whole horizon (knot_begin = knot_end = 0; desc->device is ignored); devices[n_devices]: HIP ordinals
 *   hipnlp_multi_info       *n_shards: in = capacity of the arrays, out = shards of the handle (0: a plain handle); devices / knot_begin /
 *                           knot_end / waves (4 or 8: the instantiation of the knot kernel the shard launches) may be NULL
 *   hipnlp_multi_breakdown  us[n_shards][2]: host clock of the last evaluation from the start of the launch loop — shard i enqueued,
 *                           shard i seen complete (the shards are waited for in order).  Diagnostic.
 * MEASURED on one card only (several shards of one device, tests/test_gpu_multi.py, bench.py `one_caller`); with one device per shard:
 * unmeasured in this repository. */
int hipnlp_multi_create(const hipnlp_desc* desc, const int32_t* devices, int n_devices, hipnlp_handle** out);
/* The cut and what of x a shard reads — pure arithmetic, no device (what hipnlp_multi_create follows): knots [*knot_begin, *knot_end) of
 * shard `shard` of n_shards, and x_ranges[4][2] = {offset, count} in doubles of one trajectory's x: [0] the shard's records with the halo
 * record in front of them, [1] the six horizon-global variables, [2] the last knot's record for the owner of knot 0, [3] the first knot's
 * record for the owner of the last knot (count 0: not read).  A shard's kernel reads x NOWHERE else: a caller that keeps x in pieces
 * (or a transport that ships it) needs to deliver exactly these. */
int hipnlp_multi_plan(int horizon, int n_shards, int shard, int32_t* knot_begin, int32_t* knot_end, int64_t* x_ranges /*[4][2]*/);
int hipnlp_multi_info(const hipnlp_handle* h, int32_t* n_shards, int32_t* devices, int32_t* knot_begin, int32_t* knot_end, int32_t* waves);
int hipnlp_multi_breakdown(const hipnlp_handle* h, double* us);
/* One launching thread per shard.  Enqueueing a kernel costs the host 4 - 6 us and waiting for its stream a few more: from the caller's
 * thread, one shard after the other, shard i starts ~i x 10 us behind shard 0 — with eight shards longer than the callback of ONE device.
 * on = 1: shards 1 .. n-1 get a worker thread bound to their device; an evaluation publishes its job, the caller's thread does shard 0's
 * part and waits for the workers.  A worker polls for the next job for spin_us microseconds after its last one (IPOPT's callbacks come in
 * bursts, microseconds apart: default 200; < 0: leave as it is) and then sleeps until woken (IPOPT factorises for milliseconds between
 * bursts: no core is burnt meanwhile, the first callback of a burst pays one wake-up).  on = 0: the caller's thread launches every shard,
 * then waits for every shard.  hipnlp_multi_create turns the threads ON for handles of more than one shard.  Values are the same either way. */
int hipnlp_multi_set_threads(hipnlp_handle* h, int on, double spin_us);

/* Per-named-cost values of the last evaluation (Output.cost_values, base/problem.py:28-56):
 * values[batch][HIPNLP_NCOST_TERMS], summed over knots, in the order of hipnlp_cost_term_name(). */
#define HIPNLP_NCOST_TERMS 12
int hipnlp_cost_terms(hipnlp_handle* h, double* values);
const char* hipnlp_cost_term_name(int i);

/* Row-block directory: name (reference constraint base name), first row, rows per knot, first knot, knots — in the numbering of the
 * FULL subject_to list (m_full rows), whether or not the handle was created with DETECT_SIMPLE_BOUNDS. */
int hipnlp_num_row_blocks(const hipnlp_handle* h);
int hipnlp_row_block(const hipnlp_handle* h, int i, const char** name,
                     int32_t* first_row, int32_t* rows_per_knot, int32_t* first_knot, int32_t* n_knots);

/* Kernel time (ms) of the last TIMED evaluation, measured with hipEvents on the stream the kernels ran on; blocks until
 * that launch has finished.  hipnlp_eval() launches are always timed; hipnlp_eval_device*() launches only when selected by an
 * armed profile (an event record drains the stream and costs microseconds: it must not sit around every device-path launch). */
int hipnlp_last_kernel_ms(hipnlp_handle* h, float* ms);

/* HIP-event timing over a region of launches (bench.py roofline leg).  profile_begin arms up to max_launches event triples;
 * every stride-th launch from then on is bracketed (before the knot kernel, after it, after the reduction kernel) on the
 * launch stream.  profile_end synchronises and returns the mean durations (ms) and the number of timed launches.   */
int hipnlp_profile_begin(hipnlp_handle* h, int max_launches, int stride);
int hipnlp_profile_end(hipnlp_handle* h, double* mean_knot_kernel_ms, double* mean_launch_ms, int* count);
/* Runs instead of single launches: one event before the first and one after the last of every run_len consecutive launches (nothing
 * in between); profile_end then returns the mean duration per launch of the runs (dispatch gaps between the launches of a run
 * included) and the number of launches measured.  For launches so short that an event pair around each would dominate. */
int hipnlp_profile_begin_runs(hipnlp_handle* h, int max_runs, int run_len);
/* Kernel launches behind one device-path evaluation: 1 (trajectories of at most 256 knots in launches of at most 32768 knots: the
 * cost is summed inside the knot launch by one reducer workgroup per trajectory) or 2 (otherwise: knot kernel + cost reduction
 * kernel). */
int hipnlp_kernels_per_eval(const hipnlp_handle* h);

/* =====================================================================================================================
 * Static pose finder (BASELINE config 2): the single-knot NLP of
 *   src/hippopt/turnkey_planners/humanoid_pose_finder/planner.py:323-399 (Planner.__init__) with
 *   _add_contact_point_feasibility :670-722, _add_contact_kinematic_consistency :631-668,
 *   _add_kinematics_constraints :446-519, _add_kinematics_regularization :521-629, _add_foot_regularization :724-768.
 * One handle evaluates `batch` independent poses per launch (one workgroup per pose).
 *
 *   x [81]   reference creation order: per contact point c (left[0..3], right[0..3]): p (3), f (3); base position (3),
 *            base quaternion xyzw (4), joint positions (23), com (3)
 *   p [202]  reference creation order: descriptors 8x3 | mass | parametric_link_length_multipliers (1) |
 *            parametric_link_densities (1) | gravity 6 | references.state: per point p,f,descriptor (9 each) |
 *            base position 3 | base quaternion 4 | joints 23 | com 3 | references.frame_quaternion_xyzw 4 |
 *            left_hand_position 3 | right_hand_position 3 | relaxed_complementarity_epsilon | static_friction |
 *            maximum_joint_positions 23 | minimum_joint_positions 23 | left/right_hand_position_in_frame 3+3
 *   g        subject_to call order: per point {complementarity, height, normal, friction, kinematics_consistency(3)},
 *            unitary_quaternion, com_kinematics_consistency(3), centroidal_momentum_dynamics(6),
 *            joint_position_bounds(23), then the optional sumsqr(.) == 0 rows of com / point positions in subject_to mode
 *            ... and of the left / right hand position (three rows each) in subject_to mode
 *   f        base_quaternion_error, frame_rotation_error, com_position_error, joint_positions_error, per foot the
 *            force average / point position / force regularisations, left / right hand position error (minimize mode)
 * Not built (HIPNLP_E_INVALID): the parametric-link model (adam.parametric; parametric_link_names is None in
 * humanoid_pose_finder/main.py).
 * ===================================================================================================================== */
#define HIPNLP_POSE_NX 81
#define HIPNLP_POSE_NP 202
#define HIPNLP_POSE_NCOST_TERMS 9

typedef struct hipnlp_pose_settings {
    int32_t terrain;                    /* settings.terrain: HIPNLP_TERRAIN_*                       (planner.py:80) */
    int32_t n_terrain_steps;
    hipnlp_terrain_step terrain_steps[HIPNLP_MAX_TERRAIN_STEPS];
    int32_t com_position_type;          /* settings.com_position_expression_type          (planner.py:566-573) */
    int32_t left_point_position_type;   /* settings.left_point_position_expression_type   (planner.py:385-390) */
    int32_t right_point_position_type;  /* settings.right_point_position_expression_type  (planner.py:391-396) */
    int32_t reserved_;
    double base_quaternion_cost_multiplier;
    double desired_frame_quaternion_cost_multiplier;
    double com_regularization_cost_multiplier;
    double joint_regularization_cost_weights[HIPNLP_NJ];
    double joint_regularization_cost_multiplier;
    double force_regularization_cost_multiplier;
    double average_force_regularization_cost_multiplier;
    double point_position_regularization_cost_multiplier;
    /* hand position expressions (planner.py:596-660; a zeroed tail = the settings' default: both skipped):
     *   position of the point hand_position_in_frame (parameter p[196..201]) of the frame settings.left/right_hand_frame_name
     *   == references.left/right_hand_position (p[142..147]);  subject_to: three rows "left_hand_position_error" /
     *   "right_hand_position_error" behind the rows above, in Opti's canonical form of `expression == parameter`: g = the position,
     *   lbg = ubg = the reference;  minimize: multiplier * sumsqr(position - reference)  (base/problem.py:95-174).
     *   The frame is given as in hipnlp_robot_model: the link it is rigidly attached to and link_T_frame. */
    int32_t hand_type[2];               /* settings.left/right_hand_expression_type: HIPNLP_EXPR_*  (planner.py:88, 91) */
    int32_t hand_frame_link[2];
    double hand_frame_R[2][9];
    double hand_frame_o[2][3];
    double hand_regularization_cost_multiplier[2];   /* settings.left/right_hand_regularization_cost_multiplier */
} hipnlp_pose_settings;

typedef struct hipnlp_pose_desc {
    hipnlp_pose_settings settings;
    hipnlp_robot_model model;
    int32_t batch;    /* independent poses per call (>= 1) */
    int32_t device;
    int32_t abi_version;   /* HIPNLP_ABI_VERSION (checked by hipnlp_pose_create) */
    int32_t flags;         /* none defined: 0 */
} hipnlp_pose_desc;

typedef struct hipnlp_pose_dims { int32_t n, m, nnz, np; } hipnlp_pose_dims;
typedef struct hipnlp_pose_handle hipnlp_pose_handle;

/* Same contracts as the hipnlp_* functions of the same name above (shapes with n = 81, np = 202). */
int hipnlp_pose_create(const hipnlp_pose_desc* desc, hipnlp_pose_handle** out);
void hipnlp_pose_destroy(hipnlp_pose_handle* h);
const char* hipnlp_pose_last_error(const hipnlp_pose_handle* h);
int hipnlp_pose_get_dims(const hipnlp_pose_handle* h, hipnlp_pose_dims* out);
int hipnlp_pose_set_params(hipnlp_pose_handle* h, const double* p /*[batch][202]*/);
int hipnlp_pose_bounds(const hipnlp_pose_handle* h, double* lbg, double* ubg /*[batch][m]: bounds of every pose*/);
int hipnlp_pose_sparsity(const hipnlp_pose_handle* h, int32_t* irow, int32_t* jcol);
int hipnlp_pose_eval(hipnlp_pose_handle* h, const double* x, double* f, double* grad_f, double* g, double* jac);
int hipnlp_pose_eval_device(hipnlp_pose_handle* h, const double* x_dev, double* f_dev, double* grad_dev, double* g_dev,
                            double* jac_dev, void* stream);
/* Exact Hessian of the Lagrangian  obj_factor * f(x) + lambda^T g(x)  (IPOPT eval_h: bool eval_h(n, x, new_x, obj_factor, m,
 * lambda, new_lambda, nele_hess, iRow, jCol, values, ud)).  The reference reaches it through CasADi's nlp_hess_l: the pose finder
 * runs IPOPT with its default exact-Hessian option (humanoid_pose_finder/main.py:101 `casadi_solver_options = {}`,
 * planner.py:334-339 -> base/opti_solver.py:123-125,479).  Lower triangle (irow >= jcol), column major, fixed pattern:
 *   hipnlp_pose_hess_nnz / _sparsity   structure (the `values == NULL` call of eval_h)
 *   hipnlp_pose_eval_hess              host buffers: x [batch][81], obj_factor [batch], lambda [batch][m] -> hess [batch][nnz_h]
 *   hipnlp_pose_eval_hess_device       device pointers, enqueued on `stream`, not synchronised                                  */
int hipnlp_pose_hess_nnz(const hipnlp_pose_handle* h, int32_t* nnz_h);
int hipnlp_pose_hess_sparsity(const hipnlp_pose_handle* h, int32_t* irow, int32_t* jcol);
int hipnlp_pose_eval_hess(hipnlp_pose_handle* h, const double* x, const double* obj_factor, const double* lambda, double* hess);
int hipnlp_pose_eval_hess_device(hipnlp_pose_handle* h, const double* x_dev, const double* obj_factor_dev, const double* lambda_dev,
                                 double* hess_dev, void* stream);
int hipnlp_pose_cost_terms(hipnlp_pose_handle* h, double* values /*[batch][HIPNLP_POSE_NCOST_TERMS]*/);
const char* hipnlp_pose_cost_term_name(int i);
int hipnlp_pose_num_row_blocks(const hipnlp_pose_handle* h);
int hipnlp_pose_row_block(const hipnlp_pose_handle* h, int i, const char** name, int32_t* first_row, int32_t* rows);
/* kernel duration of the last host-buffer call by HIP events — only for calls made with the timing ON (off by default since round 5: the
 * two event records drain the stream around the launch, ~3 us of a 25 us call for one pose) */
int hipnlp_pose_set_host_timing(hipnlp_pose_handle* h, int on);
int hipnlp_pose_last_kernel_ms(hipnlp_pose_handle* h, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* HIPNLP_H */
