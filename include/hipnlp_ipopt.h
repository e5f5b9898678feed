/*
 * hipnlp_ipopt.h — IPOPT's C callback quartet (+ eval_h) as compiled symbols of libhipnlp.so.
 *
 * The reference reaches IPOPT through CasADi: opti.solver("ipopt", ...) (src/hippopt/base/opti_solver.py:123-125) and
 * self._solver.solve() (:479); nlpsol's IPOPT plugin then hands IPOPT the callbacks of IpStdCInterface.h, which evaluate CasADi's
 * nlp_f / nlp_grad_f / nlp_g / nlp_jac_g (/ nlp_hess_l) — the hot path.  The functions below ARE those callbacks
 * for the engine: exactly the signatures IpStdCInterface.h declares (Eval_F_CB, Eval_Grad_F_CB, Eval_G_CB, Eval_Jac_G_CB, Eval_H_CB),
 * user_data = the hipnlp_handle* of a handle with batch 1.  A maintainer binds them with
 *
 *     IpoptProblem nlp = CreateIpoptProblem(n, x_L, x_U, m, g_L, g_U, nele_jac, nele_hess, 0,   // C index style
 *                                           hipnlp_ipopt_eval_f, hipnlp_ipopt_eval_g, hipnlp_ipopt_eval_grad_f,
 *                                           hipnlp_ipopt_eval_jac_g, hipnlp_ipopt_eval_h);
 *     IpoptSolve(nlp, x, NULL, &obj, mult_g, mult_x_L, mult_x_U, (UserDataPtr)handle);
 *
 * (INTEGRATION.md has the whole program).  IPOPT's protocol is honoured to the letter: a call with values == NULL returns the fixed
 * sparsity structure (x may be NULL then), new_x == FALSE reuses the evaluation of the previous callback (one kernel evaluation per
 * iterate, whichever callback comes first), an evaluation that produced a NaN / Inf returns FALSE (IPOPT then cuts the step: what it
 * does with CasADi's NaNs), every other failure returns FALSE with the reason in hipnlp_last_error.
 *
 * The types are restated so that this header compiles without IPOPT; they are the ones of a default IPOPT build (IpTypes.h:
 * ipnumber = double, ipindex = int; IpStdCInterface.h: Number, Index, Bool = int, UserDataPtr = void*).  Define
 * HIPNLP_IPOPT_WITH_IPOPT_HEADER to take them from <IpStdCInterface.h> instead (a mismatch is then a compile error).
 */
#ifndef HIPNLP_IPOPT_H
#define HIPNLP_IPOPT_H

#include "hipnlp.h"

#ifdef HIPNLP_IPOPT_WITH_IPOPT_HEADER
#include <IpStdCInterface.h>
/* the callbacks below return and take Bool as an int-sized value (IPOPT <= 3.13: typedef int Bool; 3.14: typedef int Bool unless
 * built otherwise): a build whose Bool has another size must not bind them silently ("IPOPT builds whose Bool is bool") */
#if defined(__cplusplus)
static_assert(sizeof(Bool) == sizeof(int), "hipnlp_ipopt.h: this IPOPT build's Bool is not int-sized; see INTEGRATION.md");
#else
_Static_assert(sizeof(Bool) == sizeof(int), "hipnlp_ipopt.h: this IPOPT build's Bool is not int-sized; see INTEGRATION.md");
#endif
#else
typedef double Number;       /* IpStdCInterface.h: typedef ipnumber Number;  IpTypes.h: typedef double ipnumber */
typedef int Index;           /* IpStdCInterface.h: typedef ipindex Index;    IpTypes.h: typedef int ipindex     */
typedef int Bool;            /* IpStdCInterface.h: typedef int Bool                                           */
typedef void* UserDataPtr;   /* IpStdCInterface.h: typedef void* UserDataPtr                                   */
#ifndef TRUE
#define TRUE (1)
#endif
#ifndef FALSE
#define FALSE (0)
#endif
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* Eval_F_CB */
Bool hipnlp_ipopt_eval_f(Index n, Number* x, Bool new_x, Number* obj_value, UserDataPtr user_data);
/* Eval_Grad_F_CB */
Bool hipnlp_ipopt_eval_grad_f(Index n, Number* x, Bool new_x, Number* grad_f, UserDataPtr user_data);
/* Eval_G_CB */
Bool hipnlp_ipopt_eval_g(Index n, Number* x, Bool new_x, Index m, Number* g, UserDataPtr user_data);
/* Eval_Jac_G_CB: values == NULL -> iRow / jCol (0-based, CCS order: sorted by column, then row); else values */
Bool hipnlp_ipopt_eval_jac_g(Index n, Number* x, Bool new_x, Index m, Index nele_jac, Index* iRow, Index* jCol, Number* values,
                             UserDataPtr user_data);
/* Eval_H_CB: lower triangle (iRow >= jCol) of obj_factor * hess f + sum_r lambda_r hess g_r; values == NULL -> structure.
 * Only when the solve does not run hessian_approximation = limited-memory (the kinodynamic scripts do, main_periodic_step.py:116;
 * the pose finder does not, humanoid_pose_finder/main.py:101). */
Bool hipnlp_ipopt_eval_h(Index n, Number* x, Bool new_x, Number obj_factor, Index m, Number* lambda, Bool new_lambda, Index nele_hess,
                         Index* iRow, Index* jCol, Number* values, UserDataPtr user_data);

/* What CreateIpoptProblem needs.  Sizes of the handle's NLP (the reduced one for a handle created with
 * HIPNLP_FLAG_DETECT_SIMPLE_BOUNDS); nele_hess may be NULL (limited-memory runs pass 0 to CreateIpoptProblem). */
int hipnlp_ipopt_sizes(hipnlp_handle* h, Index* n, Index* m, Index* nele_jac, Index* nele_hess);
/* Bounds in IPOPT's convention: an infinite bound is -/+ 2e19 (beyond nlp_lower_bound_inf / nlp_upper_bound_inf = -/+ 1e19).
 * Valid after hipnlp_set_params.  Any pointer may be NULL. */
int hipnlp_ipopt_bounds(hipnlp_handle* h, Number* x_L, Number* x_U, Number* g_L, Number* g_U);
/* Optional, around IpoptSolve.  attach: auto-registration of IPOPT's arrays (the g / jac-value buffers of the TNLPAdapter and the
 * gradient vector become direct kernel outputs at their second sight, verified at every use) and NOTHING that writes into an array
 * before IPOPT hands it to a callback: eval_f at a new x brings f, grad f and g to the library's pinned block (eval_g / eval_grad_f at
 * that x are host copies), eval_jac_g with new_x = FALSE fetches the Jacobian from HBM into IPOPT's array — only its varying entries
 * when the handle was created with HIPNLP_FLAG_JAC_VARYING_FIRST (recommended for IPOPT: triplets may come in any order).
 * detach releases the registrations (call it before the arrays IPOPT owned are freed, i.e. before FreeIpoptProblem). */
int hipnlp_ipopt_attach(hipnlp_handle* h);
int hipnlp_ipopt_detach(hipnlp_handle* h);
/* Early outputs for g and jac g (hipnlp_set_early_outputs(h, 1)) — OPT-IN, off after attach.  The first callback at a new x then also
 * fills the arrays earlier eval_g / eval_jac_g calls passed: one PCIe transfer per iterate instead of two (47 against 70 us per
 * 100-knot iterate).  It rests on an ASSUMPTION about IPOPT that this repository could not test (IPOPT is not in its
 * image; only a C replay of the documented call order was run): that the adapter's g / jac buffers are scratch between callbacks.
 * IPOPT's TNLPAdapter keeps them as CACHES keyed by the tag of x (x_tag_for_g_, x_tag_for_jac_g_): Eval_c and Eval_d (Eval_jac_c and
 * Eval_jac_d) of one x share one evaluation.  If a build of IPOPT evaluates another point between the two — eval_f at a trial point
 * between Eval_c(x1) and Eval_d(x1) — the early store has replaced the cached values and IPOPT would use g or jac g of the wrong point
 * without any error.  Turn this on only after checking the iterates against a run with it off on the IPOPT build at hand (identical
 * iterates = the interleaving does not occur there, including restoration and watchdog phases).  grad f is never written early: its
 * destination is IPOPT's own gradient vector. */
int hipnlp_ipopt_set_early_outputs(hipnlp_handle* h, int on);

#ifdef __cplusplus
}
#endif
#endif /* HIPNLP_IPOPT_H */
