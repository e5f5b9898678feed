/* harness.c — TEST ONLY: drives libhipnlp.so exactly the way IPOPT's C interface does, without IPOPT (absent from the image).
 *
 * 1. The callback typedefs below are IpStdCInterface.h's (Ipopt 3.14), spelled out; the engine's functions are ASSIGNED to variables
 *    of those types, so a signature that IPOPT could not bind is a compile error here (-Werror).
 * 2. The run replays IPOPT's call protocol against a recorded problem: structure calls with values == NULL (and x == NULL), the
 *    starting point (gradient first), then trial points — eval_f(new_x = TRUE), eval_g(new_x = FALSE) — of which the accepted ones
 *    go on to eval_grad_f / eval_jac_g / eval_h with new_x = FALSE, the rejected ones do not; g and jac values always in the SAME two
 *    arrays (IPOPT's TNLPAdapter keeps full_g_ / jac_g_ for the whole solve), grad f alternating between two; a last point that
 *    evaluates to NaN must return FALSE.  Every returned Bool and every value goes to the output file; tests/test_ipopt_binding.py
 *    compares them with the CPU oracle.
 * 3. A timing loop: IPOPT iterates as four C calls, microseconds per iterate on stdout.
 *
 * usage: harness <input.bin> <output.bin> [timing iterations]
 * input : int32 magic 0x49504F54, int32 desc_bytes, int32 np, int32 points, int32 attach, int32 shards (0: hipnlp_create; > 0:
 *         hipnlp_multi_create with that many shards, all on desc.device — one caller, several shard handles behind the same callbacks),
 *         desc bytes, p[np], x[points][n], lambda[m], obj_factor                                     (doubles little endian)
 * output: records { int32 kind (0 f, 1 grad, 2 g, 3 jac, 4 hess, 5 jac structure, 6 hess structure, 7 bounds), int32 point, int32 ok,
 *                   int32 count, double values[count] }
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "hipnlp_ipopt.h"

/* ---- IpStdCInterface.h, callback function types (Number = double, Index = int, Bool = int, UserDataPtr = void*) ---- */
typedef Bool (*Eval_F_CB)(Index n, Number* x, Bool new_x, Number* obj_value, UserDataPtr user_data);
typedef Bool (*Eval_Grad_F_CB)(Index n, Number* x, Bool new_x, Number* grad_f, UserDataPtr user_data);
typedef Bool (*Eval_G_CB)(Index n, Number* x, Bool new_x, Index m, Number* g, UserDataPtr user_data);
typedef Bool (*Eval_Jac_G_CB)(Index n, Number* x, Bool new_x, Index m, Index nele_jac, Index* iRow, Index* jCol, Number* values,
                              UserDataPtr user_data);
typedef Bool (*Eval_H_CB)(Index n, Number* x, Bool new_x, Number obj_factor, Index m, Number* lambda, Bool new_lambda, Index nele_hess,
                          Index* iRow, Index* jCol, Number* values, UserDataPtr user_data);

static Eval_F_CB eval_f = hipnlp_ipopt_eval_f;
static Eval_Grad_F_CB eval_grad_f = hipnlp_ipopt_eval_grad_f;
static Eval_G_CB eval_g = hipnlp_ipopt_eval_g;
static Eval_Jac_G_CB eval_jac_g = hipnlp_ipopt_eval_jac_g;
static Eval_H_CB eval_h = hipnlp_ipopt_eval_h;

static FILE* out;
static void record(int kind, int point, int ok, int count, const double* v) {
    int head[4];
    head[0] = kind; head[1] = point; head[2] = ok; head[3] = count;
    fwrite(head, sizeof(int), 4, out);
    if (count > 0) fwrite(v, sizeof(double), (size_t)count, out);
}
static void record_indices(int kind, int ok, int count, const Index* a, const Index* b) {
    double* v = (double*)malloc(sizeof(double) * 2 * (size_t)count);
    for (int i = 0; i < count; ++i) { v[i] = a[i]; v[count + i] = b[i]; }
    record(kind, -1, ok, 2 * count, v);
    free(v);
}
static double now_us(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return 1e6 * (double)t.tv_sec + 1e-3 * (double)t.tv_nsec;
}
static void* must(void* p) { if (!p) { fprintf(stderr, "out of memory\n"); exit(3); } return p; }

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: harness <input.bin> <output.bin> [timing iterations]\n"); return 2; }
    const int timing = argc > 3 ? atoi(argv[3]) : 0;
    FILE* in = fopen(argv[1], "rb");
    if (!in) { perror(argv[1]); return 2; }
    int head[6];
    if (fread(head, sizeof(int), 6, in) != 6 || head[0] != 0x49504F54) { fprintf(stderr, "bad input file\n"); return 2; }
    const int desc_bytes = head[1], np = head[2], points = head[3], attach = head[4], shards = head[5];
    if (desc_bytes != (int)sizeof(hipnlp_desc)) { fprintf(stderr, "descriptor of %d bytes, header says %zu\n", desc_bytes, sizeof(hipnlp_desc)); return 2; }
    hipnlp_desc desc;
    if (fread(&desc, 1, sizeof desc, in) != sizeof desc) return 2;
    double* p = (double*)must(malloc(sizeof(double) * (size_t)np));
    if (fread(p, sizeof(double), (size_t)np, in) != (size_t)np) return 2;

    hipnlp_handle* h = NULL;
    if (shards > 0) {   /* IPOPT's process drives several shard handles through the same five callbacks */
        int32_t devices[64];
        if (shards > 64) { fprintf(stderr, "at most 64 shards\n"); return 2; }
        for (int i = 0; i < shards; ++i) devices[i] = desc.device;
        if (hipnlp_multi_create(&desc, devices, shards, &h) != HIPNLP_OK) { fprintf(stderr, "hipnlp_multi_create: %s\n", hipnlp_last_error(NULL)); return 1; }
    } else if (hipnlp_create(&desc, &h) != HIPNLP_OK) { fprintf(stderr, "hipnlp_create: %s\n", hipnlp_last_error(NULL)); return 1; }
    if (hipnlp_set_params(h, p) != HIPNLP_OK) { fprintf(stderr, "hipnlp_set_params: %s\n", hipnlp_last_error(h)); return 1; }
    Index n = 0, m = 0, nele_jac = 0, nele_hess = 0;
    if (hipnlp_ipopt_sizes(h, &n, &m, &nele_jac, &nele_hess) != HIPNLP_OK) { fprintf(stderr, "sizes: %s\n", hipnlp_last_error(h)); return 1; }
    double* xs = (double*)must(malloc(sizeof(double) * (size_t)points * (size_t)n));
    double* lambda = (double*)must(malloc(sizeof(double) * (size_t)m));
    double obj_factor = 1.0;
    if (fread(xs, sizeof(double), (size_t)points * (size_t)n, in) != (size_t)points * (size_t)n) return 2;
    if (fread(lambda, sizeof(double), (size_t)m, in) != (size_t)m || fread(&obj_factor, sizeof(double), 1, in) != 1) return 2;
    fclose(in);
    out = fopen(argv[2], "wb");
    if (!out) { perror(argv[2]); return 2; }

    /* IPOPT's own arrays: g and the Jacobian values live for the whole solve, gradients come and go */
    double* g = (double*)must(malloc(sizeof(double) * (size_t)m));
    double* jac = (double*)must(malloc(sizeof(double) * (size_t)nele_jac));
    double* grad[2];
    grad[0] = (double*)must(malloc(sizeof(double) * (size_t)n));
    grad[1] = (double*)must(malloc(sizeof(double) * (size_t)n));
    double* hess = (double*)must(malloc(sizeof(double) * (size_t)nele_hess));
    Index* iRow = (Index*)must(malloc(sizeof(Index) * (size_t)nele_jac));
    Index* jCol = (Index*)must(malloc(sizeof(Index) * (size_t)nele_jac));
    Index* hRow = (Index*)must(malloc(sizeof(Index) * (size_t)nele_hess));
    Index* hCol = (Index*)must(malloc(sizeof(Index) * (size_t)nele_hess));
    UserDataPtr ud = (UserDataPtr)h;

    {   /* bounds as CreateIpoptProblem wants them */
        double* b = (double*)must(malloc(sizeof(double) * 2 * (size_t)(n + m)));
        const int ok = hipnlp_ipopt_bounds(h, b, b + n, b + 2 * n, b + 2 * n + m) == HIPNLP_OK;
        record(7, -1, ok, 2 * (n + m), b);
        free(b);
    }
    /* attach: 0 = plain handle, 1 = hipnlp_ipopt_attach (nothing is written into an array before IPOPT hands it to a callback),
     * 2 = attach + the opt-in early outputs for g and jac g */
    if (attach && hipnlp_ipopt_attach(h) != HIPNLP_OK) { fprintf(stderr, "attach: %s\n", hipnlp_last_error(h)); return 1; }
    if (attach == 2 && hipnlp_ipopt_set_early_outputs(h, 1) != HIPNLP_OK) { fprintf(stderr, "early outputs: %s\n", hipnlp_last_error(h)); return 1; }

    /* structure calls: values == NULL, x == NULL */
    record_indices(5, eval_jac_g(n, NULL, FALSE, m, nele_jac, iRow, jCol, NULL, ud), nele_jac, iRow, jCol);
    record_indices(6, eval_h(n, NULL, FALSE, 1.0, m, NULL, FALSE, nele_hess, hRow, hCol, NULL, ud), nele_hess, hRow, hCol);
    /* wrong sizes are refused, not evaluated */
    if (eval_g(n, xs, TRUE, m + 1, g, ud) != FALSE || eval_jac_g(n, xs, TRUE, m, nele_jac - 1, NULL, NULL, jac, ud) != FALSE ||
        eval_f(n + 1, xs, TRUE, &obj_factor, ud) != FALSE) { fprintf(stderr, "a call with wrong sizes was served\n"); return 1; }

    /* IPOPT hands eval_grad_f the storage of ITS OWN gradient vector: the gradient at the current iterate must survive the
     * evaluation of trial points untouched (no early output may land in it) */
    double* grad_kept = (double*)must(malloc(sizeof(double) * (size_t)n));
    const double* grad_current = NULL;
    /* The TNLPAdapter keeps its g / jac-value buffers as CACHES keyed by the tag of x (Eval_c and Eval_d of one x share one
     * evaluation): unless the early outputs were asked for, a callback at ANOTHER x must leave both buffers as their own callbacks
     * filled them */
    double* g_kept = (double*)must(malloc(sizeof(double) * (size_t)m));
    double* jac_kept = (double*)must(malloc(sizeof(double) * (size_t)nele_jac));
    int g_valid = 0, jac_valid = 0;
    for (int i = 0; i < points; ++i) {
        double* x = xs + (size_t)i * (size_t)n;
        double f = 0.0;
        int ok;
        if (i == 0) {   /* the starting point: derivatives first */
            ok = eval_grad_f(n, x, TRUE, grad[0], ud);  record(1, i, ok, n, grad[0]);
            grad_current = grad[0];  memcpy(grad_kept, grad[0], sizeof(double) * (size_t)n);
            ok = eval_jac_g(n, x, FALSE, m, nele_jac, NULL, NULL, jac, ud);  record(3, i, ok, nele_jac, jac);
            memcpy(jac_kept, jac, sizeof(double) * (size_t)nele_jac);  jac_valid = 1;
            ok = eval_f(n, x, FALSE, &f, ud);  record(0, i, ok, 1, &f);
            ok = eval_g(n, x, FALSE, m, g, ud);  record(2, i, ok, m, g);
            memcpy(g_kept, g, sizeof(double) * (size_t)m);  g_valid = 1;
            ok = eval_h(n, x, FALSE, obj_factor, m, lambda, TRUE, nele_hess, NULL, NULL, hess, ud);  record(4, i, ok, nele_hess, hess);
            continue;
        }
        ok = eval_f(n, x, TRUE, &f, ud);  record(0, i, ok, 1, &f);                 /* a line-search trial point */
        if (attach != 2 && ((g_valid && memcmp(g, g_kept, sizeof(double) * (size_t)m) != 0) ||
                            (jac_valid && memcmp(jac, jac_kept, sizeof(double) * (size_t)nele_jac) != 0))) {
            fprintf(stderr, "eval_f at point %d wrote into the g / jac buffers of the previous point (no early outputs were asked for)\n", i);
            return 1;
        }
        ok = eval_g(n, x, FALSE, m, g, ud);  record(2, i, ok, m, g);
        memcpy(g_kept, g, sizeof(double) * (size_t)m);  g_valid = 1;
        if (grad_current && memcmp(grad_current, grad_kept, sizeof(double) * (size_t)n) != 0) {
            fprintf(stderr, "the gradient vector of the current iterate was overwritten while trial point %d was evaluated\n", i);
            return 1;
        }
        if (i % 3 == 2) continue;                                                   /* rejected: IPOPT moves on */
        ok = eval_grad_f(n, x, FALSE, grad[i & 1], ud);  record(1, i, ok, n, grad[i & 1]);
        grad_current = grad[i & 1];  memcpy(grad_kept, grad[i & 1], sizeof(double) * (size_t)n);
        ok = eval_jac_g(n, x, FALSE, m, nele_jac, NULL, NULL, jac, ud);  record(3, i, ok, nele_jac, jac);
        memcpy(jac_kept, jac, sizeof(double) * (size_t)nele_jac);  jac_valid = 1;
        if (i % 2 == 0) { ok = eval_h(n, x, FALSE, obj_factor, m, lambda, TRUE, nele_hess, NULL, NULL, hess, ud);  record(4, i, ok, nele_hess, hess); }
    }

    if (timing > 0) {   /* an IPOPT iterate as four C calls on IPOPT's own arrays, new x every iterate */
        double best = 1e30, per_call[4] = {0.0, 0.0, 0.0, 0.0};   /* per_call: eval_f, eval_g, eval_grad_f, eval_jac_g of the best pass */
        for (int pass = 0; pass < 3; ++pass) {
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
            const double t0 = now_us();
            for (int it = 0; it < timing; ++it) {
                double* x = xs + (size_t)(it % (points - 1)) * (size_t)n;   /* (the last point is the NaN one) */
                double f;
                const double a0 = now_us();
                Bool ok = eval_f(n, x, TRUE, &f, ud);
                const double a1 = now_us();
                ok = ok && eval_g(n, x, FALSE, m, g, ud);
                const double a2 = now_us();
                ok = ok && eval_grad_f(n, x, FALSE, grad[0], ud);
                const double a3 = now_us();
                ok = ok && eval_jac_g(n, x, FALSE, m, nele_jac, NULL, NULL, jac, ud);
                const double a4 = now_us();
                if (!ok) { fprintf(stderr, "timing loop: a callback failed\n"); return 1; }
                acc[0] += a1 - a0; acc[1] += a2 - a1; acc[2] += a3 - a2; acc[3] += a4 - a3;
            }
            const double us = (now_us() - t0) / timing;
            if (us < best) { best = us; for (int c = 0; c < 4; ++c) per_call[c] = acc[c] / timing; }
        }
        double trial = 1e30;
        for (int pass = 0; pass < 3; ++pass) {
            const double t0 = now_us();
            for (int it = 0; it < timing; ++it) {
                double* x = xs + (size_t)(it % (points - 1)) * (size_t)n;
                double f;
                if (!eval_f(n, x, TRUE, &f, ud) || !eval_g(n, x, FALSE, m, g, ud)) return 1;
            }
            const double us = (now_us() - t0) / timing;
            if (us < trial) trial = us;
        }
        long stats[8];
        hipnlp_host_stats(h, stats);
        printf("{\"ipopt_iterate_four_c_calls_us\": %.2f, \"trial_point_two_c_calls_us\": %.2f, \"attach\": %d, \"shards\": %d, "
               "\"auto_registered\": %ld, \"evaluations\": %ld, \"constant_fills\": %ld, \"constant_refills\": %ld, \"constant_entries\": %ld, "
               "\"per_call_us\": [%.2f, %.2f, %.2f, %.2f]}\n",
               best, trial, attach, shards, stats[0], stats[3], stats[4], stats[5], stats[6], per_call[0], per_call[1], per_call[2], per_call[3]);
    }
    if (attach) hipnlp_ipopt_detach(h);
    fclose(out);
    hipnlp_destroy(h);
    return 0;
}
