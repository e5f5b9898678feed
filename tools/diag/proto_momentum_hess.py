#!/usr/bin/env python3
"""numpy prototype of the hand-derived Hessian of  lambda^T (angular centroidal momentum rows)  against the oracle's AD Hessian.
Derivation notes for knot_hess_body.h (spatial vectors [angular; linear] about the world origin)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hippopt_amd import kinodyn_layout as L  # noqa: E402
from hippopt_amd.kinodyn_settings import periodic_step_settings  # noqa: E402
from hippopt_amd.robot_model import rot_axis_angle, rot_from_quat_xyzw, synthetic_ergocub  # noqa: E402
from hippopt_amd.synthetic import make_workload  # noqa: E402
from oracle_lib import Oracle  # noqa: E402


def sk(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def crm(S):
    out = np.zeros((6, 6))
    out[:3, :3] = sk(S[:3]); out[3:, :3] = sk(S[3:]); out[3:, 3:] = sk(S[:3])
    return out


def crf(S):
    return -crm(S).T


def What(q):   # 3x4: dtheta = What(qhat) dqhat ; omega = What(qhat) qdot
    v, w = q[:3], q[3]
    out = np.zeros((3, 4))
    out[:, :3] = 2.0 * (w * np.eye(3) + sk(v))     # d/dx_v of 2 (q_w x_v - x_v x q_v) = 2 (q_w I + [q_v]x)
    out[:, 3] = -2.0 * v
    return out


def hessian(md, lam, pb, q, qd, s, sd):
    NJ = md.NDoF
    NL = NJ + 1
    M = md.get_total_mass()
    nq = np.linalg.norm(q); qh = q / nq
    Rb = rot_from_quat_xyzw(qh)
    Wh = What(qh)
    omega = Wh @ qd
    R = [Rb]; o = [np.asarray(pb, float)]
    for j in range(NJ):
        par = int(md.parent[j])
        R.append(R[par] @ md.R_fix[j] @ rot_axis_angle(md.axis[j], s[j]))
        o.append(o[par] + R[par] @ md.o_fix[j])
    a = [R[j + 1] @ md.axis[j] for j in range(NJ)]
    S = [np.concatenate([a[j], np.cross(o[j + 1], a[j])]) for j in range(NJ)]
    # velocities (v_b = 0: the rows do not depend on it)
    v = [np.concatenate([omega, -np.cross(omega, pb)])]
    for j in range(NJ):
        v.append(v[int(md.parent[j])] + S[j] * sd[j])
    I = []
    for i in range(NL):
        c = o[i] + R[i] @ md.com[i]
        Ib = R[i] @ md.inertia[i] @ R[i].T
        m = md.mass[i]
        Ii = np.zeros((6, 6))
        Ii[:3, :3] = Ib + m * sk(c) @ sk(c).T; Ii[:3, 3:] = m * sk(c); Ii[3:, :3] = m * sk(c).T; Ii[3:, 3:] = m * np.eye(3)
        I.append(Ii)
    IC = [Ii.copy() for Ii in I]
    hC = [I[i] @ v[i] for i in range(NL)]
    for j in reversed(range(NJ)):
        par = int(md.parent[j])
        IC[par] += IC[j + 1]; hC[par] += hC[j + 1]
    hO = hC[0]
    P = hO[3:]
    com = np.array([IC[0][2, 4], IC[0][0, 5], IC[0][1, 3]]) / M   # m [c]x block
    mu = -np.asarray(lam) / M
    ell = np.concatenate([mu, np.cross(com, mu)])
    anc = []   # ancestors-or-self (joint indices) of joint j
    for j in range(NJ):
        ch = [j]; l = int(md.parent[j])
        while l > 0:
            ch.append(l - 1); l = int(md.parent[l - 1])
        anc.append(set(ch))
    E = [crf(S[j]) @ hC[j + 1] - IC[j + 1] @ (crm(S[j]) @ v[j + 1]) for j in range(NJ)]
    G = [IC[j + 1] @ S[j] for j in range(NJ)]
    Sxl = [crm(S[j]) @ ell for j in range(NJ)]
    Cv = [IC[j + 1] @ Sxl[j] - crf(S[j]) @ (IC[j + 1] @ ell) for j in range(NJ)]
    w = [crm(S[j]) @ v[j + 1] for j in range(NJ)]
    dc = [G[j][3:] / M for j in range(NJ)]
    dP = [E[j][3:] for j in range(NJ)]
    muP = np.cross(mu, P)
    # base (virtual rotation joints about the world axes through p_b)
    Sb = [np.concatenate([np.eye(3)[m], np.cross(pb, np.eye(3)[m])]) for m in range(3)]
    Eb = [crf(Sb[m]) @ hO - IC[0] @ (crm(Sb[m]) @ v[0]) for m in range(3)]
    Gb = [IC[0] @ Sb[m] for m in range(3)]
    Sxlb = [crm(Sb[m]) @ ell for m in range(3)]
    wb = [crm(Sb[m]) @ v[0] for m in range(3)]
    dcb = [np.cross(np.eye(3)[m], com - pb) for m in range(3)]
    dPb = [Eb[m][3:] for m in range(3)]
    K = np.array([ell @ Gb[m] for m in range(3)])     # dPhi/d omega
    LG = hO[:3] - np.cross(com, P)
    IG = np.array([[np.concatenate([np.eye(3)[r], np.cross(com, np.eye(3)[r])]) @ IC[0] @ np.concatenate([np.eye(3)[c_], np.cross(com, np.eye(3)[c_])])
                    for c_ in range(3)] for r in range(3)])
    assert np.allclose(K, IG @ mu)

    def d2c(k, j):   # k ancestor-or-self of j
        sub_m = IC[j + 1][3, 3]
        sub_h = np.array([IC[j + 1][2, 4], IC[j + 1][0, 5], IC[j + 1][1, 3]])
        return np.cross(a[k], np.cross(a[j], sub_h - sub_m * o[j + 1])) / M

    Hss = np.zeros((NJ, NJ)); Hssd = np.zeros((NJ, NJ))    # [s_k, s_j], [s_k, sd_l]
    for k in range(NJ):
        for j in range(NJ):
            val = np.cross(dc[k], mu) @ dP[j] + np.cross(dc[j], mu) @ dP[k]
            if k in anc[j]:
                val += -Sxl[k] @ E[j] + w[k] @ Cv[j] + d2c(k, j) @ muP
            elif j in anc[k]:
                val += -Sxl[j] @ E[k] + w[j] @ Cv[k] + d2c(j, k) @ muP
            Hss[k, j] = val
            val = np.cross(dc[k], mu) @ G[j][3:]
            if k in anc[j]:
                val += -Sxl[k] @ G[j]
            elif j in anc[k]:
                val += -Cv[k] @ S[j]
            Hssd[k, j] = val
    # theta (world rotation of the base) and omega blocks
    Hts = np.zeros((3, NJ)); Htsd = np.zeros((3, NJ)); Hws = np.zeros((3, NJ)); Htw = np.zeros((3, 3))
    for m in range(3):
        for j in range(NJ):
            Hts[m, j] = -Sxlb[m] @ E[j] + wb[m] @ Cv[j] + np.cross(dcb[m], mu) @ dP[j] + np.cross(dc[j], mu) @ dPb[m] + np.cross(np.eye(3)[m], dc[j]) @ muP
            Htsd[m, j] = np.cross(dcb[m], mu) @ G[j][3:] - Sxlb[m] @ G[j]
            Hws[m, j] = np.cross(dc[j], mu) @ Gb[m][3:] - Cv[j] @ Sb[m]
        for m2 in range(3):
            Htw[m, m2] = np.cross(dcb[m], mu) @ Gb[m2][3:] - Sxlb[m] @ Gb[m2] - (IC[0] @ ell) @ (crm(Sb[m]) @ Sb[m2])
    g = Wh / nq                                    # columns g_l: dtheta / dq_l
    J = (np.eye(4) - np.outer(qh, qh)) / nq
    dw_dq = -What(qd) @ J                          # omega = -What(qd) qhat
    dw_dqd = Wh
    Hqs = g.T @ Hts + dw_dq.T @ Hws
    Hqsd = g.T @ Htsd
    Hqds = dw_dqd.T @ Hws
    # (q, qd): K . d2 omega / dq dqd  +  g^T Htw dw/dqd
    Hqqd = g.T @ Htw @ dw_dqd
    for l in range(4):
        for l2 in range(4):
            e = np.zeros(4); e[l2] = 1.0
            Hqqd[l, l2] += K @ (-What(e) @ J[:, l])
    # (q, q)
    def norm2(g4):
        gq = g4 @ qh
        return (-(np.outer(g4, qh) + np.outer(qh, g4) + gq * np.eye(4)) + 3.0 * gq * np.outer(qh, qh)) / nq**2

    def qq_machinery(Mw):
        Mm = Mw @ Rb
        trM = np.trace(Mm)
        al = np.array([Mm[2, 1] - Mm[1, 2], Mm[0, 2] - Mm[2, 0], Mm[1, 0] - Mm[0, 1]])
        B = np.zeros((4, 4))
        B[:3, :3] = 2.0 * (Mm + Mm.T) - 4.0 * trM * np.eye(3)
        B[:3, 3] = B[3, :3] = 2.0 * al
        g4 = np.zeros(4)
        g4[:3] = 2.0 * qh[3] * al + B[:3, :3] @ qh[:3]
        g4[3] = 2.0 * qh[:3] @ al
        return J @ B @ J + norm2(g4)

    Hqq = qq_machinery(np.outer(mu, LG) + np.outer(omega, K)) + norm2(-What(qd).T @ K)
    for r in range(4):
        for c_ in range(4):
            Hqq[r, c_] += dw_dq[:, r] @ np.cross(g[:, c_], K) + dw_dq[:, c_] @ np.cross(g[:, r], K)
            Hqq[r, c_] += np.cross(mu, g[:, r]) @ IG @ (np.cross(omega, g[:, c_]) + dw_dq[:, c_]) + np.cross(mu, g[:, c_]) @ IG @ (np.cross(omega, g[:, r]) + dw_dq[:, r])
    return dict(ss=Hss, ssd=Hssd, qs=Hqs, qsd=Hqsd, qds=Hqds, qqd=Hqqd, qq=Hqq)


def main():
    md = synthetic_ergocub()
    st = periodic_step_settings(3, md)
    x, p = make_workload(st, md, 1, 21)
    x, p = x[0], p[0]
    o = Oracle(st, md)
    blocks = {b[0]: b for b in o.row_blocks()}
    name, first, rows, k0, nk = blocks["centroidal_momentum_kinematics_consistency"]
    k = 1
    lam = np.zeros(o.m)
    lam3 = np.array([0.7, -1.3, 0.4])
    lam[first + rows * (k - k0):first + rows * (k - k0) + 3] = lam3
    r, c, v = o.hess(x, p, 0.0, lam)
    H = np.zeros((o.n, o.n)); H[r, c] = v; H = H + np.tril(H, -1).T
    b = 189 * k
    Hk = H[b:b + 189, b:b + 189]
    xk = x[b:b + 189]
    q, qd, s, sd, pb = xk[L.QB:L.QB + 4], xk[L.QD:L.QD + 4], xk[L.S:L.S + 23], xk[L.SD:L.SD + 23], xk[L.PB:L.PB + 3]
    mine = hessian(md, lam3, pb, q, qd, s, sd)
    iq, iqd, isx, isd = np.arange(L.QB, L.QB + 4), np.arange(L.QD, L.QD + 4), np.arange(L.S, L.S + 23), np.arange(L.SD, L.SD + 23)
    ref = dict(ss=Hk[np.ix_(isx, isx)], ssd=Hk[np.ix_(isx, isd)], qs=Hk[np.ix_(iq, isx)], qsd=Hk[np.ix_(iq, isd)], qds=Hk[np.ix_(iqd, isx)],
               qqd=Hk[np.ix_(iq, iqd)], qq=Hk[np.ix_(iq, iq)])
    for key in ref:
        err = np.max(np.abs(mine[key] - ref[key])); sc = np.max(np.abs(ref[key]))
        print("%4s  max|ref| %.3e  max err %.3e" % (key, sc, err))
    rest = Hk.copy()
    idx = np.concatenate([iq, iqd, isx, isd])
    rest[np.ix_(idx, idx)] = 0.0
    print("outside the (q, qd, s, sd) block:", np.max(np.abs(rest)))


if __name__ == "__main__":
    main()
