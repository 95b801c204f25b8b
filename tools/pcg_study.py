#!/usr/bin/env python3
"""north_star names a "PCG solve" of the reduced camera system; the reference solves it directly (SimplicialLDLT,
code/Thirdparty/g2o/g2o/solvers/linear_solver_eigen.h:94-124), and so does libswarmorb (block-skyline Cholesky on FP64 MFMA
tiles).  This tool measures what preconditioned conjugate gradients would need on the systems the product solves, so
that the choice is a measurement (round 4's verdict, item 5):

  CPU part (here, no GPU): the reduced camera system S x = b of a whole-map bundle adjustment - Schur complement of the
  lambda-damped normal equations, block_solver.hpp:354-486 - at a given LM iteration of the oracle's run, assembled in numpy
  from the same Jacobians (types_six_dof_expmap.cpp:103-139); block-Jacobi PCG (6 x 6 diagonal blocks, the preconditioner
  of "Bundle Adjustment in the Large") down to a list of relative residuals; for each: iterations, and the error of the
  pose increment against the direct solve (what decides whether ten LM iterations end within the parity tolerance of
  2e-5 on the pose entries).

  GPU part (tools/pcg_spmv_bench.py): time of one application of S in 6 x 6 block-sparse form + the two dot products of an
  iteration, on the same block structure.

iterations x time per iteration against the direct solve's time (bench.py configs.global_ba) decides.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from swarmmap_amd import synth  # noqa: E402


def quat_pose(T12):
    T = np.asarray(T12, np.float64).reshape(-1, 3, 4)
    return T[:, :, :3], T[:, :, 3]


def linearise(p, Tcw, Xw, lam=None, robust=False):
    """Normal equations of the map at (Tcw, Xw): returns (Hpp blocks [nf, 6, 6], bp [nf, 6], W blocks per edge [E, 6, 3],
    Hll [L, 3, 3], bl [L, 3], hidx per pose, chi2)."""
    R, t = quat_pose(Tcw)
    ep, el = np.asarray(p["edge_pose"]), np.asarray(p["edge_point"])
    intr = np.asarray(p["intr"], np.float64)
    fx, fy, cx, cy = intr[ep, 0], intr[ep, 1], intr[ep, 2], intr[ep, 3]
    X = np.asarray(Xw, np.float64)[el]
    pc = np.einsum("eij,ej->ei", R[ep], X) + t[ep]
    x, y, z = pc[:, 0], pc[:, 1], pc[:, 2]
    z2 = z * z
    obs = np.asarray(p["obs"], np.float64)
    err = obs - np.stack([fx * x / z + cx, fy * y / z + cy], 1)
    w = np.asarray(p["inv_sigma2"], np.float64)
    chi2 = w * (err ** 2).sum(1)
    rho1 = np.ones_like(chi2)
    if robust:
        d2 = 5.991
        big = chi2 > d2
        rho1[big] = np.sqrt(d2) / np.sqrt(chi2[big])
    ww = w * rho1
    E = len(ep)
    Jl = np.zeros((E, 2, 3))
    tmp = np.zeros((E, 2, 3))
    tmp[:, 0, 0] = fx; tmp[:, 0, 2] = -x / z * fx
    tmp[:, 1, 1] = fy; tmp[:, 1, 2] = -y / z * fy
    Jl = -(1.0 / z)[:, None, None] * np.einsum("eij,ejk->eik", tmp, R[ep])
    Jp = np.zeros((E, 2, 6))
    Jp[:, 0, 0] = x * y / z2 * fx; Jp[:, 0, 1] = -(1 + x * x / z2) * fx; Jp[:, 0, 2] = y / z * fx
    Jp[:, 0, 3] = -1.0 / z * fx; Jp[:, 0, 5] = x / z2 * fx
    Jp[:, 1, 0] = (1 + y * y / z2) * fy; Jp[:, 1, 1] = -x * y / z2 * fy; Jp[:, 1, 2] = -x / z * fy
    Jp[:, 1, 4] = -1.0 / z * fy; Jp[:, 1, 5] = y / z2 * fy
    fixed = np.asarray(p["fixed"]) != 0
    touched = np.zeros(len(fixed), bool)
    touched[ep] = True
    free = touched & ~fixed
    hidx = np.full(len(fixed), -1)
    hidx[free] = np.arange(free.sum())
    nf, L = int(free.sum()), len(Xw)
    wr = ww[:, None] * err  # Omega rho' e
    Hll = np.zeros((L, 3, 3)); bl = np.zeros((L, 3))
    np.add.at(Hll, el, ww[:, None, None] * np.einsum("eki,ekj->eij", Jl, Jl))
    np.add.at(bl, el, -np.einsum("eki,ek->ei", Jl, wr))
    eh = hidx[ep]
    act = eh >= 0
    Hpp = np.zeros((nf, 6, 6)); bp = np.zeros((nf, 6))
    np.add.at(Hpp, eh[act], ww[act, None, None] * np.einsum("eki,ekj->eij", Jp[act], Jp[act]))
    np.add.at(bp, eh[act], -np.einsum("eki,ek->ei", Jp[act], wr[act]))
    W = ww[:, None, None] * np.einsum("eki,ekj->eij", Jp, Jl)  # 6 x 3 per edge (pose x point)
    return Hpp, bp, W, Hll, bl, hidx, eh, float(chi2.sum())


def reduced_system(p, Tcw, Xw, lam, robust=False):
    """S (scipy BSR, 6 x 6 blocks, full symmetric) and b of the lambda-damped reduced camera system."""
    import scipy.sparse as sp
    Hpp, bp, W, Hll, bl, hidx, eh, chi2 = linearise(p, Tcw, Xw, robust=robust)
    nf, L = len(Hpp), len(Hll)
    I3, I6 = np.eye(3), np.eye(6)
    Dinv = np.linalg.inv(Hll + lam * I3[None])
    el = np.asarray(p["edge_point"])
    act = np.nonzero(eh >= 0)[0]
    # Y_e = W_e D^-1 (6 x 3), per edge of a free keyframe
    Y = np.einsum("eij,ejk->eik", W[act], Dinv[el[act]])
    b = bp - np.zeros_like(bp)
    np.subtract.at(b, eh[act], np.einsum("eij,ej->ei", Y, bl[el[act]]))
    # S = Hpp + lam I - sum over landmarks of Y_i W_j^T for all pairs of its observers
    order = np.argsort(el[act], kind="stable")
    a_s, l_s = act[order], el[act][order]
    Ys = Y[order]
    Ws = W[a_s]
    hs = eh[a_s]
    start = np.flatnonzero(np.r_[True, l_s[1:] != l_s[:-1]])
    cnt = np.diff(np.r_[start, len(l_s)])
    rows, cols, blocks = [], [], []
    # group landmarks by their number of observers: one einsum per group
    for k in np.unique(cnt):
        g = start[cnt == k]
        idx = g[:, None] + np.arange(k)[None, :]            # [G, k]
        Yg, Wg, hg = Ys[idx], Ws[idx], hs[idx]              # [G, k, 6, 3]
        blk = -np.einsum("gaij,gbkj->gabik", Yg, Wg)        # [G, k, k, 6, 6]
        rows.append(np.repeat(hg[:, :, None], k, 2).reshape(-1))
        cols.append(np.repeat(hg[:, None, :], k, 1).reshape(-1))
        blocks.append(blk.reshape(-1, 6, 6))
    rows = np.concatenate(rows + [np.arange(nf)])
    cols = np.concatenate(cols + [np.arange(nf)])
    blocks = np.concatenate(blocks + [Hpp + lam * I6[None]])
    # sum duplicate (row, col) blocks
    key = rows.astype(np.int64) * nf + cols
    uq, inv = np.unique(key, return_inverse=True)
    acc = np.zeros((len(uq), 6, 6))
    np.add.at(acc, inv, blocks)
    r, c = uq // nf, uq % nf
    indptr = np.zeros(nf + 1, np.int64)
    np.add.at(indptr, r + 1, 1)
    indptr = np.cumsum(indptr)
    S = sp.bsr_matrix((acc, c, indptr), shape=(6 * nf, 6 * nf))
    return S, b.reshape(-1), chi2, len(uq)


def pcg(S, b, Minv_blocks, tols, max_it=5000):
    """Block-Jacobi PCG; returns {tol: (iterations, x)} for the relative residuals |r| / |b| in tols (descending)."""
    nf = len(Minv_blocks)
    apply_M = lambda r: np.einsum("nij,nj->ni", Minv_blocks, r.reshape(nf, 6)).reshape(-1)  # noqa: E731
    x = np.zeros_like(b)
    r = b.copy()
    z = apply_M(r)
    pvec = z.copy()
    rz = r @ z
    nb = np.linalg.norm(b)
    out, todo = {}, sorted(tols, reverse=True)
    for it in range(1, max_it + 1):
        Sp = S @ pvec
        alpha = rz / (pvec @ Sp)
        x += alpha * pvec
        r -= alpha * Sp
        rel = np.linalg.norm(r) / nb
        while todo and rel <= todo[0]:
            out[todo.pop(0)] = (it, x.copy())
        if not todo:
            break
        z = apply_M(r)
        rz_new = r @ z
        pvec = z + (rz_new / rz) * pvec
        rz = rz_new
    for t in todo:
        out[t] = (None, x.copy())
    return out


def study(name, seed, lm_iterations, tols, robust=False):
    import scipy.sparse.linalg as spla
    sys.path.insert(0, os.path.join(ROOT))
    from oracle import oracle_py
    p = synth.make_ba_case(name, seed)
    out = {"map": name, "free_keyframes": None, "edges": int(len(p["edge_pose"])), "points": int(len(p["Xw"])), "robust": robust, "at": []}
    for it in lm_iterations:
        if it <= 1:
            Tcw, Xw = np.asarray(p["Tcw"], np.float64), np.asarray(p["Xw"], np.float64)
            lam_src = None
        else:  # the state after it - 1 iterations of the oracle (float outputs: good enough for a conditioning study)
            r = oracle_py.bundle_adjust(p, its1=it - 1, its2=0, robust=robust, huber_delta=np.float32(np.sqrt(np.float32(5.99))))
            Tcw, Xw = np.asarray(r["Tcw"], np.float64), np.asarray(r["Xw"], np.float64)
            lam_src = r["info"].get("lambda_final")
        Hpp = linearise(p, Tcw, Xw, robust=robust)[0]
        lam0 = 1e-5 * max(Hpp[:, np.arange(6), np.arange(6)].max(), 0.0)  # computeLambdaInit (levenberg.cpp:166-189, tau = 1e-5)
        lam = float(lam_src) if lam_src else lam0
        t0 = time.time()
        S, b, chi2, nnzb = reduced_system(p, Tcw, Xw, lam, robust)
        t_build = time.time() - t0
        nf = S.shape[0] // 6
        out["free_keyframes"] = nf
        S.sort_indices()
        diag = np.zeros((nf, 6, 6))
        for i in range(nf):
            lo, hi = S.indptr[i], S.indptr[i + 1]
            diag[i] = S.data[lo + np.searchsorted(S.indices[lo:hi], i)]
        Minv = np.linalg.inv(diag)
        t0 = time.time()
        x_direct = spla.spsolve(S.tocsc(), b) if nnzb < 0.2 * nf * nf else np.linalg.solve(S.toarray(), b)
        t_direct = time.time() - t0
        t0 = time.time()
        res = pcg(S, b, Minv, tols)
        t_pcg = time.time() - t0
        rec = {"lm_iteration": it, "lambda": lam, "chi2": chi2, "nonzero_blocks": int(nnzb), "fill": nnzb / float(nf * nf),
               "assemble_s": t_build, "direct_cpu_s": t_direct, "pcg_cpu_s": t_pcg, "step_inf_norm": float(np.abs(x_direct).max()),
               "pcg": {}}
        for tol in tols:
            n_it, x = res[tol]
            rec["pcg"]["%g" % tol] = {"iterations": n_it, "max_abs_error_of_pose_increment": float(np.abs(x - x_direct).max()),
                                      "relative_error": float(np.linalg.norm(x - x_direct) / np.linalg.norm(x_direct))}
        out["at"].append(rec)
        print(json.dumps(rec), flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--maps", default="GBA-1,GBA-1r,GBA-2r,GBA-2")
    ap.add_argument("--iterations", default="1,5")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    tols = [1e-2, 1e-4, 1e-6, 1e-8, 1e-10]
    res = [study(m, 1, [int(v) for v in a.iterations.split(",")], tols) for m in a.maps.split(",")]
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
