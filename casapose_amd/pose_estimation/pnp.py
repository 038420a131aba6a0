"""Host PnP: 2D-3D keypoint correspondences -> 6-DoF pose, without OpenCV.

The reference keeps this step on the host (`tf.numpy_function` around cv2, ransac_voting.py:13-57; bpnp_layers.py:86-117):
    cv2.solvePnPRansac(EPNP, confidence .9999, reprojectionError 12)  ->  cv2.solvePnP(ITERATIVE, useExtrinsicGuess)
and so does this build (north_star: "the final PnP solve stays on host").  cv2 is not part of this image, so the three
ingredients are restated here in NumPy fp64:
  * epnp()            -- Lepetit/Moreno-Noguer/Fua EPnP: 4 control points, null space of the 2n x 12 system, the
                         N = 1..3 linearisations of the control-point distances, Gauss-Newton on the betas;
  * solve_pnp_ransac  -- minimal sets of 5 (OpenCV's model size for EPnP), inliers by reprojection error, adaptive
                         iteration count from the confidence, final EPnP on the consensus set;
  * refine_lm         -- Levenberg-Marquardt on the reprojection error over ALL points in (rvec, t) (what
                         SOLVEPNP_ITERATIVE does from an extrinsic guess; 20 iterations like OpenCV's CvLevMarq).
Because the last stage minimises the same objective over the same points, the result is the local optimum OpenCV
converges to whenever both start in its basin; sampling order inside RANSAC is not reproducible across libraries
(parity with cv2 unpinned).

Also here: bpnp_backward(), the implicit-function-theorem gradient of that optimum with respect to the 2-D points
(BPNP_fast, bpnp_layers.py:138-212,278-359).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


# ------------------------------------------------------------------------------------------------
# rotations / projection
# ------------------------------------------------------------------------------------------------
def rodrigues(rvec: np.ndarray) -> np.ndarray:
    """axis-angle [3] -> rotation matrix [3,3] (cv2.Rodrigues; utils/geometry_utils.py:206-236 rodrigues_batch)."""
    r = np.asarray(rvec, np.float64).reshape(3)
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3) + _skew(r)
    k = r / th
    Kx = _skew(k)
    return np.eye(3) + np.sin(th) * Kx + (1.0 - np.cos(th)) * (Kx @ Kx)


def rodrigues_inverse(R: np.ndarray) -> np.ndarray:
    """rotation matrix -> axis-angle, stable near 0 and pi."""
    R = np.asarray(R, np.float64)
    c = np.clip((np.trace(R) - 1.0) * 0.5, -1.0, 1.0)
    th = np.arccos(c)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if th < 1e-8:
        return 0.5 * w
    if np.pi - th < 1e-6:  # near pi: take the axis from the symmetric part
        A = (R + np.eye(3)) * 0.5
        i = int(np.argmax(np.diag(A)))
        ax = A[:, i] / np.sqrt(max(A[i, i], 1e-300))
        if np.dot(ax, w) < 0:
            ax = -ax
        return ax * th
    return w * (th / (2.0 * np.sin(th)))


def _skew(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def project(points_3d: np.ndarray, K: np.ndarray, R: np.ndarray, t: np.ndarray) -> np.ndarray:
    cam = points_3d @ R.T + t.reshape(1, 3)
    uvw = cam @ K.T
    return uvw[:, :2] / uvw[:, 2:3]


# ------------------------------------------------------------------------------------------------
# EPnP
# ------------------------------------------------------------------------------------------------
def _control_points(X: np.ndarray) -> np.ndarray:
    c0 = X.mean(axis=0)
    Xc = X - c0
    w, V = np.linalg.eigh(Xc.T @ Xc / X.shape[0])
    cws = [c0]
    for i in range(3):
        cws.append(c0 + np.sqrt(max(w[i], 0.0)) * V[:, i])
    return np.array(cws)  # [4,3]


def _barycentric(X: np.ndarray, cws: np.ndarray) -> np.ndarray:
    A = (cws[1:] - cws[0]).T  # 3x3
    if abs(np.linalg.det(A)) < 1e-12:
        A = A + 1e-9 * np.eye(3)
    a = np.linalg.solve(A, (X - cws[0]).T).T  # [n,3]
    return np.concatenate([1.0 - a.sum(axis=1, keepdims=True), a], axis=1)  # [n,4]


def _pose_from_betas(betas, V, alphas, X):
    """control points in the camera frame = sum beta_k v_k; fix the sign, then absolute orientation (Horn/SVD)."""
    cc = np.zeros(12)
    for b, v in zip(betas, V):
        cc += b * v
    ccs = cc.reshape(4, 3)
    pc = alphas @ ccs
    if pc[:, 2].mean() < 0:
        ccs, pc = -ccs, -pc
    mu_c, mu_w = pc.mean(0), X.mean(0)
    H = (pc - mu_c).T @ (X - mu_w)
    U, _, Vt = np.linalg.svd(H)
    R = U @ Vt
    if np.linalg.det(R) < 0:
        U[:, 2] *= -1
        R = U @ Vt
    t = mu_c - R @ mu_w
    return R, t


def _reproj_error(X, x, K, R, t):
    cam = X @ R.T + t
    z = cam[:, 2:3]
    z = np.where(np.abs(z) < 1e-12, 1e-12, z)
    uv = (cam @ K.T)[:, :2] / z
    return np.sqrt(((uv - x) ** 2).sum(axis=1))


def epnp(points_3d: np.ndarray, points_2d: np.ndarray, K: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """EPnP.  Returns (R [3,3], t [3]) of the best of the N = 1, 2, 3 null-space solutions after Gauss-Newton on the betas.
    Exact for n >= 5 noise-free points in general position (the 2n x 12 system then has a 1-dimensional null space);
    with n = 4 the estimate is only a starting point for refine_lm."""
    X = np.asarray(points_3d, np.float64).reshape(-1, 3)
    x = np.asarray(points_2d, np.float64).reshape(-1, 2)
    K = np.asarray(K, np.float64)
    n = X.shape[0]
    fu, fv, uc, vc = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    cws = _control_points(X)
    al = _barycentric(X, cws)
    M = np.zeros((2 * n, 12))
    for j in range(4):
        M[0::2, 3 * j] = al[:, j] * fu
        M[0::2, 3 * j + 2] = al[:, j] * (uc - x[:, 0])
        M[1::2, 3 * j + 1] = al[:, j] * fv
        M[1::2, 3 * j + 2] = al[:, j] * (vc - x[:, 1])
    w, Vfull = np.linalg.eigh(M.T @ M)
    V = [Vfull[:, i] for i in range(4)]  # ascending eigenvalues: v0 spans the N=1 null space
    pairs = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
    rho = np.array([((cws[a] - cws[b]) ** 2).sum() for a, b in pairs])
    dv = np.zeros((4, 6, 3))
    for k in range(4):
        vk = V[k].reshape(4, 3)
        for p, (a, b) in enumerate(pairs):
            dv[k, p] = vk[a] - vk[b]

    def dist_residual_jac(betas):
        nb = len(betas)
        res, J = np.zeros(6), np.zeros((6, nb))
        for p in range(6):
            d = sum(betas[k] * dv[k, p] for k in range(nb))
            res[p] = d @ d - rho[p]
            for k in range(nb):
                J[p, k] = 2.0 * (dv[k, p] @ d)
        return res, J

    def gauss_newton(betas, iters=10):
        betas = np.array(betas, np.float64)
        for _ in range(iters):
            r, J = dist_residual_jac(betas)
            step, *_ = np.linalg.lstsq(J, -r, rcond=None)
            betas = betas + step
            if np.linalg.norm(step) < 1e-12:
                break
        return betas

    cands = []
    # N = 1: beta^2 * |dv0|^2 = rho
    num = sum(np.sqrt((dv[0, p] @ dv[0, p]) * rho[p]) for p in range(6))
    den = sum(dv[0, p] @ dv[0, p] for p in range(6))
    cands.append([num / max(den, 1e-300)])
    # N = 2: linear in (b00, b01, b11)
    L = np.array([[dv[0, p] @ dv[0, p], 2.0 * dv[0, p] @ dv[1, p], dv[1, p] @ dv[1, p]] for p in range(6)])
    b, *_ = np.linalg.lstsq(L, rho, rcond=None)
    if b[0] < 0:
        b = -b
    b0 = np.sqrt(max(b[0], 0.0))
    b1 = np.sqrt(max(abs(b[2]), 0.0)) * (1.0 if b[1] >= 0 else -1.0)
    cands.append([b0, b1])
    # N = 3: linear in (b00, b01, b02, b11, b12, b22)
    L3 = np.array([[dv[0, p] @ dv[0, p], 2 * dv[0, p] @ dv[1, p], 2 * dv[0, p] @ dv[2, p], dv[1, p] @ dv[1, p], 2 * dv[1, p] @ dv[2, p],
                    dv[2, p] @ dv[2, p]] for p in range(6)])
    b3, *_ = np.linalg.lstsq(L3, rho, rcond=None)
    if b3[0] < 0:
        b3 = -b3
    c0 = np.sqrt(max(b3[0], 0.0))
    c1 = np.sqrt(max(abs(b3[3]), 0.0)) * (1.0 if b3[1] >= 0 else -1.0)
    c2 = np.sqrt(max(abs(b3[5]), 0.0)) * (1.0 if b3[2] >= 0 else -1.0)
    cands.append([c0, c1, c2])

    best, best_err = None, np.inf
    for betas in cands:
        betas = gauss_newton(betas)
        R, t = _pose_from_betas(betas, V, al, X)
        err = _reproj_error(X, x, K, R, t).mean()
        if np.isfinite(err) and err < best_err:
            best, best_err = (R, t), err
    if best is None:
        raise np.linalg.LinAlgError("EPnP failed")
    return best


# ------------------------------------------------------------------------------------------------
# LM refinement (SOLVEPNP_ITERATIVE with an extrinsic guess)
# ------------------------------------------------------------------------------------------------
def _residual_and_jacobian(X, x, K, rvec, t):
    R = rodrigues(rvec)
    Xr = X @ R.T              # rotated points
    cam = Xr + t
    z = cam[:, 2]
    fu, fv = K[0, 0], K[1, 1]
    u = fu * cam[:, 0] / z + K[0, 2] + K[0, 1] * cam[:, 1] / z
    v = fv * cam[:, 1] / z + K[1, 2]
    r = np.stack([u - x[:, 0], v - x[:, 1]], axis=1).reshape(-1)
    n = X.shape[0]
    # d(u,v)/d cam
    du = np.stack([fu / z, K[0, 1] / z, -(fu * cam[:, 0] + K[0, 1] * cam[:, 1]) / z ** 2], axis=1)
    dv_ = np.stack([np.zeros(n), fv / z, -fv * cam[:, 1] / z ** 2], axis=1)
    # d cam / d rvec through a left perturbation: R(rvec + d) ~ exp(J_l d) R  =>  d cam = -[Xr]x J_l d
    th = np.linalg.norm(rvec)
    if th < 1e-8:
        Jl = np.eye(3) + 0.5 * _skew(rvec)
    else:
        k = rvec / th
        Kx = _skew(k)
        Jl = np.eye(3) + (1.0 - np.cos(th)) / th * Kx + (1.0 - np.sin(th) / th) * (Kx @ Kx)
    J = np.zeros((2 * n, 6))
    for i in range(n):
        dcam_dr = -_skew(Xr[i]) @ Jl
        J[2 * i, :3] = du[i] @ dcam_dr
        J[2 * i + 1, :3] = dv_[i] @ dcam_dr
        J[2 * i, 3:] = du[i]
        J[2 * i + 1, 3:] = dv_[i]
    return r, J


def refine_lm(points_3d, points_2d, K, rvec, t, iters: int = 20, eps: float = 1e-10):
    X = np.asarray(points_3d, np.float64).reshape(-1, 3)
    x = np.asarray(points_2d, np.float64).reshape(-1, 2)
    K = np.asarray(K, np.float64)
    p = np.concatenate([np.asarray(rvec, np.float64).reshape(3), np.asarray(t, np.float64).reshape(3)])
    lam = 1e-3
    r, J = _residual_and_jacobian(X, x, K, p[:3], p[3:])
    cost = r @ r
    for _ in range(iters):
        A = J.T @ J
        g = J.T @ r
        improved = False
        for _try in range(10):
            try:
                step = np.linalg.solve(A + lam * np.diag(np.maximum(np.diag(A), 1e-12)), -g)
            except np.linalg.LinAlgError:
                lam *= 10.0
                continue
            q = p + step
            r2, J2 = _residual_and_jacobian(X, x, K, q[:3], q[3:])
            c2 = r2 @ r2
            if np.isfinite(c2) and c2 < cost:
                p, r, J, lam = q, r2, J2, max(lam * 0.1, 1e-12)
                done = (cost - c2) < eps * max(cost, 1e-30)
                cost = c2
                improved = True
                break
            lam *= 10.0
        if not improved or done:
            break
    return p[:3], p[3:]


# ------------------------------------------------------------------------------------------------
# RANSAC + the reference's wrappers
# ------------------------------------------------------------------------------------------------
def solve_pnp_ransac(points_3d, points_2d, K, reprojection_error: float = 12.0, confidence: float = 0.9999, iterations: int = 100,
                     rng: Optional[np.random.Generator] = None):
    """-> (ok, rvec, t, inlier mask).  Minimal sets of 5 points solved with EPnP (4 when only 4 points are given)."""
    X = np.asarray(points_3d, np.float64).reshape(-1, 3)
    x = np.asarray(points_2d, np.float64).reshape(-1, 2)
    K = np.asarray(K, np.float64)
    n = X.shape[0]
    if n < 4:
        return False, np.zeros(3), np.zeros(3), np.zeros(n, bool)
    rng = rng if rng is not None else np.random.default_rng(0)
    m = 5 if n >= 5 else 4
    best_inl, best_cnt = None, -1
    max_it, it = iterations, 0
    while it < max_it:
        it += 1
        idx = rng.choice(n, m, replace=False)
        try:
            R, t = epnp(X[idx], x[idx], K)
        except np.linalg.LinAlgError:
            continue
        err = _reproj_error(X, x, K, R, t)
        inl = err <= reprojection_error
        cnt = int(inl.sum())
        if cnt > best_cnt:
            best_cnt, best_inl = cnt, inl
            w = cnt / n
            if w >= 1.0:
                break
            denom = np.log(max(1.0 - w ** m, 1e-300))
            need = np.log(max(1.0 - confidence, 1e-300)) / denom if denom < 0 else iterations  # no inlier yet: keep sampling
            max_it = int(min(iterations, max(np.ceil(need), 1)))
    if best_inl is None or best_cnt < m:
        return False, np.zeros(3), np.zeros(3), np.zeros(n, bool)
    try:
        R, t = epnp(X[best_inl], x[best_inl], K)
    except np.linalg.LinAlgError:
        return False, np.zeros(3), np.zeros(3), best_inl
    return True, rodrigues_inverse(R), t, best_inl


def pnp_rvec_t(points_3d, points_2d, camera_matrix, init_pose=None, rng=None) -> np.ndarray:
    """bpnp_layers.pnp (:86-117): RANSAC-EPnP (or the given initial pose) then iterative refinement over all points;
    returns the 6-vector (rvec, t) as float32."""
    X = np.asarray(points_3d, np.float64).reshape(-1, 3)
    x = np.asarray(points_2d, np.float64).reshape(-1, 2)
    if init_pose is None:
        ok, rvec, t, _ = solve_pnp_ransac(X, x, camera_matrix, 12.0, 0.9999, rng=rng)
        if not ok:
            try:
                R, t = epnp(X, x, camera_matrix)
                rvec = rodrigues_inverse(R)
            except np.linalg.LinAlgError:
                return np.full(6, np.nan, np.float32)
    else:
        rvec, t = np.asarray(init_pose[:3], np.float64), np.asarray(init_pose[3:6], np.float64)
    rvec, t = refine_lm(X, x, camera_matrix, rvec, t)
    return np.concatenate([rvec, t]).astype(np.float32)


def pnp(points_3d, points_2d, camera_matrix, rng=None) -> np.ndarray:
    """ransac_voting.pnp (:13-57): [3,4] float32 pose; zeros if the 2-D points are all zero (object absent, :17-18) or the
    solve fails (:48-49); the pose is negated when t_z < 0 (:53-55)."""
    x = np.asarray(points_2d, np.float64).reshape(-1, 2)
    X = np.asarray(points_3d, np.float64).reshape(-1, 3)
    assert X.shape[0] == x.shape[0], "points 3D and points 2D must have same number of vertices"
    if abs(x.sum()) < 1e-4:
        return np.zeros((3, 4), np.float32)
    p = pnp_rvec_t(X, x, camera_matrix, rng=rng)
    if not np.all(np.isfinite(p)):
        return np.zeros((3, 4), np.float32)
    R, t = rodrigues(p[:3]), p[3:].astype(np.float64)
    if t[2] < 0:
        t, R = -t, -R
    return np.concatenate([R, t.reshape(3, 1)], axis=1).astype(np.float32)


# ------------------------------------------------------------------------------------------------
# BPnP: gradient of the PnP optimum (implicit function theorem)
# ------------------------------------------------------------------------------------------------
def bpnp_backward(grad_pose: np.ndarray, points_2d: np.ndarray, points_3d: np.ndarray, K: np.ndarray, pose6: np.ndarray) -> np.ndarray:
    """d loss / d points_2d given d loss / d (rvec, t) at the optimum pose6 of  min_y sum |pi(y; z, K) - x|^2
    (bpnp_layers.py:138-212).  Stationarity g(x, y) = J_r(y)^T r(x, y) = 0 gives  dy/dx = -(dg/dy)^-1 dg/dx  with
    dg/dx = -J_r^T (r = pi - x) and dg/dy evaluated by central differences of g (it contains the second-order term
    sum r_i d2 pi_i / dy2 that the reference obtains from nested GradientTapes)."""
    X = np.asarray(points_3d, np.float64).reshape(-1, 3)
    x = np.asarray(points_2d, np.float64).reshape(-1, 2)
    K = np.asarray(K, np.float64)
    y = np.asarray(pose6, np.float64).reshape(6)

    def g(yv):
        r, J = _residual_and_jacobian(X, x, K, yv[:3], yv[3:])
        return J.T @ r

    _, J = _residual_and_jacobian(X, x, K, y[:3], y[3:])
    H = np.zeros((6, 6))
    for k in range(6):
        h = 1e-6 * max(1.0, abs(y[k]))
        e = np.zeros(6)
        e[k] = h
        H[:, k] = (g(y + e) - g(y - e)) / (2 * h)
    H = 0.5 * (H + H.T)
    dydx = np.linalg.solve(H, J.T)  # [6, 2n]:  -(dg/dy)^-1 (-J^T)
    return (np.asarray(grad_pose, np.float64).reshape(6) @ dydx).reshape(-1, 2)
