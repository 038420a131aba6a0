"""Host PnP (EPnP + RANSAC + LM), BPnP gradient and the ADD / ADD-S / 2-D evaluation -- known-answer and property
tests (the reference delegates these to OpenCV, which is not available here: parity with cv2 unpinned)."""
import numpy as np
import pytest

from casapose_amd.pose_estimation import pnp as P
from casapose_amd.pose_estimation import pose_evaluation as E

K = np.array([[572.4, 0, 325.3], [0, 573.6, 242.0], [0, 0, 1]])


def _scene(rng, n=9, noise=0.0):
    rv = rng.normal(0, 1, 3)
    rv *= rng.uniform(0.1, 3.0) / np.linalg.norm(rv)
    R = P.rodrigues(rv)
    t = np.array([rng.uniform(-100, 100), rng.uniform(-100, 100), rng.uniform(600, 1200)])
    X = rng.uniform(-60, 60, (n, 3))
    x = P.project(X, K, R, t) + rng.normal(0, noise, (n, 2)) if noise else P.project(X, K, R, t)
    return X, x, R, t, rv


def test_rodrigues_round_trip_and_known_values():
    assert np.allclose(P.rodrigues(np.zeros(3)), np.eye(3))
    Rz = P.rodrigues(np.array([0, 0, np.pi / 2]))
    assert np.allclose(Rz, [[0, -1, 0], [1, 0, 0], [0, 0, 1]], atol=1e-12)
    rng = np.random.default_rng(0)
    for _ in range(50):
        rv = rng.normal(0, 1, 3)
        rv *= rng.uniform(0, np.pi - 1e-3) / np.linalg.norm(rv)
        R = P.rodrigues(rv)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(R) - 1) < 1e-12
        assert np.allclose(P.rodrigues_inverse(R), rv, atol=1e-8)
    Rpi = P.rodrigues(np.array([np.pi, 0, 0]))
    assert np.allclose(P.rodrigues(P.rodrigues_inverse(Rpi)), Rpi, atol=1e-6)


def test_epnp_exact_on_noise_free_points():
    rng = np.random.default_rng(1)
    for n in (5, 6, 9, 30):  # (four points leave a 4-dimensional null space: EPnP proper needs n >= 5, like OpenCV's RANSAC model size)
        X, x, R, t, _ = _scene(rng, n)
        Re, te = P.epnp(X, x, K)
        assert np.abs(Re - R).max() < 1e-6 and np.abs(te - t).max() < 1e-4
        assert np.abs(P.project(X, K, Re, te) - x).max() < 1e-5


def test_pnp_wrapper_conventions():
    rng = np.random.default_rng(2)
    X, x, R, t, _ = _scene(rng)
    pose = P.pnp(X, x, K)
    assert pose.dtype == np.float32 and pose.shape == (3, 4)
    assert np.abs(pose[:, :3] - R).max() < 1e-5 and np.abs(pose[:, 3] - t).max() < 1e-2
    assert np.all(P.pnp(X, np.zeros_like(x), K) == 0)          # absent object -> zero pose
    with pytest.raises(AssertionError):
        P.pnp(X, x[:5], K)
    # noisy points: the refined pose is a stationary point of the reprojection error over ALL points
    xn = x + rng.normal(0, 1.0, x.shape)
    p6 = P.pnp_rvec_t(X, xn, K).astype(np.float64)
    rv, tt = P.refine_lm(X, xn, K, p6[:3], p6[3:], iters=50, eps=1e-16)
    r, J = P._residual_and_jacobian(X, xn, K, rv, tt)
    assert np.abs(J.T @ r).max() < 1e-3 * np.abs(J).max() * np.abs(r).max()


def test_ransac_rejects_outliers():
    rng = np.random.default_rng(3)
    X, x, R, t, _ = _scene(rng)
    xo = x.copy()
    xo[3] += 80
    xo[7] -= 60
    ok, rv, tt, inl = P.solve_pnp_ransac(X, xo, K, rng=np.random.default_rng(0))
    assert ok and list(np.where(~inl)[0]) == [3, 7]
    assert np.abs(P.rodrigues(rv) - R).max() < 1e-6 and np.abs(tt - t).max() < 1e-3


def test_analytic_jacobian_and_bpnp_gradient():
    rng = np.random.default_rng(4)
    X, x, R, t, rv = _scene(rng)
    r, J = P._residual_and_jacobian(X, x + 1.0, K, rv, t)
    y = np.concatenate([rv, t])
    for k in range(6):
        e = np.zeros(6)
        e[k] = 1e-6
        rp, _ = P._residual_and_jacobian(X, x + 1.0, K, (y + e)[:3], (y + e)[3:])
        rm, _ = P._residual_and_jacobian(X, x + 1.0, K, (y - e)[:3], (y - e)[3:])
        assert np.allclose(J[:, k], (rp - rm) / 2e-6, rtol=1e-5, atol=1e-6 * np.abs(J).max())
    xn = x + rng.normal(0, 1.0, x.shape)
    p6 = P.pnp_rvec_t(X, xn, K).astype(np.float64)
    p6 = np.concatenate(P.refine_lm(X, xn, K, p6[:3], p6[3:], iters=100, eps=1e-16))
    gp = rng.normal(0, 1, 6)
    gx = P.bpnp_backward(gp, xn, X, K, p6)
    num = np.zeros_like(xn)
    for i in range(9):
        for j in range(2):
            d = np.zeros_like(xn)
            d[i, j] = 1e-4
            a = np.concatenate(P.refine_lm(X, xn + d, K, p6[:3], p6[3:], iters=100, eps=1e-16))
            b = np.concatenate(P.refine_lm(X, xn - d, K, p6[:3], p6[3:], iters=100, eps=1e-16))
            num[i, j] = gp @ (a - b) / 2e-4
    assert np.abs(gx - num).max() < 2e-3 * np.abs(num).max()


def test_transform_points_back_matches_affine():
    from casapose_amd.train_engine import crop_to_image_affine

    rng = np.random.default_rng(5)
    off = np.array([12.0, 40.0, 0, 0, 3.0, -2.0, 17.0, 1.1, 640, 480])
    pts = rng.uniform(0, 448, (9, 2))
    A = crop_to_image_affine(off[None]).reshape(2, 3).astype(np.float64)
    exp = pts @ A[:, :2].T + A[:, 2]
    assert np.allclose(E.transform_points_back(pts, off), exp, atol=1e-3)


def _eval_scene(rng, b=2, oc=3):
    kp3 = rng.uniform(-50, 50, (b, oc, 1, 9, 3))
    kp3[:] = kp3[0:1]
    poses = np.zeros((b, oc, 1, 3, 4))
    for n in range(b):
        for o in range(oc):
            rv = rng.normal(0, 0.5, 3)
            poses[n, o, 0, :, :3] = P.rodrigues(rv)
            poses[n, o, 0, :, 3] = [rng.uniform(-50, 50), rng.uniform(-50, 50), rng.uniform(700, 900)]
    cams = np.tile(K, (b, 1, 1))
    offsets = np.tile(np.array([[0.0, 0, 0, 0, 0, 0, 0, 1, 640, 480]]), (b, 1))
    pts = np.zeros((b, oc, 9, 2))
    for n in range(b):
        for o in range(oc):
            pts[n, o] = P.project(kp3[n, o, 0], K, poses[n, o, 0, :, :3], poses[n, o, 0, :, 3])
    return kp3, poses, cams, offsets, pts


def test_estimate_and_evaluate_poses_bookkeeping():
    rng = np.random.default_rng(6)
    b, oc = 2, 3
    kp3, poses, cams, offsets, pts = _eval_scene(rng, b, oc)
    valid = np.array([[1, 1, 0], [1, 1, 1]])
    pts_in = pts.copy()
    pts_in[0, 1] = 0            # object 1 of image 0 was not voted at all -> missing
    est, false_pos = E.estimate_poses(pts_in, kp3, cams, valid, offsets)
    assert est.shape == (b, oc, 3, 4) and np.all(est[0, 1] == 0)
    assert list(false_pos) == [0, 0, 1]                      # object 2 of image 0 voted although absent from the GT
    assert np.abs(est[1, 2] - poses[1, 2, 0]).max() < 1e-2
    diam = np.full((b, oc, 1), 100.0)
    cnt = np.full((b, oc, 1), 9)
    e2, e3, v2, v3, miss, vcount, fp = E.evaluate_poses(est, poses, pts_in, kp3, cnt, cams, diam, valid, 5.0)
    assert list(vcount) == [2, 2, 1] and list(miss) == [0, 1, 0] and list(fp) == [0, 0, 1]
    assert list(v3) == [2, 1, 1] and list(v2) == [2, 1, 1]
    assert abs(e2[1] - 99.9) < 0.1 and abs(e3[1] - 999.9) < 0.1   # the missing object contributes the sentinel errors
    assert e3[0] < 0.1 and e2[0] < 0.1


def test_add_versus_adds_selection():
    """ADD-S (nearest neighbour) is selected purely by the vertex count 7862 / 3417 (ransac_voting.py:619)."""
    rng = np.random.default_rng(7)
    V = 3417
    mesh = rng.uniform(-40, 40, (V, 3))
    mesh = np.concatenate([mesh, -mesh])[:V]  # roughly symmetric under the point reflection used below
    pose_gt = np.concatenate([np.eye(3), np.array([[0.0], [0.0], [800.0]])], axis=1)
    Rflip = P.rodrigues(np.array([0, 0, np.pi]))
    pose_est = np.concatenate([Rflip, np.array([[0.0], [0.0], [800.0]])], axis=1)
    args = dict(camera_matrixes=K[None], diameters=np.array([[[120.0]]]), valid_points_filter=np.array([[1]]), allowed_error_2d=5.0)
    pts = mesh[None, None, None]
    add = E.evaluate_poses(pose_est[None, None], pose_gt[None, None, None], None, pts[:, :, :, :V - 1], np.array([[[V - 1]]]), **args)
    adds = E.evaluate_poses(pose_est[None, None], pose_gt[None, None, None], None, pts, np.array([[[V]]]), **args)
    cam_gt = mesh @ pose_gt[:, :3].T + pose_gt[:, 3]
    cam_est = mesh @ pose_est[:, :3].T + pose_est[:, 3]
    assert abs(add[1][0] - np.linalg.norm(cam_gt[:V - 1] - cam_est[:V - 1], axis=1).mean()) < 1e-2
    brute = np.sqrt(np.array([((cam_est - a) ** 2).sum(1).min() for a in cam_gt[:200]]) + 1e-5).mean()
    d_all = E._adds_error(cam_gt[:200], cam_est).mean()
    assert abs(d_all - brute) < 1e-6
    assert adds[1][0] < add[1][0]


def test_poses_pnp_zeroes_small_objects():
    import torch

    rng = np.random.default_rng(8)
    b, oc = 1, 2
    kp3, poses, cams, offsets, pts = _eval_scene(rng, b, oc)
    seg = torch.zeros(b, 32, 32, oc + 1)
    seg[..., 0] = 1.0
    seg[0, 4:20, 4:20, 1] = 2.0      # object 1: 256 px; object 2: none
    out = E.poses_pnp(pts[..., ::-1].copy(), seg, kp3, cams, oc, min_num=20)
    assert out.shape == (b, oc, 1, 3, 4)
    assert np.abs(out[0, 0, 0] - poses[0, 0, 0]).max() < 1e-2 and np.all(out[0, 1] == 0)


def test_bpnp_reprojection_loss_gradient_by_finite_differences():
    """use_bpnp_reprojection_loss (loss_functions.py:264-323): the host loss's analytic gradient (explicit term + IFT through
    the PnP optimum) equals central differences of the loss with the PnP re-solved at every perturbed point set."""
    from casapose_amd.training import bpnp_reprojection_loss_host
    from casapose_amd.train_engine import crop_to_image_affine

    rng = np.random.default_rng(9)
    B, oc, kp = 1, 2, 9
    kp3, poses, cams, offsets, pts = _eval_scene(rng, B, oc)
    off = np.array([[12.0, 40.0, 0, 0, 3.0, -2.0, 9.0, 1.1, 640, 480]])
    A = crop_to_image_affine(off).astype(np.float64).reshape(B, 2, 3)
    Li = np.linalg.inv(A[0, :, :2])
    crop_xy = (pts[0] + rng.normal(0, 1.5, pts[0].shape) - A[0, :, 2]) @ Li.T      # noisy votes, in crop pixels
    coords = crop_xy[None, ..., ::-1].copy()                                          # (y,x)
    gt = pts + rng.normal(0, 2.0, pts.shape)
    avail = np.ones((B, oc))

    def f(c):
        return bpnp_reprojection_loss_host(c, gt, A.reshape(B, 6), avail, kp3[:, :, 0], K, 12.5, 1.0, rng=np.random.default_rng(0))

    loss, g, est = f(coords)
    assert np.isfinite(loss) and est.shape == (B, oc, 1, 3, 4)
    num = np.zeros_like(coords)
    for o in range(oc):
        for j in range(0, kp, 2):
            for a in range(2):
                d = np.zeros_like(coords)
                d[0, o, j, a] = 1e-3
                num[0, o, j, a] = (f(coords + d)[0] - f(coords - d)[0]) / 2e-3
    m = num != 0
    assert np.abs(g[m] - num[m]).max() < 2e-2 * np.abs(num[m]).max()


# --------------------------------------------------------------------------------------------------
# host geometry against the REFERENCE'S OWN NumPy functions (tests/golden/make_geometry_golden.py executes them; round 4)
# --------------------------------------------------------------------------------------------------
def test_host_geometry_equals_the_reference_functions():
    """crop -> image transform (ransac_voting.py:71-89 through map_offsets' argument order), its inverse (geometry_utils.py:7-34), projection
    (:161-170), quaternion -> pose (geometry_utils.py:144-181, both quaternion orders), fixed-transform application (:48-57) and the 2-D rotation
    matrix (:37-45): golden outputs of the reference's code on random inputs, including rotation / translation jitter and scales != 1."""
    import json
    import os

    from casapose_amd.data_handler import vectorfield_dataset as VD
    from casapose_amd.pose_estimation import pose_evaluation as PE
    from casapose_amd.train_engine import crop_to_image_affine, project_keypoints

    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "geometry_ref.json")))
    for c in ref["transform_points_back"]:
        pts, off = np.asarray(c["points_xy"]), np.asarray(c["offsets"])
        got = PE.transform_points_back(pts, off)
        assert np.allclose(got, c["image_xy"], atol=2e-3), np.abs(got - c["image_xy"]).max()       # the reference computes in float32
        A = crop_to_image_affine(off[None])[0].reshape(2, 3).astype(np.float64)                     # the same map as one 2x3 matrix (training path)
        assert np.allclose(pts @ A[:, :2].T + A[:, 2], c["image_xy"], atol=5e-3)
    for c, back in zip(ref["apply_offsets_round_trip"], ref["transform_points_back"]):
        assert np.allclose(c["crop_xy"], back["points_xy"], atol=5e-3)      # consistency of the golden itself: image -> crop inverts crop -> image
    for c in ref["project"]:
        xy, cam = PE.project(np.asarray(c["xyz"]), np.asarray(c["K"]), np.asarray(c["RT"]))
        assert np.allclose(xy, c["xy"], atol=1e-3) and np.allclose(cam, c["xyz_cam"], atol=1e-3)
        assert np.allclose(project_keypoints(np.asarray(c["xyz"]), np.asarray(c["K"]), np.asarray(c["RT"])), c["xy"], atol=1e-3)
    for c in ref["quaternion_matrix"]:
        assert np.allclose(VD.quaternion_matrix(c["q_xyzw"], c["t"]), c["RT"], atol=1e-12)
        assert np.allclose(VD.quaternion_matrix(c["q_xyzw"], c["t"], wxyz_input=True), c["RT_wxyz_input"], atol=1e-12)
        assert np.allclose(VD.quaternion_matrix(c["q_xyzw"]), c["R"], atol=1e-12)
    t = ref["transform_points"]
    assert np.allclose(VD.transform_points(np.asarray(t["points"]), np.asarray(t["M"])), t["out"], atol=1e-12)
    for c in ref["get_rotation_matrix_2D"]:
        assert np.allclose(VD.get_rotation_matrix_2D(c["center"], c["angle"]), c["M"], atol=1e-4)
