"""Synthetic input for the per-feature epipolar update: a textured slanted plane seen by a
moving camera.  Every view of a plane is a homography of the texture, so each frame is rendered exactly (up to the
8-bit quantisation) and the true inverse depth of every feature is known in closed form.

Conventions: camera C maps points of the first camera A by X_C = R_C X_A + t_C; the plane is n . X_A = d.
T_ref_to_new (flame.cc:1315, fnew.pose.inverse() * pf.pose) is then (R_n R_r^T, t_n - R_n R_r^T t_r).
"""
from __future__ import annotations

import numpy as np

from .synth import uniform01


def intrinsics(width: int, height: int):
    f = 525.0 * width / 640.0
    K = np.array([[f, 0, width / 2.0], [0, f, height / 2.0], [0, 0, 1]], np.float64)
    return K.astype(np.float32), np.linalg.inv(K).astype(np.float32)


def texture(width: int, height: int, seed: int, sigma: float = 1.6, margin: int = 64) -> np.ndarray:
    """Band-limited random texture (float64, 0..255) on a canvas `margin` px larger than the image on each side."""
    W, H = width + 2 * margin, height + 2 * margin
    noise = uniform01(seed, W * H, stream=11).astype(np.float64).reshape(H, W) - 0.5
    fy = np.fft.fftfreq(H)[:, None]
    fx = np.fft.rfftfreq(W)[None, :]
    g = np.exp(-2.0 * (np.pi * sigma) ** 2 * (fx * fx + fy * fy))
    img = np.fft.irfft2(np.fft.rfft2(noise) * g, s=(H, W))
    img = img / img.std() * 55.0 + 128.0
    return np.clip(img, 0.0, 255.0)


def rot(axis, angle):
    axis = np.asarray(axis, np.float64)
    axis = axis / np.linalg.norm(axis)
    a = angle
    Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(a) * Kx + (1 - np.cos(a)) * (Kx @ Kx)


def quat_from_rot(R):
    """(w, x, y, z), w >= 0."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    if q[0] < 0:
        q = -q
    return q / np.linalg.norm(q)


def _bilinear(img, x, y):
    x0 = np.floor(x).astype(np.int64)
    y0 = np.floor(y).astype(np.int64)
    x0 = np.clip(x0, 0, img.shape[1] - 2)
    y0 = np.clip(y0, 0, img.shape[0] - 2)
    dx, dy = x - x0, y - y0
    return ((1 - dx) * (1 - dy) * img[y0, x0] + dx * (1 - dy) * img[y0, x0 + 1] + (1 - dx) * dy * img[y0 + 1, x0]
            + dx * dy * img[y0 + 1, x0 + 1])


class PlaneScene:
    def __init__(self, width=640, height=480, seed=7, normal=(0.25, -0.1, 1.0), distance=2.0, margin=64):
        self.width, self.height, self.margin = width, height, margin
        self.K32, self.Kinv32 = intrinsics(width, height)
        self.K = self.K32.astype(np.float64)
        self.Kinv = np.linalg.inv(self.K)
        n = np.asarray(normal, np.float64)
        self.n = n / np.linalg.norm(n)
        self.d = float(distance)
        self.tex = texture(width, height, seed, margin=margin)
        self.cams = {}

    def add_camera(self, cam_id: int, R, t):
        self.cams[cam_id] = (np.asarray(R, np.float64), np.asarray(t, np.float64))

    def render(self, cam_id: int) -> np.ndarray:
        R, t = self.cams[cam_id]
        H = self.K @ (R + np.outer(t, self.n) / self.d) @ self.Kinv  # A pixels -> C pixels
        Hi = np.linalg.inv(H)
        ys, xs = np.mgrid[0:self.height, 0:self.width].astype(np.float64)
        w = Hi[2, 0] * xs + Hi[2, 1] * ys + Hi[2, 2]
        ax = (Hi[0, 0] * xs + Hi[0, 1] * ys + Hi[0, 2]) / w + self.margin
        ay = (Hi[1, 0] * xs + Hi[1, 1] * ys + Hi[1, 2]) / w + self.margin
        img = _bilinear(self.tex, ax, ay)
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)

    def true_idepth(self, cam_id: int, xy: np.ndarray) -> np.ndarray:
        R, t = self.cams[cam_id]
        nc = R @ self.n
        dc = self.d + nc @ t
        rays = (self.Kinv @ np.concatenate([xy.astype(np.float64), np.ones((xy.shape[0], 1))], axis=1).T).T
        return (rays @ nc) / dc

    def relative(self, ref_id: int, cmp_id: int):
        """(q, t) float32 of T_ref_to_cmp."""
        Rr, tr = self.cams[ref_id]
        Rc, tc = self.cams[cmp_id]
        R = Rc @ Rr.T
        t = tc - R @ tr
        return quat_from_rot(R).astype(np.float32), t.astype(np.float32)


def make_features(scene: PlaneScene, dtype, anchors, n_per_anchor: int, seed: int, mu_noise=0.08, var=0.02, border=12):
    """Features on a jittered grid in each anchor frame with a noisy prior around the true inverse depth."""
    feats = []
    fid = 0
    for a_i, anchor in enumerate(anchors):
        nx = int(np.sqrt(n_per_anchor * scene.width / scene.height))
        ny = max(1, n_per_anchor // nx)
        n = nx * ny
        u = uniform01(seed, n, stream=20 + a_i)
        v = uniform01(seed, n, stream=30 + a_i)
        e = uniform01(seed, n, stream=40 + a_i)
        cx = (np.tile(np.arange(nx), ny) + u) * ((scene.width - 2 * border) / nx) + border
        cy = (np.repeat(np.arange(ny), nx) + v) * ((scene.height - 2 * border) / ny) + border
        xy = np.stack([cx, cy], axis=1).astype(np.float32)
        truth = scene.true_idepth(anchor, xy)
        arr = np.zeros(n, dtype=dtype)
        arr["id"] = np.arange(fid, fid + n)
        arr["frame_id"] = anchor
        arr["x"], arr["y"] = xy[:, 0], xy[:, 1]
        arr["idepth_mu"] = (truth * (1.0 + mu_noise * (2.0 * e - 1.0))).astype(np.float32)
        arr["idepth_var"] = np.float32(var)
        arr["valid"] = 1
        fid += n
        feats.append(arr)
    return np.concatenate(feats)


def standard_scene(width=640, height=480, seed=7):
    """Pose-frames 10 and 11 (11 = newest pose-frame), new frame 12: a sideways-and-forward dolly with a small yaw."""
    s = PlaneScene(width, height, seed)
    s.add_camera(10, np.eye(3), [0, 0, 0])
    s.add_camera(11, rot([0, 1, 0], 0.01), [-0.06, 0.005, -0.02])
    s.add_camera(12, rot([0.2, 1, 0.1], 0.02), [-0.12, 0.01, -0.05])
    return s


def poses_for(scene: PlaneScene, anchors, new_id: int, curr_pf_id: int):
    out = []
    for a in anchors:
        qn, tn = scene.relative(a, new_id)
        qp, tp = scene.relative(a, curr_pf_id)
        out.append(dict(id=a, q_to_new=qn, t_to_new=tn, q_to_pf=qp, t_to_pf=tp))
    return out
