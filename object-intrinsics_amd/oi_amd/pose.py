"""Host-side pose prior (numpy, O(batch) work per step; SURVEY.md 2 #17: feeds the path its
(bs,4,4) box-to-world poses).  `Plane` restates src/utils/pose_sampler.py:66-90, 137-261 for the
scalar-range configuration data/example uses (cfg.yaml:1-9); list-valued ranges are not needed there."""
import numpy as np
from scipy.spatial.transform import Rotation as R


def _mat44(rot, trans=None):
    rot = np.asarray(rot)
    out = np.tile(np.eye(4), rot.shape[:-2] + (1, 1))
    out[..., :3, :3] = rot
    if trans is not None:
        out[..., :3, 3] = trans
    return out


def look_at_rot(eye, center=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0)):
    """src/utils/pose.py:13-61 (columns: right, up, forward)."""
    eye, center, up = (np.asarray(v, dtype=np.float64) for v in (eye, center, up))
    fwd = center - eye
    fwd = fwd / np.linalg.norm(fwd)
    up = up / np.linalg.norm(up)
    if np.allclose(fwd, [0, 1, 0]) and np.allclose(up, [0, 1, 0]):
        return np.array([[1.0, 0, 0], [0, 0, 1.0], [0, -1.0, 0]])
    right = np.cross(up, fwd)
    right /= np.linalg.norm(right)
    up2 = np.cross(fwd, right)
    up2 /= np.linalg.norm(up2)
    return np.stack([right, up2, fwd], -1)


class Plane:
    repr_dim = 6

    def __init__(self, cam_loc, rot_degree_range_scale, xy_range_scale, rot_roll_degree_range_scale):
        if isinstance(rot_degree_range_scale, (list, tuple)) or isinstance(rot_roll_degree_range_scale, (list, tuple)):
            raise NotImplementedError("list-valued rotation ranges are outside the data/example configuration")
        self.p2c_rot = look_at_rot(tuple(cam_loc)).T
        self.vec_phy = np.array([0.0, -1.0, 0.0])
        self.vec_cam = self.p2c_rot @ self.vec_phy
        self.rot_scale = rot_degree_range_scale
        self.roll_scale = rot_roll_degree_range_scale
        if isinstance(xy_range_scale, (int, float)):
            xy_range_scale = (xy_range_scale, xy_range_scale)
        self.xy = tuple(xy_range_scale)
        self.canonical_vec = np.asarray([0, -1, 0])

    def __call__(self, bs):
        rnd = np.random.uniform(size=(bs, 3))
        rot = (rnd[:, 0] - 0.5) * self.rot_scale / 180 * np.pi
        rot = self.p2c_rot @ R.from_rotvec(self.vec_phy[None, :] * rot[:, None]).as_matrix()
        x = (rnd[:, 1] * 2 - 1) * self.xy[0]
        y = (rnd[:, 2] * 2 - 1) * self.xy[1]
        z = -(self.vec_cam[0] * x + self.vec_cam[1] * y)
        z = np.zeros_like(z) if np.allclose(z, 0) else z / self.vec_cam[2]
        mat = _mat44(rot, np.stack([x, y, z], -1))
        roll = np.random.uniform(low=0, high=self.roll_scale / 180 * np.pi, size=bs)
        rot_roll = R.from_rotvec(np.asarray([0.0, 0, 1])[None] * roll[:, None]).as_matrix()
        return mat @ _mat44(rot_roll)

    @staticmethod
    def pose_to_vec_repr(pose):
        return pose[..., :2, :3].flatten(-2, -1)
