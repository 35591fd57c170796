"""Fixed pinhole scene camera (host-side inputs of the path).  Mirrors
src/models/camera_network.py:9-28 and get_identity_pose (src/utils/pose.py:190-206): same buffer
names (`intrinsics`, `intrinsics_inv`, `c2w`, `w2c`) so checkpoints interchange."""
import numpy as np
import torch
import torch.nn as nn


class Camera(nn.Module):
    def __init__(self, cam_dist, fov, resolution):
        super().__init__()
        self.resolution = resolution
        self.cam_dist = cam_dist
        focal = (resolution / 2) * 1 / np.tan(0.5 * fov * np.pi / 180.0)
        K = torch.tensor([[focal, 0, 0.5 * resolution, 0], [0, focal, 0.5 * resolution, 0],
                          [0, 0, 1, 0], [0, 0, 0, 1]], dtype=torch.float32)
        self.register_buffer("intrinsics", K)
        self.register_buffer("intrinsics_inv", torch.tensor(np.linalg.inv(K.numpy()), dtype=torch.float32))
        # look_at((0,0,-1)) is the identity rotation; the camera sits at -cam_dist on z
        c2w = torch.eye(4, dtype=torch.float32)
        c2w[2, 3] = -float(cam_dist)
        w2c = torch.eye(4, dtype=torch.float32)
        w2c[2, 3] = float(cam_dist)
        self.register_buffer("c2w", c2w)
        self.register_buffer("w2c", w2c)
