"""Registration metrics and pose helpers -- counterpart of the reference's
registration/train_utils.py (quat2mat :38-52, transform_point_cloud :55-60,
rt_to_transformation :71-74, rotation_error :77-80, translation_error :83-84,
rmse_loss :87-90, rotation_geodesic_error :93-105).  Device-agnostic: the
reference's `.cuda()` constants become tensors on the input's device."""
import math

import torch


def quat2mat(quat):
    """(B,4) unit quaternions (x, y, z, w) -> (B,3,3) rotation matrices."""
    x, y, z, w = quat[:, 0], quat[:, 1], quat[:, 2], quat[:, 3]
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    rows = [w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
            2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
            2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2]
    return torch.stack(rows, dim=1).reshape(quat.size(0), 3, 3)


def transform_point_cloud(point_cloud, rotation, translation):
    """point_cloud (B,3,N); rotation (B,3,3) or quaternion (B,4); translation (B,3)."""
    rot = quat2mat(rotation) if rotation.dim() == 2 else rotation
    return torch.matmul(rot, point_cloud) + translation.unsqueeze(2)


def rt_to_transformation(R, t):
    """R (B,3,3), t (B,3,1) -> homogeneous (B,4,4)."""
    bottom = R.new_tensor([0.0, 0.0, 0.0, 1.0]).expand(R.shape[0], 1, 4)
    return torch.cat([torch.cat([R, t], dim=2), bottom], dim=1)


def rotation_error(R, R_gt):
    """Isotropic rotation error in degrees."""
    cos_theta = (torch.einsum('bij,bij->b', R, R_gt) - 1) / 2
    return torch.acos(cos_theta.clamp(-1, 1)) * 180 / math.pi


def translation_error(t, t_gt):
    return torch.norm(t - t_gt, dim=1)


def rmse_loss(pts, T, T_gt):
    """Mean point distance between pts (B,N,3) moved by T and by T_gt."""
    moved = pts @ T[:, :3, :3].transpose(1, 2) + T[:, :3, 3].unsqueeze(1)
    moved_gt = pts @ T_gt[:, :3, :3].transpose(1, 2) + T_gt[:, :3, 3].unsqueeze(1)
    return torch.norm(moved - moved_gt, dim=2).mean(dim=1)


def rotation_geodesic_error(m1, m2):
    """Geodesic angle (radians) between batches of rotations."""
    m = torch.bmm(m1, m2.transpose(1, 2))
    cos = (m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2] - 1) / 2
    return torch.acos(cos.clamp(-1, 1))
