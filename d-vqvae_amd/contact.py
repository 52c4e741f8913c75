"""Contact / penetration proxies on the device: mirrors of the reference's
``utils/utils_loss.py`` (``get_NN`` :7-24, ``get_interior`` :27-45) and of the penetration / contact terms of
``utils/loss.py`` ``TTT_loss`` (:144-160), batched over grasps.  The reference runs them through pytorch3d
(``knn_points``, ``Meshes.verts_normals_packed``), which is not available here: parity is pinned against
``oracle/contact_oracle.py`` (a numpy restatement of the published algorithms), not against pytorch3d output.
"""
import numpy as np
import torch

from . import ops


def face_csr(faces, n_verts):
    """faces [F,3] (array-like) -> (faces int32 [F,3], vf_off int32 [V+1], vf_face int32 [3F]) numpy arrays: for each
    vertex the faces that contain it, ascending."""
    f = np.asarray(faces, dtype=np.int64).reshape(-1, 3)
    vert = f.reshape(-1)
    face = np.repeat(np.arange(f.shape[0]), 3)
    order = np.lexsort((face, vert))
    counts = np.bincount(vert, minlength=n_verts)
    off = np.zeros(n_verts + 1, dtype=np.int64)
    off[1:] = np.cumsum(counts)
    return f.astype(np.int32), off.astype(np.int32), face[order].astype(np.int32)


class HandTopology:
    """Device copies of a mesh topology (MANO: 778 vertices, 1538 faces) for ``vertex_normals``."""

    def __init__(self, faces, n_verts, device):
        f, off, vf = face_csr(faces, n_verts)
        self.n_verts = n_verts
        self.faces = torch.from_numpy(f).to(device)
        self.vf_off = torch.from_numpy(off).to(device)
        self.vf_face = torch.from_numpy(vf).to(device)

    def normals(self, verts):
        return ops.vertex_normals(verts.contiguous(), self.faces, self.vf_off, self.vf_face)


def get_NN(src_xyz, trg_xyz):
    """(nn_dists [B,N1] squared, nn_idx [B,N1]) -- utils_loss.get_NN with k=1."""
    return ops.nn_points(src_xyz, trg_xyz)


def get_interior(src_face_normal, src_xyz, trg_xyz, trg_NN_idx):
    """utils_loss.get_interior(hand normals, hand verts, object points, NN index of each object point in the hand)."""
    return ops.interior(src_face_normal.contiguous(), src_xyz.contiguous(), trg_xyz, trg_NN_idx.contiguous())


def grasp_proxies(topology, hand_xyz, obj_xyz, contact_threshold=0.02 ** 2):
    """Per-grasp proxies from TTT_loss (loss.py:154-164), unreduced so that callers can rank grasps:
    ``penetration`` [B] = sum of squared NN distances over interior object points (the reference reports
    120 * sum / B), ``n_interior`` [B], ``n_contact`` [B] = object points within 2 cm of the hand (the dynamic contact
    region ``nn_dist < 0.02**2``), plus the raw ``nn_dist``, ``nn_idx``, ``interior``.
    hand_xyz [B,778,3]; obj_xyz [B,N,3] (any strides, e.g. ``cloud[:, :3].transpose(1, 2)``)."""
    normals = topology.normals(hand_xyz)
    nn_dist, nn_idx = get_NN(obj_xyz, hand_xyz.contiguous())
    inside = get_interior(normals, hand_xyz, obj_xyz, nn_idx)
    zero = torch.zeros((), dtype=nn_dist.dtype, device=nn_dist.device)
    return {"penetration": torch.where(inside, nn_dist, zero).sum(dim=1), "n_interior": inside.sum(dim=1),
            "n_contact": (nn_dist < contact_threshold).sum(dim=1), "nn_dist": nn_dist, "nn_idx": nn_idx,
            "interior": inside, "normals": normals}
