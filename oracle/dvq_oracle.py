"""CPU oracle for the D-VQVAE batched grasp-generation path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch-CPU (fp32) restatement of the reference algorithm,
written functionally over a ``state_dict`` so that it shares no module code with
either the reference or the product.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it; the product package
(``d-vqvae_amd``) never does.

Parity status: PINNED for every stage except MANO.  ``tools/make_golden.py``
imports the real reference from ``/root/reference`` in the build container, runs
it on seeded inputs/weights and commits the results under ``tests/golden``;
``tests/test_oracle_golden.py`` checks every function here against those
vectors.  The MANO layer (third-party ``mano`` pip package, no version pinned by
the reference, not installed, not vendored) is restated in ``mano_oracle.py``
from the published smplx-style LBS algorithm: that sub-step is "parity
unpinned".

Batched semantics.  The reference only works for B=1 (``get_emb`` does
``.view(1, dim)``, network/vqvae/quantizer.py:68-75; ``idx6.repeat(1,3,3)``
collapses the label to sample 0, network/gen_net.py:88-89).  The contract of
this oracle (and of the product) is "B independent B=1 reference calls":
per-sample label, per-sample codebook row gather.

Each function cites the reference lines it follows.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

BN_EPS = 1e-5  # torch.nn.BatchNorm1d default; layers declared pointnet_encoder.py:21-25,132-134


# --------------------------------------------------------------------------- PointNet
def _conv_bn(sd: SD, pre: str, conv: str, bn: str, x: Tensor, relu: bool) -> Tensor:
    """1x1 Conv1d followed by eval-mode BatchNorm1d (pointnet_encoder.py:29-31,150,161-162)."""
    y = F.conv1d(x, sd[f"{pre}{conv}.weight"], sd[f"{pre}{conv}.bias"])
    y = F.batch_norm(y, sd[f"{pre}{bn}.running_mean"], sd[f"{pre}{bn}.running_var"],
                     sd[f"{pre}{bn}.weight"], sd[f"{pre}{bn}.bias"], False, 0.0, BN_EPS)
    return F.relu(y) if relu else y


def _fc_bn(sd: SD, pre: str, fc: str, bn: Optional[str], x: Tensor, relu: bool) -> Tensor:
    y = F.linear(x, sd[f"{pre}{fc}.weight"], sd[f"{pre}{fc}.bias"])
    if bn is not None:
        y = F.batch_norm(y, sd[f"{pre}{bn}.running_mean"], sd[f"{pre}{bn}.running_var"],
                         sd[f"{pre}{bn}.weight"], sd[f"{pre}{bn}.bias"], False, 0.0, BN_EPS)
    return F.relu(y) if relu else y


def stn3d(sd: SD, pre: str, x: Tensor) -> Tensor:
    """STN3d.forward, network/pointnet_encoder.py:27-45.  x [B,C,N] -> trans [B,3,3]."""
    B = x.shape[0]
    y = _conv_bn(sd, pre, "conv1", "bn1", x, True)
    y = _conv_bn(sd, pre, "conv2", "bn2", y, True)
    y = _conv_bn(sd, pre, "conv3", "bn3", y, True)          # STN's third layer HAS a ReLU (:31)
    y = y.max(dim=2)[0]                                      # :32-33
    y = _fc_bn(sd, pre, "fc1", "bn4", y, True)
    y = _fc_bn(sd, pre, "fc2", "bn5", y, True)
    y = _fc_bn(sd, pre, "fc3", None, y, False)
    iden = torch.tensor([1, 0, 0, 0, 1, 0, 0, 0, 1], dtype=torch.float32).view(1, 9)
    return (y + iden).view(B, 3, 3)                          # :39-44


def pointnet_encode(sd: SD, pre: str, x: Tensor) -> Tuple[Tensor, Tensor]:
    """PointNetEncoder.forward (global_feat=True, feature_transform=False),
    network/pointnet_encoder.py:140-169.  x [B,C,N] (C = 3 or 4) -> (feat [B,1024], trans [B,3,3]).
    The extra channel (C=4) bypasses the 3x3 transform (utils/utils.py:163-181 size_splits)."""
    B, C, N = x.shape
    trans = stn3d(sd, pre + "stn.", x)
    pts = x.transpose(2, 1)                                  # [B,N,C]
    xyz = torch.bmm(pts[:, :, :3], trans)                    # row vector @ trans (:146)
    if C > 3:
        pts = torch.cat([xyz, pts[:, :, 3:]], dim=2)
    else:
        pts = xyz
    y = pts.transpose(2, 1)
    y = _conv_bn(sd, pre, "conv1", "bn1", y, True)
    y = _conv_bn(sd, pre, "conv2", "bn2", y, True)
    y = _conv_bn(sd, pre, "conv3", "bn3", y, False)          # NO ReLU on the last trunk layer (:162)
    return y.max(dim=2)[0], trans


# --------------------------------------------------------------------------- VQ
def vq_distances(z: Tensor, E: Tensor) -> Tensor:
    """d = sum(z^2) + sum(E^2) - 2 z E^T in the reference's association,
    network/vqvae/quantizer.py:46-48."""
    return torch.sum(z ** 2, dim=1, keepdim=True) + torch.sum(E ** 2, dim=1) - 2 * torch.matmul(z, E.t())


def vq_inference(E: Tensor, z: Tensor) -> Tuple[Tensor, Tensor]:
    """VectorQuantizer.forward(z, istrain=False), quantizer.py:30-54.
    Returns (idx [M,1] int64, z_q with z's shape)."""
    zf = z.reshape(-1, E.shape[1])
    d = vq_distances(zf, E)
    idx = torch.argmin(d, dim=1).unsqueeze(1)
    onehot = torch.zeros(idx.shape[0], E.shape[0]).scatter_(1, idx, 1)
    z_q = torch.matmul(onehot, E).view(z.shape)
    return idx, z_q


def vq_train_forward(E: Tensor, z: Tensor, beta: float, al: float):
    """VectorQuantizer.forward(z, istrain=True), quantizer.py:36-64:
    (loss, z_q (straight-through), perplexity, one-hot encodings, idx)."""
    zf = z.reshape(-1, E.shape[1])
    d = vq_distances(zf, E)
    idx = torch.argmin(d, dim=1).unsqueeze(1)
    onehot = torch.zeros(idx.shape[0], E.shape[0]).scatter_(1, idx, 1)
    z_q = torch.matmul(onehot, E).view(z.shape)
    loss = al * torch.mean((z_q.detach() - z) ** 2) + beta * torch.mean((z_q - z.detach()) ** 2)
    z_st = z + (z_q - z).detach()
    e_mean = torch.mean(onehot, dim=0)
    perplexity = torch.exp(-torch.sum(e_mean * torch.log(e_mean + 1e-10)))
    return loss, z_st, perplexity, onehot, idx


def vq_lookup(E: Tensor, idx: Tensor) -> Tensor:
    """Batched form of VectorQuantizer.get_emb (quantizer.py:68-75): one-hot @ E per sample,
    i.e. a row gather.  Raises (like scatter_) when an index is out of range."""
    idx = idx.reshape(-1)
    if idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= E.shape[0]):
        raise RuntimeError(f"index out of bounds for codebook with {E.shape[0]} rows")
    onehot = torch.zeros(idx.shape[0], E.shape[0]).scatter_(1, idx.view(-1, 1), 1)
    return torch.matmul(onehot, E)


# --------------------------------------------------------------------------- MLP decoder / encoder
def mlp_decoder(sd: SD, pre: str, z: Tensor) -> Tensor:
    """Decoder.forward, network/DVQVAE.py:169-185: Linear+ReLU ... Linear (no final activation)."""
    n = 0
    while f"{pre}MLP.L{n}.weight" in sd:
        n += 1
    x = z
    for i in range(n):
        x = F.linear(x, sd[f"{pre}MLP.L{i}.weight"], sd[f"{pre}MLP.L{i}.bias"])
        if i + 1 < n:
            x = F.relu(x)
    return x


def mlp_encoder(sd: SD, pre: str, x: Tensor) -> Tensor:
    """Encoder.forward, network/DVQVAE.py:145-166: (Linear+ReLU)* then linear_means."""
    i = 0
    while f"{pre}MLP.L{i}.weight" in sd:
        x = F.relu(F.linear(x, sd[f"{pre}MLP.L{i}.weight"], sd[f"{pre}MLP.L{i}.bias"]))
        i += 1
    return F.linear(x, sd[f"{pre}linear_means.weight"], sd[f"{pre}linear_means.bias"])


# --------------------------------------------------------------------------- gated PixelCNN
def _n_layers(sd: SD, pre: str) -> int:
    n = 0
    while f"{pre}layers.{n}.vert_stack.weight" in sd:
        n += 1
    return n


def _gate(x: Tensor) -> Tensor:
    a, b = x.chunk(2, dim=1)                                  # models.py:25-27
    return torch.tanh(a) * torch.sigmoid(b)


def pixelcnn_forward(sd: SD, pre: str, x: Tensor, label: Tensor) -> Tensor:
    """GatedPixelCNN.forward, network/pixelcnn/models.py:161-174 with GatedMaskedConv2d.forward
    :65-88.  x [B,H,W] int64, label [B] int64 -> logits [B,input_dim,H,W].
    Mask 'A' (layer 0) zeroes the last kernel row / column (:61-63); done on a copy here."""
    B, H, W = x.shape
    emb = F.embedding(x.reshape(-1), sd[f"{pre}embedding.weight"]).view(B, H, W, -1).permute(0, 3, 1, 2)
    x_v = x_h = emb
    for i in range(_n_layers(sd, pre)):
        lp = f"{pre}layers.{i}."
        wv, wh = sd[lp + "vert_stack.weight"], sd[lp + "horiz_stack.weight"]
        if i == 0:
            wv = wv.clone(); wv[:, :, -1] = 0
            wh = wh.clone(); wh[:, :, :, -1] = 0
        k = wv.shape[3]
        h = F.embedding(label, sd[lp + "class_cond_embedding.weight"])          # :70
        h_vert = F.conv2d(x_v, wv, sd[lp + "vert_stack.bias"], 1, (k // 2, k // 2))[:, :, :x_v.size(-1), :]
        out_v = _gate(h_vert + h[:, :, None, None])                              # :75
        h_horiz = F.conv2d(x_h, wh, sd[lp + "horiz_stack.bias"], 1, (0, k // 2))[:, :, :, :x_h.size(-2)]
        v2h = F.conv2d(h_vert, sd[lp + "vert_to_horiz.weight"], sd[lp + "vert_to_horiz.bias"])   # :79
        out = _gate(v2h + h_horiz + h[:, :, None, None])                         # :81
        out_h = F.conv2d(out, sd[lp + "horiz_resid.weight"], sd[lp + "horiz_resid.bias"])
        if i > 0:                                                                # residual only for i>=1 (:82-86)
            out_h = out_h + x_h
        x_v, x_h = out_v, out_h
    y = F.conv2d(x_h, sd[f"{pre}output_conv.0.weight"], sd[f"{pre}output_conv.0.bias"])
    y = F.relu(y)
    return F.conv2d(y, sd[f"{pre}output_conv.2.weight"], sd[f"{pre}output_conv.2.bias"])


def sample_from_logits(logits: Tensor, q: Tensor) -> Tensor:
    """probs = softmax(logits); draw = argmax(probs / q), q ~ Exp(1): the exponential-race form
    that torch.multinomial(1) itself evaluates (models.py:190-197; equivalence probed in SURVEY 3.5).
    The reference's global ``probs / probs.sum()`` (:194) is a uniform rescale and is dropped."""
    p = F.softmax(logits, -1)
    return torch.argmax(p / q, dim=-1)


def pixelcnn_generate(sd: SD, pre: str, label: Tensor, q: Tensor, shape=(3, 3), return_race_gap: bool = False):
    """GatedPixelCNN.generate, models.py:176-198, naive form: one full forward per grid position.
    label [B] int64, q [B, H*W, n_out] Exp(1) noise -> x [B,H,W] int64.  (x_start is ignored by the
    reference: the copy at :186 is commented out.)  ``return_race_gap``: also, per sample, the smallest relative margin
    by which a draw won its exponential race, 1 - (second largest p/q) / (largest p/q), evaluated in float64 (tests use it
    to set aside samples whose draw an fp32 rounding difference of the logits could flip)."""
    B = label.shape[0]
    H, W = shape
    x = torch.zeros(B, H, W, dtype=torch.int64)
    gap = torch.ones(B, dtype=torch.float64)
    for i in range(H):
        for j in range(W):
            logits = pixelcnn_forward(sd, pre, x, label)
            x[:, i, j] = sample_from_logits(logits[:, :, i, j], q[:, i * W + j])
            if return_race_gap:
                r = torch.topk(F.softmax(logits[:, :, i, j].double(), -1) / q[:, i * W + j].double(), 2, dim=-1)[0]
                gap = torch.minimum(gap, 1.0 - r[:, 1] / r[:, 0])
    return (x, gap.float()) if return_race_gap else x


# --------------------------------------------------------------------------- GenNet.gen
CODE_SLOTS = ((0, 1), (0, 2), (1, 1), (1, 2), (2, 1), (2, 2))    # gen_net.py:95-100 -> vqvae0..5


def gen(sd: SD, obj: Tensor, q: Tensor, mano, return_aux: bool = False):
    """GenNet.gen, network/gen_net.py:78-125, batched as B independent B=1 calls.
    obj [B,4,N] f32, q [B,9,n_out] Exp(1) noise, mano: callable(betas[B,10], pose[B,45]) -> verts [B,778,3]
    (global_orient = transl = 0).  Returns (recon [B,55], recon_pos [B,6])."""
    B = obj.shape[0]
    feat_type, _ = pointnet_encode(sd, "obj_encoder_type.", obj)                 # :81
    feat_pos, _ = pointnet_encode(sd, "obj_encoder_pos.", obj)                   # :82
    idx6, _ = vq_inference(sd["vqvae6.vector_quantization.embedding.weight"], feat_type)   # :83
    label = idx6[:, 0]                                                           # per-sample label
    n_cls = sd["GatedPixelCNN.layers.0.class_cond_embedding.weight"].shape[0]
    if int(label.max()) >= n_cls:
        raise RuntimeError("label out of range for the prior's class embedding")
    codes, race_gap = pixelcnn_generate(sd, "GatedPixelCNN.", label, q, return_race_gap=True)   # :92
    embs = [vq_lookup(sd[f"vqvae{k}.vector_quantization.embedding.weight"], codes[:, i, j])
            for k, (i, j) in enumerate(CODE_SLOTS)]                              # :95-106
    z_out = torch.cat(embs + [feat_type], dim=1)                                 # :109 raw feature, not obj_emb
    recon = mlp_decoder(sd, "decoder.", z_out).view(B, 55)                       # :112-113
    verts = mano(recon[:, :10], recon[:, 10:55])                                 # :116-118
    hand_feat, _ = pointnet_encode(sd, "recon_encoder.", verts.permute(0, 2, 1).contiguous())   # :120
    z_pos = torch.cat([hand_feat, feat_pos], dim=1)                              # :121
    recon_pos = mlp_decoder(sd, "pos_decoder.", z_pos).view(B, 6)                # :122-123
    if return_aux:
        E6 = sd["vqvae6.vector_quantization.embedding.weight"].double()
        d64 = (feat_type.double() ** 2).sum(1, keepdim=True) + (E6 ** 2).sum(1) - 2 * feat_type.double() @ E6.t()
        top2, top2_k = torch.topk(d64, 2, dim=1, largest=False)
        return recon, recon_pos, dict(feat_type=feat_type, feat_pos=feat_pos, idx6=idx6, codes=codes,
                                      verts=verts, hand_feat=hand_feat, race_gap=race_gap,
                                      idx6_gap=(top2[:, 1] - top2[:, 0]).float(),
                                      idx6_top2_spread=(E6[top2_k[:, 0]] - E6[top2_k[:, 1]]).norm(dim=1),   # |e_1 - e_2| (fp64)
                                      idx6_emax=float(E6.norm(dim=1).max()))
    return recon, recon_pos


def object_code_margin(aux, feat_type_other: Tensor) -> Tensor:
    """Per grasp, how large the fp64 top-2 distance gap of the object code must be before two fp32 evaluations of the path can
    be REQUIRED to agree on the argmin -- derived from what actually differs between them, not a round number:
      * the other path's PointNet feature differs from this one's by delta (measured, row by row): the two distances move
        apart by at most 2 |delta| |e_1 - e_2| (d_k = |z|^2 + |e_k|^2 - 2 z.e_k; |z|^2 cancels in the difference);
      * each fp32 distance is three 1024-term sums evaluated in some order: 32 ulps of (|z|^2 + |e|^2) covers both sides
        (observed: a few ulps).
    ``feat_type_other`` = the feature the path under test computed for the same clouds."""
    z = aux["feat_type"].double()
    delta = (feat_type_other.double().cpu() - z).norm(dim=1)
    fp32 = 32 * 2.0 ** -24 * ((z ** 2).sum(1) + aux["idx6_emax"] ** 2)
    return (2.0 * delta * aux["idx6_top2_spread"] + fp32).float()


def assemble61(recon: Tensor, recon_pos: Tensor) -> Tensor:
    """gen_diverse_grasp_obman.py:243-247: [betas(10) | global_orient(3) | pca_pose(45) | transl(3)]."""
    out = torch.zeros(recon.shape[0], 61)
    out[:, 0:10] = recon[:, 0:10]
    out[:, 10:13] = recon_pos[:, 0:3]
    out[:, 13:58] = recon[:, 10:55]
    out[:, 58:61] = recon_pos[:, 3:6]
    return out


# --------------------------------------------------------------------------- DVQVAE.forward (eval)
def _thumb_vertices():
    """The reference uses an undefined ``f0hand`` (network/DVQVAE.py:93).  The 83 MANO vertices not
    covered by the other five lists -- a documented assumption."""
    return [240] + list(range(248, 254)) + [266, 267, 286, 287] + list(range(697, 769))


def dvqvae_eval_forward(sd: SD, obj_pc: Tensor, hand_xyz: Tensor, parts) -> Tuple[Tensor, Tensor]:
    """DVQVAE.forward eval branch, network/DVQVAE.py:42-99,130-142.
    parts: the six vertex-index lists [f0..f4, handc].  Returns (emb_idx [7B,1] int64 in the order
    idx6, idx0..idx5, obj_emb [B,1024])."""
    hand = hand_xyz - hand_xyz.mean(dim=2, keepdim=True)                         # :48-51
    feat_type, _ = pointnet_encode(sd, "obj_encoder_type.", obj_pc)
    idxs = []
    for k, part in enumerate(parts):
        f, _ = pointnet_encode(sd, f"fing_{k}.", hand[:, :, part].contiguous())
        e = mlp_encoder(sd, f"emb_{k}.", f)
        i_k, _ = vq_inference(sd[f"vqvae{k}.vector_quantization.embedding.weight"], e)
        idxs.append(i_k)
    idx6, obj_emb = vq_inference(sd["vqvae6.vector_quantization.embedding.weight"], feat_type)
    return torch.cat([idx6] + idxs, dim=0), obj_emb


# --------------------------------------------------------------------------- helpers for tests
def exp1_noise(B: int, n_pos: int, n_out: int, seed: int) -> Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.empty(B, n_pos, n_out).exponential_(1.0, generator=g)
