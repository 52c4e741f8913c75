"""GenNet mirror (reference: network/gen_net.py:13-125): the batched grasp-generation graph.

Batched semantics = B independent B=1 reference calls (the reference's own ``gen`` only works for B=1:
per-sample label instead of ``idx6[:,0,0]`` of sample 0, row-gather lookup instead of ``view(1,dim)``)."""
import os

import torch
import torch.nn as nn

from .. import _lib, ops, packing
from .DVQVAE import Decoder, _codebooks
from .pixelcnn.models import GatedPixelCNN
from .pointnet_encoder import PointNetEncoder

CODE_SLOTS = ((0, 1), (0, 2), (1, 1), (1, 2), (2, 1), (2, 2))     # grid position -> vqvae0..5 (gen_net.py:95-100)


class GenNet(nn.Module):
    def __init__(self, n_embeddings=128, prior_tokens=512, prior_dim=512, prior_layers=15, prior_classes=128):
        """Defaults are the reference's fixed sizes (gen_net.py:17-34).  ``n_embeddings`` = 512 builds the
        K=512 codebooks of the synthetic benchmark (prior emits 512 classes, checkpoint codebooks 128)."""
        super().__init__()
        self.obj_encoder_type = PointNetEncoder(global_feat=True, feature_transform=False, channel=4)
        self.obj_encoder_pos = PointNetEncoder(global_feat=True, feature_transform=False, channel=4)
        _codebooks(self, n_embeddings)
        self.decoder = Decoder(layer_sizes=[1024, 256, 55], latent_size=2560)
        self.num = 0
        self.rh_mano = None
        self.recon_encoder = PointNetEncoder(global_feat=True, feature_transform=False, channel=3)
        self.pos_decoder = Decoder(layer_sizes=[1024, 128, 6], latent_size=2048)
        self.GatedPixelCNN = GatedPixelCNN(prior_tokens, prior_dim, prior_layers, prior_classes)
        self.noise_seed = None       # key of the device Philox generator that replaces multinomial's draws (set_noise_seed);
        #                              None: torch.initial_seed(), i.e. torch.manual_seed governs the draws as it does the reference's
        self._noise_stream = 0       # one Philox stream per gen() call unless the caller names it
        self.sort_by_label = os.environ.get("DVQ_SORT_LABELS", "1") != "0"   # prior evaluated in label order (gather locality)
        self.range_fallbacks = 0     # gen() calls in which rows were generated again on the bf16 split (an activation left fp16's range)
        self.range_fallback_rows = 0 # ... and how many rows that was

    def set_noise_seed(self, seed, first_stream=0):
        """Seed of the prior's sampling noise; every gen() call without explicit ``noise`` / ``stream_id`` uses the next stream."""
        self.noise_seed = int(seed)
        self._noise_stream = int(first_stream)
        return self

    def set_rh_mano(self, Rh_mano):
        Rh_mano.eval()
        self.rh_mano = Rh_mano

    # ------------------------------------------------------------------------------------------
    def _hand_vertices(self, recon):
        """rh_mano(betas, global_orient=0, hand_pose, transl=0).vertices as [B,3,778] (gen_net.py:116-120)."""
        m = self.rh_mano
        if m is None:
            raise RuntimeError("GenNet.gen: call set_rh_mano(layer) first (gen_diverse_grasp_obman.py:361)")
        if hasattr(m, "vertices_channel_major"):
            return m.vertices_channel_major(recon[:, :10], recon[:, 10:55])
        zero = torch.zeros(recon.shape[0], 3, device=recon.device)
        v = m(betas=recon[:, :10], global_orient=zero, hand_pose=recon[:, 10:55], transl=zero).vertices
        return v.detach().permute(0, 2, 1).contiguous()

    def _decode(self, codes, feat_type, feat_pos_into, err):
        """codes [B,3,3] + object feature -> recon [B,55]; shared by gen and gen_byid."""
        B, dev = codes.shape[0], codes.device
        z_out = feat_type["z_out"]
        flat = codes.view(B, 9)
        for k, (i, j) in enumerate(CODE_SLOTS):
            E = getattr(self, f"vqvae{k}").vector_quantization.embedding.weight.detach()
            ops.vq_lookup(E, flat[:, i * 3 + j], out=z_out[:, 256 * k: 256 * (k + 1)], err=err)
        return self.decoder(z_out).view(B, 55)

    def _noise_key(self, seed, row0, stream_id):
        """(seed, first global row, stream) of the device Philox generator under gen()'s key rules."""
        if stream_id is None:
            stream_id = self._noise_stream
            self._noise_stream += 1
        dseed, drow = ops.default_noise_key()
        if seed is None:
            seed = dseed if self.noise_seed is None else self.noise_seed
        return seed, (drow if row0 is None else row0), stream_id

    def _draw_noise(self, B, dev, key, perm=None):
        """[B, 9, prior_tokens] Exp(1) variates; with ``perm``, row r holds the draws of row perm[r] of the keyed batch."""
        n_in = self.GatedPixelCNN.packed().n_in
        seed, row0, stream_id = key
        return ops.exp1_noise(B, 9 * n_in, seed, row0, stream_id, device=dev, perm=perm).view(B, 9, n_in)

    def _gen_impl(self, obj, noise, key=None, rows=None):
        """The device work of gen(): no host synchronisation inside (gen() checks the error flag once at the end).
        ``noise`` None: the prior's draws come from the device generator under ``key``, drawn directly in the order the prior
        is evaluated in (no gather of the [B, 9, tokens] tensor).  ``rows`` (int64 [B] on the device): ``obj`` holds rows
        ``rows`` of the keyed batch -- sample b draws the noise of row ``rows[b]`` (the per-row range fallback of gen())."""
        if obj.dim() != 3:
            raise RuntimeError(f"gen: expected obj [B,4,N], got {tuple(obj.shape)}")
        B, dev = obj.shape[0], obj.device
        obj = obj.contiguous()
        z_out = torch.empty(B, 2560, device=dev, dtype=torch.float32)      # [emb0..5 | obj_type_feature] (:109)
        z_pos = torch.empty(B, 2048, device=dev, dtype=torch.float32)      # [hand_feat | obj_pos_feature]  (:121)
        self.obj_encoder_type(obj, out=z_out[:, 1536:])                    # :81 written in place
        self.obj_encoder_pos(obj, out=z_pos[:, 1024:])                     # :82
        feat_type = z_out[:, 1536:]
        idx6, _ = self.vqvae6.inference(feat_type)                         # :83 (obj_emb unused, as in the reference)
        label = idx6[:, 0].contiguous()                                    # per-sample label
        err = ops.new_err_flag(dev)
        pk = self.GatedPixelCNN.packed()
        # The gated GEMMs add the class-conditional row cls[label[m]] in their epilogue: with rows in arrival order every lane of a
        # store instruction gathers from a different 4 KB row of the table; sorted by label a 128-row tile holds one or two labels
        # and the gather is a broadcast again (measured: -10 % on the gated GEMMs at 65 536 grasps with 123 distinct object codes).
        # Rows are independent, so the order changes no result; the codes are scattered back.
        if B >= 512 and self.sort_by_label:
            order = torch.argsort(label, stable=True)
            noise_s = (self._draw_noise(B, dev, key, perm=order if rows is None else rows[order].contiguous()) if noise is None
                       else noise.index_select(0, order))
            codes_s = ops.pixelcnn_sample(pk, label[order].contiguous(), noise_s, err=err)   # :92
            codes = torch.empty_like(codes_s)
            codes[order] = codes_s
        else:
            if noise is None:
                noise = self._draw_noise(B, dev, key, perm=rows)
            codes = ops.pixelcnn_sample(pk, label, noise.contiguous(), err=err)   # :92
        # a position drawn from all-NaN logits carries -1 (bit 2 of err is set; gen() regenerates those rows): decode token 0 there
        recon = self._decode(codes.clamp_min(0), {"z_out": z_out}, None, err)   # :95-113
        verts = self._hand_vertices(recon)                                 # :116-118
        self.recon_encoder(verts, out=z_pos[:, :1024])                     # :120
        recon_pos = self.pos_decoder(z_pos).view(B, 6)                     # :122-123
        aux = dict(idx6=idx6, codes=codes, feat_type=feat_type, feat_pos=z_pos[:, 1024:], verts=verts, hand_feat=z_pos[:, :1024])
        return recon, recon_pos, aux, err

    _RANGE_ERROR = ("GenNet.gen: code or label index out of range (prior classes vs codebook rows, "
                    "gen_net.py:20-34); build GenNet(n_embeddings=...) to match the prior")

    @torch.no_grad()
    def gen(self, obj, noise=None, return_aux=False, seed=None, row0=None, stream_id=None, check=True):
        """obj [B,4,N] f32 on the GPU -> (recon [B,55], recon_pos [B,6]).
        ``noise`` [B,9,prior_tokens] ~ Exp(1) fixes the prior's draws (parity runs).  Without it the draws come from the
        device Philox generator keyed by (seed, stream_id, row0 + b): a batch sharded over ranks (``row0`` = first global
        row of the shard, same ``seed`` / ``stream_id``) generates exactly what the unsharded call generates.
        Defaults: seed = set_noise_seed's, else torch.initial_seed(); one stream per call; rows of this rank
        (ops.default_noise_key), so ranks that name nothing never share noise.
        ``check=False``: no host synchronisation at all -- the call returns as soon as the work is enqueued (a loop of B = 1 calls
        then overlaps the host side of call i + 1 with the device side of call i); the caller gives up the index-range error
        and the fp16-range fallback below, and gets ``aux["err"]`` (device int32: bit 0 range, bit 2 all-NaN logits) to check later."""
        if obj.dim() != 3:
            raise RuntimeError(f"gen: expected obj [B,4,N], got {tuple(obj.shape)}")
        key = self._noise_key(seed, row0, stream_id) if noise is None else None
        with ops.no_range_check():                                         # ONE check for the whole path, below
            recon, recon_pos, aux, err = self._gen_impl(obj, noise, key)
        aux["err"] = err
        if not check:
            return (recon, recon_pos, aux) if return_aux else (recon, recon_pos)
        # one host synchronisation per call: the index-range flag and "every parameter is finite"
        status = int((err + 2 * (~(torch.isfinite(recon).all() & torch.isfinite(recon_pos).all())).to(torch.int32)).item())
        if status & 1:
            raise RuntimeError(self._RANGE_ERROR)
        if (status & 6) and packing.gemm_kind() == _lib.PLANES_F16X2:     # non-finite parameters, or a draw from all-NaN logits (bit 2)
            # The default GEMM arithmetic splits activations into fp16 pieces: a value beyond fp16's range (|x| >= 65 520) turns its
            # ROW into NaN -- never a silently wrong number (rows do not interact anywhere on the path).  Exactly those rows -- a
            # non-finite parameter, or a -1 where the sampler drew from all-NaN logits -- are generated again on the six-product bf16
            # split, which has fp32's range (csrc/gemm_f16x2.hip), under the same noise keys, and scattered back; every other row keeps
            # its bits.  NaN / Inf INPUTS come out non-finite there too, as in the reference.
            bad = ~(torch.isfinite(recon).all(dim=1) & torch.isfinite(recon_pos).all(dim=1)) | (aux["codes"].reshape(obj.shape[0], -1) < 0).any(dim=1)
            rows = bad.nonzero().reshape(-1)
            self.range_fallbacks += 1
            self.range_fallback_rows += int(rows.numel())
            with ops.no_range_check(), packing.gemm_kind_as(_lib.PLANES_BF16X3):
                r2, p2, aux2, err2 = self._gen_impl(obj.index_select(0, rows), None if noise is None else noise.index_select(0, rows), key, rows=rows)
                if int(err2.item()) & 1:
                    raise RuntimeError(self._RANGE_ERROR)
            recon[rows], recon_pos[rows] = r2, p2
            for k in ("idx6", "codes", "feat_type", "feat_pos", "verts", "hand_feat"):
                aux[k][rows] = aux2[k]
            aux["err"] = err2
            aux["fallback_rows"] = rows
        return (recon, recon_pos, aux) if return_aux else (recon, recon_pos)

    @torch.no_grad()
    def gen_byid(self, idx6, noise=None):
        """Debug variant (gen_net.py:41-75): samples codes for a given object code, then decodes an all-zero
        latent (as the reference does) and returns zero wrist parameters."""
        idx6 = idx6.reshape(-1).contiguous()
        B, dev = idx6.shape[0], idx6.device
        self.vqvae6.get_embbeding(idx6, 1024)                              # range check, value unused (:45)
        self.GatedPixelCNN.generate(None, idx6, shape=(3, 3), batch_size=B, noise=noise)
        recon = self.decoder(torch.zeros(B, 2560, device=dev)).view(B, 55)
        return recon, torch.zeros(B, 6, device=dev)

