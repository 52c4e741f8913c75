"""GatedPixelCNN mirror (reference: network/pixelcnn/models.py:30-88,130-198)."""
import torch
import torch.nn as nn

from ... import ops, packing
from .._base import PackedModule


def weights_init(m):
    if "Conv" in m.__class__.__name__ and hasattr(m, "weight") and isinstance(m.weight, torch.Tensor):
        nn.init.xavier_uniform_(m.weight.data)
        m.bias.data.fill_(0)


class GatedMaskedConv2d(nn.Module):
    """Parameter container of one gated layer (evaluated inside the fused sampler)."""

    def __init__(self, mask_type, dim, kernel, residual=True, n_classes=128):
        super().__init__()
        assert kernel % 2 == 1, "Kernel size must be odd"
        self.mask_type, self.residual = mask_type, residual
        half = kernel // 2
        self.class_cond_embedding = nn.Embedding(n_classes, 2 * dim)
        self.vert_stack = nn.Conv2d(dim, 2 * dim, (half + 1, kernel), 1, (half, half))
        self.vert_to_horiz = nn.Conv2d(2 * dim, 2 * dim, 1)
        self.horiz_stack = nn.Conv2d(dim, 2 * dim, (1, half + 1), 1, (0, half))
        self.horiz_resid = nn.Conv2d(dim, dim, 1)


class GatedPixelCNN(PackedModule):
    def __init__(self, input_dim=256, dim=128, n_layers=15, n_classes=128):
        super().__init__()
        self.dim = dim
        self.embedding = nn.Embedding(input_dim, dim)
        self.layers = nn.ModuleList(
            GatedMaskedConv2d("A" if i == 0 else "B", dim, 5 if i == 0 else 3, i != 0, n_classes) for i in range(n_layers))
        self.output_conv = nn.Sequential(nn.Conv2d(dim, 2048, 1), nn.ReLU(True), nn.Conv2d(2048, input_dim, 1))
        self.apply(weights_init)
        self._noise_stream = 0       # one Philox stream per generate() call that does not name its noise

    def _pack(self):
        return packing.PackedPixelCNN(self.state_dict())

    def forward(self, x, label):
        """x [B,3,3] int64, label [B] -> logits [B,input_dim,3,3]"""
        return ops.pixelcnn_forward(self.packed(), x, label)

    def generate(self, x_start, label, shape=(3, 3), batch_size=64, noise=None, return_logits=False, seed=None, row0=None,
                 stream_id=None):
        """Raster-order sampling of the 3x3 grid -> int64 [B,3,3].  ``x_start`` is ignored (as in the
        reference, models.py:186).  ``noise`` [B,9,input_dim] ~ Exp(1) makes the draw reproducible
        (argmax softmax/noise == multinomial(1)); drawn by the device Philox generator keyed by (seed, stream_id,
        row0 + b) when omitted (ops.exp1_noise).  Like the reference's multinomial draws, calls that name nothing draw FRESH
        noise every time: seed = torch.initial_seed() (so torch.manual_seed governs it), one stream per call, rows of this
        rank (ops.default_noise_key)."""
        if tuple(shape) != (3, 3):
            raise NotImplementedError("the grasp path samples a 3x3 latent grid (gen_net.py:92)")
        label = label.reshape(-1).contiguous()
        if label.shape[0] != batch_size:
            raise RuntimeError(f"generate: {label.shape[0]} labels for batch_size={batch_size}")
        pk = self.packed()
        if noise is None:
            dseed, drow = ops.default_noise_key()
            if stream_id is None:
                stream_id = self._noise_stream
                self._noise_stream += 1
            noise = ops.exp1_noise(batch_size, 9 * pk.n_in, dseed if seed is None else seed, drow if row0 is None else row0,
                                   stream_id, device=label.device).view(batch_size, 9, pk.n_in)
        return ops.pixelcnn_sample(pk, label, noise.contiguous(), return_logits=return_logits)
