"""VectorQuantizer mirror (reference: network/vqvae/quantizer.py:10-75)."""
import torch
import torch.nn as nn

from ... import ops


class VectorQuantizer(nn.Module):
    def __init__(self, n_e, e_dim, beta, al):
        super().__init__()
        self.n_e, self.e_dim, self.beta, self.al = n_e, e_dim, beta, al
        self.embedding = nn.Embedding(n_e, e_dim)
        self.embedding.weight.data.uniform_(-1.0 / n_e, 1.0 / n_e)

    def _packed_codebook(self, E):
        """Fast-path image of the codebook, rebuilt whenever the parameter is replaced, moved or edited in place."""
        if not ops.vq_fast_supported(self.n_e, self.e_dim):
            return None
        w = self.embedding.weight
        key = (w.data_ptr(), w._version, str(w.device))
        cached = getattr(self, "_pack_cache", None)
        if cached is None or cached[0] != key:
            cached = (key, ops.vq_pack(E))
            object.__setattr__(self, "_pack_cache", cached)
        return cached[1]

    def _quantize(self, z):
        E = self.embedding.weight.detach()
        zf = z.detach().reshape(-1, self.e_dim)
        idx = ops.vq_argmin(zf if zf.is_contiguous() else zf.contiguous(), E, packed=self._packed_codebook(E))
        z_q = ops.vq_lookup(E, idx).view(z.shape)
        return idx.unsqueeze(1), z_q

    def forward(self, z, istrain):
        """eval: (idx [M,1] int64, z_q).  train: (loss, z_q straight-through, perplexity, one-hot [M,K], idx)
        (quantizer.py:30-64; the loss/perplexity arithmetic is elementwise glue on the device)."""
        idx, z_q = self._quantize(z)
        if not istrain:
            return idx, z_q
        z_q = self.embedding(idx.squeeze(1)).view(z.shape)          # differentiable gather for the codebook loss
        loss = self.al * torch.mean((z_q.detach() - z) ** 2) + self.beta * torch.mean((z_q - z.detach()) ** 2)
        z_st = z + (z_q - z).detach()
        onehot = torch.zeros(idx.shape[0], self.n_e, device=z.device).scatter_(1, idx, 1)
        e_mean = onehot.mean(dim=0)
        perplexity = torch.exp(-torch.sum(e_mean * torch.log(e_mean + 1e-10)))
        return loss, z_st, perplexity, onehot, idx

    def get_emb(self, min_encoding_indices, dim):
        """Batched codebook lookup: idx [B] -> [B, dim] (the reference form only works for B=1, quantizer.py:68-75).
        Raises RuntimeError for an index >= n_e like the reference's scatter_."""
        idx = min_encoding_indices.reshape(-1)
        return ops.vq_lookup(self.embedding.weight.detach(), idx).view(idx.shape[0], dim)
