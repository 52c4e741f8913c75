"""VQVAE wrapper mirror (reference: network/VQVAE.py:11-53)."""
import torch.nn as nn

from .vqvae.quantizer import VectorQuantizer


class VQVAE(nn.Module):
    def __init__(self, h_dim, res_h_dim, n_res_layers, n_embeddings, embedding_dim, beta, a=1,
                 save_img_embedding_map=False):
        super().__init__()          # h_dim / res_h_dim / n_res_layers are accepted and ignored, as in the reference
        self.vector_quantization = VectorQuantizer(n_embeddings, embedding_dim, beta, al=a)
        self.img_to_embedding_map = {i: [] for i in range(n_embeddings)} if save_img_embedding_map else None

    def forward(self, inputs, verbose=False):
        assert not verbose
        loss, z_q, perplexity, _, _ = self.vector_quantization(inputs, True)
        return loss, z_q, perplexity

    def inference(self, inputs, verbose=False):
        return self.vector_quantization(inputs, False)

    def get_embbeding(self, index, dim):
        return self.vector_quantization.get_emb(index, dim)
