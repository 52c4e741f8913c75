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
        """Fast-path image of the codebook, rebuilt whenever the parameter is replaced, moved or edited in place
        (load_state_dict, optimizer steps, no_grad in-place ops: anything that bumps the tensor version; writes
        through ``weight.data`` bypass the version counter -- call ``invalidate_pack()`` after those)."""
        if not ops.vq_fast_supported(self.n_e, self.e_dim):
            return None
        w = self.embedding.weight
        key = (w.data_ptr(), w._version, str(w.device))
        cached = getattr(self, "_pack_cache", None)
        if cached is None or cached[0] != key:
            cached = (key, ops.vq_pack(E))
            object.__setattr__(self, "_pack_cache", cached)
        return cached[1]

    # The fast kernel decides ~all rows of a well-conditioned problem from short candidate lists; on an ill-conditioned
    # one (|z| far above the codebook's spread, e.g. the reference's U(+-1/n_e) initial codebook against O(1)
    # features) most rows need its slow all-entries scan and the exact kernel is faster.  The kernel counts those rows
    # on the device; the count is fetched without a sync every few calls and the module switches to the exact kernel
    # (same indices) once more than 1/16 of a window's rows were slow.  State lives with the packed codebook.
    SLOW_FRACTION = 1.0 / 16.0
    PROBE_EVERY = 8

    def _regime(self, packed):
        st = getattr(self, "_regime_state", None)
        if st is None or st["packed"] is not packed:
            dev = packed.device
            st = dict(packed=packed, counter=torch.zeros(1, dtype=torch.int64, device=dev),
                      host=torch.zeros(1, dtype=torch.int64).pin_memory(), event=None, calls=0, rows=0,
                      rows_at_copy=0, rows_seen=0, slow_seen=0, prefer_exact=False)
            object.__setattr__(self, "_regime_state", st)
        return st

    def _regime_update(self, st, M):
        st["rows"] += M
        ev = st["event"]
        if ev is not None and ev.query():
            slow = int(st["host"][0])
            d_slow, d_rows = slow - st["slow_seen"], st["rows_at_copy"] - st["rows_seen"]
            if d_rows > 0 and d_slow > self.SLOW_FRACTION * d_rows:
                st["prefer_exact"] = True
            st["slow_seen"], st["rows_seen"], st["event"] = slow, st["rows_at_copy"], None
        if st["event"] is None and st["calls"] % self.PROBE_EVERY == 0:
            dev = st["counter"].device                       # copy and event on the stream the kernels of THIS device run on
            with torch.cuda.device(dev):
                st["host"].copy_(st["counter"], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
            st["event"], st["rows_at_copy"] = ev, st["rows"]
        st["calls"] += 1

    def invalidate_pack(self):
        object.__setattr__(self, "_pack_cache", None)
        object.__setattr__(self, "_regime_state", None)

    def _quantize(self, z):
        E = self.embedding.weight.detach()
        zf = z.detach().reshape(-1, self.e_dim)
        zf = zf if zf.is_contiguous() else zf.contiguous()
        packed = self._packed_codebook(E)
        if packed is None or zf.shape[0] == 0:
            idx = ops.vq_argmin(zf, E)
        else:
            st = self._regime(packed)
            if st["prefer_exact"]:
                idx = ops.vq_argmin(zf, E, fast=False)
            else:
                idx = ops.vq_argmin(zf, E, packed=packed, slow_rows=st["counter"])
                if not torch.cuda.is_current_stream_capturing():       # the probe copies to the host: not inside a graph capture
                    self._regime_update(st, zf.shape[0])
        # idx comes from the argmin kernels: in range by construction, so the flag is supplied and never read (no host sync here)
        z_q = ops.vq_lookup(E, idx, err=ops.new_err_flag(zf.device)).view(z.shape)
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
