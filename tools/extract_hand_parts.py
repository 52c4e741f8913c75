#!/usr/bin/env python3
"""Record the six MANO vertex subsets DVQVAE.forward feeds to its part encoders (network/DVQVAE.py:54-94)
by RUNNING the reference and spying on the tensor indexing -- the reference source is not read.
The undefined `f0hand` (DVQVAE.py:93) is supplied as the 83 vertices no other list covers.
Writes d-vqvae_amd/network/hand_parts.json (order: f0hand..f4hand, handc)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
import network.vqvae.quantizer as q
q.device = torch.device("cpu")
import network.DVQVAE as D

THUMB = [240] + list(range(248, 254)) + [266, 267, 286, 287] + list(range(697, 769))
D.f0hand = THUMB
net = D.DVQVAE().eval()
seen = []
orig = torch.Tensor.__getitem__
def spy(self, idx):
    if isinstance(idx, tuple) and len(idx) == 3 and isinstance(idx[2], list):
        seen.append([int(i) for i in idx[2]])
    return orig(self, idx)
torch.Tensor.__getitem__ = spy
with torch.no_grad():
    net(torch.randn(1, 4, 64), torch.randn(1, 3, 778))
torch.Tensor.__getitem__ = orig
assert len(seen) == 6 and seen[0] == THUMB
covered = set(sum(seen, []))
assert covered == set(range(778)), "partition does not cover the mesh"
out = os.path.join(ROOT, "d-vqvae_amd", "network", "hand_parts.json")
json.dump({"order": ["f0hand", "f1hand", "f2hand", "f3hand", "f4hand", "handc"], "parts": seen,
           "note": "f0hand is an assumption (undefined in the reference at HEAD)"}, open(out, "w"))
print("wrote", out, [len(s) for s in seen])
