import numpy as np
import torch

from conftest import SEED
from dvqvae_amd import synth


def templates_from(module):
    return {k: v.detach().clone() for k, v in module.state_dict().items()}


def load_synth(module, seed):
    sd = synth.synthetic_state_dict(module.state_dict(), seed)
    module.load_state_dict(sd, strict=True)
    module.eval()
    return sd


def assert_close(a, b, atol=1e-5, rtol=0.0, what=""):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol, err_msg=what)
