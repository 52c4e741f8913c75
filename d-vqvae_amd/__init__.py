"""MI355X-native batched grasp generation (the D-VQVAE ``GenNet.gen`` hot path).

Layout: ``csrc/`` HIP kernels + the C-ABI (``include/dvq.h``); ``ops.py`` tensors -> raw pointers;
``network/`` the host-side mirror of the reference's ``network`` package (same class names, ctor
arguments, ``state_dict`` keys and ``forward`` signatures).
"""
__version__ = "0.1.0"
