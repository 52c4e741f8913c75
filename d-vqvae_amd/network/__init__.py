"""Host-side mirror of the reference's ``network`` package for the grasp-generation path: same class
names, constructor arguments, ``state_dict`` keys/shapes and forward signatures; the arithmetic runs in
libdvq_hip.so (include/dvq.h).  Put ``d-vqvae_amd/`` on ``sys.path`` to import it under the reference's own
name (``from network.gen_net import GenNet``), or use ``dvqvae_amd.network``."""
