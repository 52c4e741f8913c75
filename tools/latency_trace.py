"""B = 1 GenNet.gen calls under rocprofv3 --kernel-trace: kernel durations and the gaps between them (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvqvae_amd
from dvqvae_amd import mano as dmano, synth
from dvqvae_amd.network.gen_net import GenNet
dev = torch.device("cuda:0")
K = 512
net = GenNet(n_embeddings=K, prior_tokens=K, prior_classes=K)
net.load_state_dict(synth.synthetic_state_dict(net.state_dict(), 1234)); net.eval().to(dev)
net.set_rh_mano(dmano.ManoLayer(dmano.synthetic_mano_arrays()).to(dev))
B = int(os.environ.get("LAT_B", "1"))
obj = synth.synthetic_clouds(B, 1024, seed=1).to(dev)
for _ in range(6): net.gen(obj)
torch.cuda.synchronize()
