"""vq_rows.hip timing (GPU box): the product library or a variant built with -DVQR_ABL=n as libdvq_hip_exp<name>.so (DVQ_EXP=<name>);
also the number of rows that took the all-entries scan."""
import os, sys
sys.path.insert(0, "/root/repo")
from dvqvae_amd import _lib
if os.environ.get("DVQ_EXP"): _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libdvq_hip_exp" + os.environ["DVQ_EXP"] + ".so")
import torch
from dvqvae_amd import ops
dev = "cuda:0"
M, D, K = 65536, 256, 512
torch.manual_seed(0)
zs = [torch.randn(M, D, device=dev) for _ in range(6)]
E = torch.randn(K, D, device=dev)
pk = ops.vq_pack(E)
os.environ["DVQ_VQ_KERNEL"] = "32"; _lib.load().dvq_reload_env()
for i in range(6): idx = ops.vq_argmin(zs[i], E, packed=pk)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(30): idx = ops.vq_argmin(zs[i % 6], E, packed=pk)
e1.record(); torch.cuda.synchronize()
print("variant", os.environ.get("DVQ_EXP", "product"), "%.2f us per call" % (e0.elapsed_time(e1) * 1e3 / 30))
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
for kern in ("32", "8"):
    os.environ["DVQ_VQ_KERNEL"] = kern; _lib.load().dvq_reload_env()
    cnt.zero_()
    ops.vq_argmin(zs[0], E, packed=pk, slow_rows=cnt)
    print("kernel", kern, "slow rows in one call:", int(cnt.item()))
