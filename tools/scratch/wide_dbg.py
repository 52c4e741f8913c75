import sys, os, subprocess, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
if len(sys.argv) > 1:
    import dvqvae_amd
    from dvqvae_amd import ops, packing
    from dvqvae_amd.network.pixelcnn.models import GatedPixelCNN
    from util import load_synth
    dev = "cuda:0"; out = {}
    torch.manual_seed(0)
    for (M, N, K) in ((300, 512, 512), (4, 512, 1024), (16384, 1024, 512)):
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
        out[f"lin{M}x{N}x{K}"] = ops.linear(x, w, b, planes=packing.split_bf16x3(w)).cpu()
    for L in (1, 2, 3, 4, 8, 15, 15):
        net = GatedPixelCNN(512, 512, L, 128); load_synth(net, 5); net = net.to(dev)
        g = torch.Generator().manual_seed(1)
        x = torch.randint(0, 512, (3, 3, 3), generator=g).to(dev); lab = torch.randint(0, 128, (3,), generator=g).to(dev)
        out[f"pix{L}_{len(out)}"] = net(x, lab).cpu()
    torch.save(out, sys.argv[1])
else:
    for tag, env in (("a", "0"), ("b", "1")):
        e = dict(os.environ); e["DVQ_GEMM_WIDE"] = env
        subprocess.check_call([sys.executable, __file__, f"/tmp/wide_{tag}.pt"], env=e)
    a, b = torch.load("/tmp/wide_a.pt"), torch.load("/tmp/wide_b.pt")
    for k in a: print(k, "max diff", float((a[k] - b[k]).abs().max()), "scale", float(a[k].abs().max()))
