"""Which (copy, channel) pairs differ from the exhaustive evaluation when one cloud fills the machine (the scenario of
test_pointnet_filter_full_machine_repeatability): channel, its lane in the publishing wave, the size of the difference, the fault counters."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dvqvae_amd
from dvqvae_amd import synth, ops
from test_gpu_parity import _gennet, _with_env
dev = torch.device("cuda:0")
net, _ = _gennet()
clouds = synth.synthetic_clouds(2048, 1024, seed=91).to(dev)
calls = int(os.environ.get("PROBE_CALLS", "10"))
for name, enc in (("pos", net.obj_encoder_pos), ("type", net.obj_encoder_type)):
    for sample in (1436, 1197):
        x = clouds[sample:sample + 1].repeat(4096, 1, 1).contiguous()
        want = _with_env("DVQ_PN_EXHAUSTIVE", "1", lambda: enc(x[:1].contiguous()))[0]
        ops.pointnet_fault_counters(reset=True)
        nbad = 0
        for call in range(calls):
            feat = enc(x)[0]
            d = (feat != want)
            if bool(d.any()):
                idx = d.nonzero()
                for cpy, ch in idx[:8].tolist():
                    print(f"{name} cloud {sample} call {call}: copy {cpy} channel {ch} (lane {ch & 63}, chunk64 {ch >> 6}) got {float(feat[cpy, ch]):.7g} want {float(want[0, ch]):.7g}")
                nbad += int(d.any(1).sum())
        print(f"{name} cloud {sample}: {nbad} bad copies in {calls} calls; fault counters {ops.pointnet_fault_counters()}", flush=True)
