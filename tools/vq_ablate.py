"""Diagnostic (GPU box): call time and phase stamps of the VQ streaming kernel under its timing-only ablations
(DVQ_VQ_ABL: 1 no DMA wait/issue, 2 no conversion, 4 no merge, 8 no scoring, 16 no MFMA, 32 no barrier; results invalid)."""
import os, sys, subprocess
here = os.path.dirname(os.path.abspath(__file__))
for abl in (0, 1, 2, 4, 8, 16, 32, 15, 47):
    env = dict(os.environ, DVQ_VQ_ABL=str(abl), DVQ_DIAG_LIB="1")   # diagnostics build: make -C d-vqvae_amd/csrc diag
    out = subprocess.run([sys.executable, os.path.join(here, "vq_phase_stamps.py")], env=env, capture_output=True, text=True).stdout
    keep = [l for l in out.splitlines() if l.startswith(("train", "prologue", "tile loop", "refine", "-- iteration t=2"))]
    print(f"=== ABL {abl}")
    print("\n".join(keep))
