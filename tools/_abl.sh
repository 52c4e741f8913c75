for a in 4; do
  touch d-vqvae_amd/csrc/vq_fast.hip
  make -C d-vqvae_amd/csrc EXTRA=-DDVQ_ABL=$a 2>&1 | grep -E "error" -A3
  echo "== ABL $a"; bash tools/_vq.sh;  python tools/_dbg.py 2>&1 | tail -5
done
