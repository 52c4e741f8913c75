"""Registers, LDS and scratch of every kernel in the built library (CPU: reads the gfx950 code objects of the fat binary)."""
import os, re, struct, subprocess, sys, tempfile
LIB = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "d-vqvae_amd", "libdvq_hip.so")
pat = sys.argv[2] if len(sys.argv) > 2 else ""
objcopy, readelf = "/opt/rocm/lib/llvm/bin/llvm-objcopy", "/opt/rocm/lib/llvm/bin/llvm-readelf"
with tempfile.TemporaryDirectory() as tmp:
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([objcopy, "--dump-section", f".hip_fatbin={fat}", LIB, os.path.join(tmp, "copy.so")], check=True, capture_output=True)
    data = open(fat, "rb").read()
    magic, pos = b"__CLANG_OFFLOAD_BUNDLE__", 0
    while True:
        j = data.find(magic, pos)
        if j < 0:
            break
        pos = j + 1
        n, off = struct.unpack_from("<Q", data, j + 24)[0], j + 32
        for _ in range(n):
            o, size, ln = struct.unpack_from("<QQQ", data, off)
            name = data[off + 24: off + 24 + ln].decode()
            off += 24 + ln
            if "gfx950" not in name or not size:
                continue
            co = os.path.join(tmp, "dev.co")
            open(co, "wb").write(data[j + o: j + o + size])
            txt = subprocess.run([readelf, "--notes", co], capture_output=True, text=True).stdout
            for blk in txt.split("- .agpr_count:")[1:]:
                g = lambda k: (re.search(rf"\.{k}:\s*(\S+)", blk) or [None, "?"])[1]
                nm = g("name")
                if pat and pat not in nm:
                    continue
                dem = subprocess.run(["c++filt", nm], capture_output=True, text=True).stdout.strip()
                print(f"{dem[:90]:90s} vgpr {g('vgpr_count'):>4s} agpr {blk.split()[0]:>4s} sgpr {g('sgpr_count'):>4s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>5s}")
