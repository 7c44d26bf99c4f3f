#!/usr/bin/env python3
"""Print register / LDS / spill figures of the kernels in a hipcc object file (uncompressed or
zstd-compressed clang offload bundle), filtered by substring."""
import re, struct, subprocess, sys, tempfile, os
obj, pats = sys.argv[1], sys.argv[2:]
tmp = tempfile.mkdtemp()
fat = os.path.join(tmp, "fat.bin")
subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
d = open(fat, "rb").read()
if d[:4] == b"CCOB":
    import zstandard  # noqa
    raise SystemExit("compressed bundle: rebuild with --no-offload-compress")
assert d[:24] == b"__CLANG_OFFLOAD_BUNDLE__", d[:24]
n = struct.unpack_from("<Q", d, 24)[0]
off = 32
for _ in range(n):
    o, s, tl = struct.unpack_from("<QQQ", d, off); off += 24
    triple = d[off:off + tl].decode(); off += tl
    if "gfx" not in triple: continue
    co = os.path.join(tmp, "k.co"); open(co, "wb").write(d[o:o + s])
    t = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    for e in re.split(r"\n\s+- \.agpr_count", t)[1:]:
        name = re.search(r"\.name:\s+(\S+)", e).group(1)
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if pats and not all(p in dn for p in pats): continue
        g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, e).group(1)
        print(f"vgpr {g('vgpr_count'):>4} sgpr {g('sgpr_count'):>4} vspill {g('vgpr_spill_count'):>4} sspill {g('sgpr_spill_count'):>4} "
              f"lds {g('group_segment_fixed_size'):>7} scratch {g('private_segment_fixed_size'):>5}  {dn[:150]}")
