#!/usr/bin/env python3
"""Registers / LDS / scratch of every gfx950 kernel in a host object or shared library (from the code objects' metadata notes).
   python tools/kernel_resources.py build/igemm_split.o [name-substring ...]"""
import re, struct, subprocess, sys, tempfile, os

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_objects(path):
    data = open(path, "rb").read()
    out = []
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data):
        p = m.start()
        (n,) = struct.unpack_from("<Q", data, p + 24)
        off = p + 32
        for _ in range(n):
            o, sz, tl = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tl].decode()
            off += tl
            if "gfx950" in triple and sz > 0:
                out.append(data[p + o:p + o + sz])
    return out


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(co)
        txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        os.unlink(f.name)
        for blk in re.split(r"\n\s+- ", txt):
            nm = re.search(r"\.name:\s+(\S+)", blk)
            if not nm or ".vgpr_count" not in blk:
                continue
            name = subprocess.run(["c++filt", nm.group(1)], capture_output=True, text=True).stdout.strip()
            if pats and not any(p in name for p in pats):
                continue
            g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1)) if re.search(rf"\.{k}:\s+(\d+)", blk) else -1
            print(f"vgpr {g('vgpr_count'):4d} agpr {g('agpr_count'):4d} sgpr {g('sgpr_count'):4d} spill {g('vgpr_spill_count'):3d} scratch {g('private_segment_fixed_size'):5d} "
                  f"lds {g('group_segment_fixed_size'):6d}  {name[:170]}")


if __name__ == "__main__":
    main()
