#!/usr/bin/env python3
"""Kernel descriptors of every gfx950 code object bundled in a HIP shared library: kernarg bytes, scratch, LDS, SGPR/VGPR counts.
usage: kernel_meta.py [lib.so]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "qingdai_amd", "libqingdai_hip.so")
    b = open(lib, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    i = b.find(magic)
    while i >= 0:
        n = struct.unpack_from("<Q", b, i + 24)[0]
        off = i + 32
        for _ in range(n):
            o, sz, tl = struct.unpack_from("<QQQ", b, off); off += 24
            t = b[off:off + tl].decode(); off += tl
            if "gfx950" in t and sz:
                with tempfile.NamedTemporaryFile(suffix=".co") as f:
                    f.write(b[i + o:i + o + sz]); f.flush()
                    notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
                for k in notes.split("- .agpr_count")[1:]:
                    def g(p):
                        m = re.search(p, k)
                        return m.group(1) if m else "?"
                    print("%-48s kernarg %5s scratch %5s lds %6s sgpr %3s vgpr %3s" % (
                        g(r"\.name:\s+(\S+)")[:48], g(r"\.kernarg_segment_size:\s+(\d+)"), g(r"\.private_segment_fixed_size:\s+(\d+)"),
                        g(r"\.group_segment_fixed_size:\s+(\d+)"), g(r"\.sgpr_count:\s+(\d+)"), g(r"\.vgpr_count:\s+(\d+)")))
        i = b.find(magic, i + 1)


if __name__ == "__main__":
    main()
