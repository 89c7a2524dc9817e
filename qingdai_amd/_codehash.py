"""
qingdai_amd/_codehash.py -- identity of the DEVICE code a profile was measured on.

A committed rocprofv3 summary (profiles/rNN_fused_kernels.json) says which kernels it timed; bench.py flags it `profile_stale` when
those kernels have changed since.  Hashing source files flags host-only edits too (round 3: a getenv moved, the kernels did not
change, the driver's line said stale), so the stamp is the SHA-256 of the `.hip_fatbin` section -- the gfx950 code objects hipcc
embeds -- of the translation units that hold the profiled kernels (their .o files, kept next to the sources by the Makefile), with
the whole library's section as a fallback.  Pure Python (a 64-bit little-endian ELF section walk): no tool needed on the GPU box.
"""
import hashlib
import os
import struct

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libqingdai_hip.so")
PROFILED_OBJECTS = ("qd_stream.o", "qd_ocntail.o")          # k_dyn_stream / k_ocn_stream, k_ocn_tail_fast


def elf_section(path, name):
    """bytes of section `name` of a 64-bit little-endian ELF file, or None"""
    with open(path, "rb") as fh:
        d = fh.read()
    if d[:4] != b"\x7fELF" or d[4] != 2 or d[5] != 1:
        return None
    shoff, = struct.unpack_from("<Q", d, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", d, 0x3A)
    if shoff == 0 or shnum == 0:
        return None

    def sh(i):
        return struct.unpack_from("<IIQQQQIIQQ", d, shoff + i * shentsize)       # name, type, flags, addr, offset, size, ...
    stroff, strsize = sh(shstrndx)[4], sh(shstrndx)[5]
    strtab = d[stroff:stroff + strsize]
    for i in range(shnum):
        h = sh(i)
        end = strtab.find(b"\0", h[0])
        if strtab[h[0]:end].decode("ascii", "replace") == name:
            return d[h[4]:h[4] + h[5]]
    return None


def fatbin_sha(path):
    try:
        sec = elf_section(path, ".hip_fatbin")
    except OSError:
        return None
    return hashlib.sha256(sec).hexdigest()[:16] if sec else None


def device_code_stamp():
    """{file name: sha256[:16] of its .hip_fatbin} for the profiled translation units (when their objects exist) and the library"""
    out = {}
    for o in PROFILED_OBJECTS:
        h = fatbin_sha(os.path.join(CSRC, o))
        if h:
            out[o] = h
    h = fatbin_sha(LIB)
    if h:
        out["libqingdai_hip.so"] = h
    return out


def stale_against(stamp):
    """True when the device code of the profiled kernels differs from `stamp` (a dict written by device_code_stamp()).  The object
    files decide when both sides have them; otherwise the whole library's device code."""
    now = device_code_stamp()
    objs = [o for o in PROFILED_OBJECTS if o in stamp and o in now]
    if objs:
        return any(stamp[o] != now[o] for o in objs)
    if "libqingdai_hip.so" in stamp and "libqingdai_hip.so" in now:
        return stamp["libqingdai_hip.so"] != now["libqingdai_hip.so"]
    return True
