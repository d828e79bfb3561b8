"""Register / scratch use of every kernel in the built library, read from the gfx950 code objects embedded in
libvc_hip.so (clang offload bundles -> ELF notes -> AMDGPU msgpack metadata).

    python tools/check_spills.py [path/to/libvc_hip.so]     # lists kernels that use scratch memory (spills)

A kernel that spills at the 256-register limit usually runs markedly slower; the build prints this list and
tests/test_abi.py holds it to the kernels known to need scratch.
"""
import os
import struct
import sys

import msgpack

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(blob):
    """Yields (triple, bytes) of every bundle entry in every offload bundle of `blob`."""
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        n = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tsz = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tsz].decode()
            p += 24 + tsz
            yield triple, blob[pos + off:pos + off + size]
        pos += len(MAGIC)


def kernel_metadata(elf):
    """AMDGPU metadata dictionaries ('.name', '.vgpr_count', '.private_segment_fixed_size', ...) of an ELF code object."""
    if elf[:4] != b"\x7fELF":
        return []
    shoff = struct.unpack_from("<Q", elf, 0x28)[0]
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    out = []
    for i in range(shnum):
        sh = shoff + i * shentsize
        sh_type = struct.unpack_from("<I", elf, sh + 4)[0]
        off, size = struct.unpack_from("<QQ", elf, sh + 0x18)
        if sh_type != 7:  # SHT_NOTE
            continue
        p = off
        while p + 12 <= off + size:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            name = elf[p + 12:p + 12 + namesz].rstrip(b"\0")
            d0 = p + 12 + (namesz + 3) // 4 * 4
            if name == b"AMDGPU" and ntype == 32:
                md = msgpack.unpackb(elf[d0:d0 + descsz], raw=False, strict_map_key=False)
                out.extend(md.get("amdhsa.kernels", []))
            p = d0 + (descsz + 3) // 4 * 4
    return out


def kernels(path):
    blob = open(path, "rb").read()
    ks = []
    for triple, obj in code_objects(blob):
        if "gfx950" in triple:
            ks.extend(kernel_metadata(obj))
    return ks


def main():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "video-classification_amd", "libvc_hip.so")
    ks = kernels(path)
    spill = [k for k in ks if k.get(".private_segment_fixed_size", 0) > 0]
    print(f"{len(ks)} kernels in {os.path.basename(path)}; {len(spill)} use scratch memory")
    for k in spill:
        print(f"  {k['.private_segment_fixed_size']:5d} B  vgpr {k.get('.vgpr_count')} agpr {k.get('.agpr_count', 0)}  {k['.name'][:110]}")


if __name__ == "__main__":
    main()
