"""No kernel of the library may need a private (scratch) segment beyond a few dwords: a kernel whose registers spill
-- an unrolled staging loop is enough -- makes every launch wait for scratch memory (the round-3 build of k_df_cuts that
did so made the GPU test suite twenty times slower).  Read off the code objects inside the built library."""
import os
import re
import struct
import subprocess
import tempfile

import pytest

from conftest import ROOT

LIB = os.path.join(ROOT, "rust-compression_amd", "libbz2_mi355x.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
# kernels known to keep a few dwords on the stack (bytes); everything else must be 0
ALLOWED = {"k_rle_cuts": 64, "k_phase_b_local": 16, "k_surv_local": 16, "k_group_refine": 16, "k_huffman": 16}


def code_objects(blob):
    pos = 0
    while True:
        at = blob.find(MAGIC, pos)
        if at < 0:
            return
        (count,) = struct.unpack_from("<Q", blob, at + len(MAGIC))
        off = at + len(MAGIC) + 8
        for _ in range(count):
            o, size, tlen = struct.unpack_from("<QQQ", blob, off)
            triple = blob[off + 24:off + 24 + tlen].decode()
            off += 24 + tlen
            if "gfx950" in triple and size:
                yield blob[at + o:at + o + size]
        pos = at + len(MAGIC)


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(READELF)), reason="needs the built library and llvm-readelf")
def test_no_kernel_spills_to_scratch():
    blob = open(LIB, "rb").read()
    seen, bad = 0, []
    for co in code_objects(blob):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for m in re.finditer(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+)", notes, re.S):
            name, size = m.group(1), int(m.group(2))
            seen += 1
            short = re.search(r"(k_[a-z0-9_]+)", name)
            limit = ALLOWED.get(short.group(1) if short else name, 0)
            if size > limit:
                bad.append((name, size))
    assert seen > 50, "no kernel metadata found in the library (%d)" % seen
    assert not bad, "kernels with a private segment: %r" % bad
