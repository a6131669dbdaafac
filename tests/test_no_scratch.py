"""No kernel of libpfk may use scratch (private segment) memory.

Scratch is not free "spill space": every scratch store is an HBM write.  Round 2 found a 16-byte slice of the kernel arguments
parked in scratch by hipcc in every LINEAR convolution kernel — PMC `WRITE_SIZE` read 1.25x the algorithmic output bytes on six
launches per GRU iteration until it was removed (profiles/r02_b).  This test reads the kernel descriptors' metadata out of the
built library (the gfx950 code objects inside `.hip_fatbin`) and fails on any `.private_segment_fixed_size` > 0, so the next
such regression is caught at build time, on CPU."""
import os
import re
import struct
import subprocess
import tempfile

import pytest

LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ptlflow_amd", "libpfk.so")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(fat: bytes):
    """gfx950 ELF images of every clang offload bundle in a .hip_fatbin section."""
    for m in re.finditer(MAGIC, fat):
        base = m.start()
        (n,) = struct.unpack_from("<Q", fat, base + len(MAGIC))
        pos = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", fat, pos)
            triple = fat[pos + 24:pos + 24 + tlen].decode()
            pos += 24 + tlen
            if "gfx950" in triple and size > 0:
                yield fat[base + off:base + off + size]


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(f"{LLVM}/llvm-objcopy") and os.path.exists(f"{LLVM}/llvm-readelf")),
                    reason="needs the built libpfk.so and the ROCm llvm tools")
def test_no_kernel_uses_scratch():
    with tempfile.TemporaryDirectory() as tmp:
        fatbin = os.path.join(tmp, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", LIB, fatbin], check=True)
        fat = open(fatbin, "rb").read()
        kernels, offenders = 0, []
        for i, elf in enumerate(_code_objects(fat)):
            path = os.path.join(tmp, f"co{i}.elf")
            open(path, "wb").write(elf)
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", path], check=True, capture_output=True, text=True).stdout
            for block in notes.split("- .agpr_count:")[1:]:
                name = re.search(r"\.name:\s+(\S+)", block)
                scratch = re.search(r"\.private_segment_fixed_size:\s+(\d+)", block)
                spills = re.search(r"\.vgpr_spill_count:\s+(\d+)", block)
                assert name and scratch, "kernel metadata layout changed"
                kernels += 1
                if int(scratch.group(1)) > 0 or (spills and int(spills.group(1)) > 0):
                    offenders.append((name.group(1), int(scratch.group(1))))
        assert kernels >= 60, f"only {kernels} kernels found: the extraction is broken"
        assert not offenders, f"kernels with scratch / VGPR spills: {offenders}"
