"""Register / scratch budgets of the hot kernels, read from the code objects inside the built library (no GPU, no recompilation):
DESIGN.md's occupancy statements (K11 three waves per SIMD without spills, K13's backward walk and K10 without scratch, ...) are
claims about exactly these numbers.  The library is a host ELF carrying one AMDGPU ELF per translation unit; each is cut out and its
kernel metadata read with llvm-readelf --notes."""
import os
import re
import struct
import subprocess
import tempfile

import pytest

from radiation_ppo_amd import build

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def _code_objects(blob: bytes):
    pos = 0
    while True:
        i = blob.find(b"\x7fELF", pos)
        if i < 0:
            return
        pos = i + 4
        if struct.unpack_from("<H", blob, i + 18)[0] != 224:             # EM_AMDGPU
            continue
        shoff, = struct.unpack_from("<Q", blob, i + 40)
        shentsize, shnum = struct.unpack_from("<HH", blob, i + 58)
        yield blob[i:i + shoff + shentsize * shnum]


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(READELF):
        pytest.skip("llvm-readelf not available")
    lib = build.build(verbose=False)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for n, co in enumerate(_code_objects(open(lib, "rb").read())):
            path = os.path.join(tmp, f"co{n}.elf")
            with open(path, "wb") as f:
                f.write(co)
            notes = subprocess.run([READELF, "--notes", path], capture_output=True, text=True, check=True).stdout
            for block in notes.split("- .agpr_count:")[1:]:
                name = re.search(r"\.name:\s+(\S+)", block).group(1)
                val = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", block).group(1))
                out[name] = dict(vgpr=val("vgpr_count"), scratch=val("private_segment_fixed_size"), lds=val("group_segment_fixed_size"),
                                 vgpr_spill=val("vgpr_spill_count"))
    assert len(out) > 40
    return out


def _find(kernels, *parts):
    hits = [k for k in kernels if all(p in k for p in parts)]
    assert len(hits) == 1, (parts, hits)
    return kernels[hits[0]]


@pytest.mark.parametrize("parts,max_vgpr", [
    (("rs_pfgru_kernelILb0ELi1E",), 168),            # K11 collector step: three waves per SIMD (512 / 3, granule 8)
    (("rs_pfgru_kernelILb0ELi4E",), 168),            # K11 pass (four steps per launch)
    (("rs_pfgru_kernelILb1ELi1E",), 168),            # K11 with recorded draws
    (("rs_pfgru_train_fwd_kernelILb1E",), 256),      # K13 forward walk (draws hashed in the kernel): two waves per SIMD
    (("rs_pfgru_train_fwd_kernelILb0E",), 256),      # K13 forward walk (draws read)
    (("rs_pfgru_train_kernel",), 512),               # K13 backward walk: one wave per SIMD
    (("rs_cnn_fwd_kernelILi6E",), 128),              # K9: four waves per SIMD
    (("rs_cnn_fwd_kernelILi4E",), 128),
    (("rs_cnn_bwd_kernelILi6E",), 168),              # K10: three waves per SIMD
    (("rs_cnn_bwd_kernelILi4E",), 168),
    (("rs_gru_fwd_kernel",), 512),
    (("rs_gru_bwd_kernel",), 512),
    (("rs_a2c_heads_kernel",), 512),
])
def test_hot_kernels_fit_their_occupancy_without_scratch(kernels, parts, max_vgpr):
    k = _find(kernels, *parts)
    assert k["scratch"] == 0 and k["vgpr_spill"] == 0, k
    assert k["vgpr"] <= max_vgpr, k


def test_lds_budgets(kernels):
    """workgroups per CU by LDS (160 KB): K13's backward walk 4 (one wave per SIMD), K10 3, K9 2, K11 >= 3"""
    cu = 160 * 1024
    assert cu // _find(kernels, "rs_pfgru_train_kernel")["lds"] == 4
    assert cu // _find(kernels, "rs_pfgru_kernelILb0ELi4E")["lds"] >= 3
    assert cu // _find(kernels, "rs_pfgru_train_fwd_kernelILb1E")["lds"] >= 2
