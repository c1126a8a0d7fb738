"""CPU: the dominant kernel's k-loop must not touch scratch memory.

hipcc allocates the 256-VGPR LDS-DMA tiles globally: an innocent change in the epilogue or the per-tile set-up can make it spill values of the slab
loop, and every reload there waits (vmcnt(0)) for the slab's LDS-DMA pieces too — the tile still passes every numerical test and runs 15 % slower
(DESIGN.md §3.1e).  This cross-compiles the kernel file to gfx950 assembly (≈1 minute, no GPU needed) and lets tools/isa_hot_scratch.py walk the
software-pipelined fp16+8 256x256 tile's slab loop."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="needs hipcc")
def test_dominant_tile_has_no_scratch_access_in_its_slab_loop(tmp_path):
    asm = str(tmp_path / "conv_igemm_dma.s")
    cc = HIPCC if os.path.exists(HIPCC) else shutil.which("hipcc")
    r = subprocess.run([cc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                        os.path.join(ROOT, "marconet_amd", "csrc", "conv_igemm_dma.hip"), "-o", asm], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_hot_scratch.py"), asm], capture_output=True, text=True, timeout=120)
    print(r.stdout)
    assert "scratch accesses on the hot path: 0" in r.stdout and r.returncode == 0, r.stdout + r.stderr
    # round 5: the build of the same tile WITH the GroupNorm-sum block — its own translation unit, compiled with the flag build.sh gives it (without the flag
    # hipcc puts scratch reloads into its slab loop)
    asm2 = str(tmp_path / "conv_dma_swp_gn.s")
    r = subprocess.run([cc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-greedy-reverse-local-assignment=1", "-S", "--cuda-device-only",
                        os.path.join(ROOT, "marconet_amd", "csrc", "conv_dma_swp_gn.hip"), "-o", asm2], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_hot_scratch.py"), asm2], capture_output=True, text=True, timeout=120)
    print(r.stdout)
    assert "ELb1ELb1EEv8ConvArgs" in r.stdout and "scratch accesses on the hot path: 0" in r.stdout and r.returncode == 0, r.stdout + r.stderr
    flag = open(os.path.join(ROOT, "marconet_amd", "csrc", "build.sh")).read()
    assert "conv_dma_swp_gn) echo \"-mllvm -greedy-reverse-local-assignment=1\"" in flag, "build.sh must compile conv_dma_swp_gn.hip with the flag this test checks"


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="needs hipcc")
def test_one_wave_per_simd_tile_isa(tmp_path):
    """round 6, conv_dma_w4.hip (fp16+8 id 16, AUTO for cout >= 256): its MFMAs are `asm volatile` statements with the accumulators in a[0:255] — hipcc neither
    pads their hazards nor may it touch the accumulator file itself.  On the built ISA: 256 AGPRs allocated; no compiler v_accvgpr_write / _mov (a lazy copy of a
    zeroed block in front of an unpadded MFMA corrupted register 0 of every block in round 6); no VALU write of an MFMA operand inside its two wait states
    (tools/isa_mfma_hazards.py); no scratch access between the slab barrier and the loop's back edge outside the once-per-tile blocks — in each of the four builds
    (with / without scale vectors, with / without residual + GroupNorm sums)."""
    import re
    asm = str(tmp_path / "conv_dma_w4.s")
    cc = HIPCC if os.path.exists(HIPCC) else shutil.which("hipcc")
    r = subprocess.run([cc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                        os.path.join(ROOT, "marconet_amd", "csrc", "conv_dma_w4.hip"), "-o", asm], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    s = open(asm).read()
    assert len(re.findall(r"\.agpr_count:\s+256", s)) == 4, "the 16 accumulator blocks of each of the four builds must occupy a[0:255]"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_mfma_hazards.py"), asm], capture_output=True, text=True, timeout=120)
    print(r.stdout)
    assert r.returncode == 0 and r.stdout.count("0 finding(s)") == 4, r.stdout + r.stderr
