"""The commit pipeline depends on its three stages FITTING on a SIMD together: the
persistent accumulate kernel holds 2 waves x 168 VGPRs, the reduce-stage kernels are compiled for the
168 VGPRs that are left, and the prep kernels must be small.  A register-allocation change in any of them
silently turns the pipeline back into three serial stages, so the budget is checked on the built code
object (no GPU needed: the numbers are in the ELF notes)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "kzg_snark_amd", "lib", "libkzg_mi355x.so")


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(LIB):
        sys.path.insert(0, ROOT)
        from kzg_snark_amd import build
        build.build(verbose=False)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py"), LIB],
                         capture_output=True, text=True, check=True).stdout
    rows = []
    for line in out.splitlines():
        m = re.match(r"(\S.*?)\s+vgpr\s+(\d+)\s+agpr\s+(\d+)\s+sgpr\s+(\d+)\s+lds\s+(\d+)\s+scratch\s+(\d+)", line)
        if m:
            rows.append((m.group(1).strip(), int(m.group(2)), int(m.group(5)), int(m.group(6))))
    assert rows, out
    return rows


def rows_of(kernels, name):
    r = [k for k in kernels if name in k[0]]
    assert r, name
    return r


def test_accumulate_leaves_room_for_the_other_stages(kernels):
    for _, vgpr, lds, scratch in rows_of(kernels, "msm_accumulate_kernel"):
        assert vgpr <= 168                 # 2 waves x 168 of the 512 VGPRs of a SIMD
        assert lds <= 28 * 1024            # 4 workgroups per CU: <= 112 KiB of the 160
        assert scratch <= 128              # a few spilled dwords at most; more means the cap bites


def test_reduce_stage_fits_beside_two_accumulate_waves(kernels):
    for name in ("msm_finalize_kernel", "msm_finalize_heavy_kernel", "msm_rc1_kernel", "msm_rc2_kernel",
                 "msm_planes_kernel"):
        for _, vgpr, lds, _ in rows_of(kernels, name):
            assert vgpr <= 168 and lds <= 1024, name


def test_prep_kernels_are_small(kernels):
    prep = [k for k in kernels if k[0].startswith("prep_") or "prep_" in k[0]]
    assert len(prep) >= 10
    for name, vgpr, lds, scratch in prep:
        # the bin sort stages a whole bin (28 KiB of table indices): ONE of its workgroups fits the 46 KiB the four
        # accumulate workgroups of a CU leave; the partition-1 histograms hold 2048 bins (8 KiB)
        # (and its threads keep the bin's entries in registers between the two passes: 2 x 28 VGPRs; two accumulate
        # waves leave 512 - 2 x 153 registers of a SIMD, so such a wave still fits twice)
        # the partition-1 histograms of the variants for 2^21 / 2^22 scalars hold 4096 / 8192 bins (16 / 32 KiB)
        cap, regs = (31 * 1024, 96) if "binsort" in name else \
                    (33 * 1024, 64) if ("prep_count1" in name or "prep_scatter1" in name) else (9 * 1024, 64)
        assert vgpr <= regs and lds <= cap and scratch == 0, name


def test_opening_tiles_fit_beside_the_accumulate_kernel(kernels):
    """While an accumulate kernel is in flight the opening takes 1024-coefficient tiles (128 fill threads / 256
    combine threads): the fill's LDS stage must fit the 46 KiB four accumulate workgroups leave on a CU, and a wave of
    either kernel the 192 VGPRs two accumulate waves leave on a SIMD.  The 2048-coefficient tiles (alone on the GPU)
    must leave room for two workgroups per CU."""
    fill = sorted(rows_of(kernels, "tile_fill_kernel"), key=lambda k: k[2])
    small, big = [k for k in fill if k[2] < 40 * 1024], [k for k in fill if k[2] >= 40 * 1024]
    assert small and big
    for _, vgpr, lds, scratch in small:
        assert lds <= 46 * 1024 and vgpr <= 128 and scratch == 0
    for _, vgpr, lds, scratch in big:
        assert lds <= 80 * 1024 and vgpr <= 128 and scratch == 0
    for _, vgpr, lds, scratch in rows_of(kernels, "tile_combine_kernel"):
        assert vgpr <= 128 and lds <= 16 * 1024 and scratch == 0      # four waves per SIMD: the kernel waits for memory
