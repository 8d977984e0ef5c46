"""Occupancy audit of the hot kernels (no GPU needed: hipcc cross-compiles gfx950).

Round 6 found a forward substitution held to ONE workgroup per CU by a loop "off the critical path" that the compiler had
unrolled to 255 registers, and a fused solve that spilled inside its refinement loop.  The compiler's own resource remarks
(-Rpass-analysis=kernel-resource-usage) are parsed here and checked against the occupancy each kernel is designed for
(the register count of a kernel is set by its greediest phase, wherever that phase sits)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "clarabel.rs_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-munsafe-fp-atomics",
         "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c"]

# kernel (substring of the mangled name) -> (file, fewest waves per SIMD it must reach, most scratch bytes per lane)
DESIGNED = {
    "11k_snode_triILb1ELi1E": ("snode.hip", 6, 0),   # forward substitution, one / two vectors: three workgroups of 8 waves per CU
    "11k_snode_triILb1ELi2E": ("snode.hip", 6, 0),
    "11k_snode_triILb0ELi1E": ("snode.hip", 5, 0),   # backward
    "11k_snode_triILb0ELi2E": ("snode.hip", 5, 0),
    "14k_snode_updateE": ("snode.hip", 4, 64),       # two workgroups of 8 waves per CU; spills only outside the tile loop
    "14k_snode_extendE": ("snode.hip", 4, 64),
    "19k_snode_extend_wideILi4E": ("snode.hip", 2, 0),  # two workgroups of 4 waves per CU, 128 accumulator registers
    "12k_bundle_irsILi256ELi4E": ("bundle_ir.hip", 4, 96),  # 1000 co-resident workgroups of 256 threads: four per CU
    "11k_dblk_symvILi1E": ("algebra.hip", 4, 0),
    "11k_dblk_symvILi2E": ("algebra.hip", 4, 0),     # two vectors: still two workgroups of 8 waves per CU
}


def _resources(src):
    out = subprocess.run([HIPCC] + FLAGS + [os.path.join(CSRC, src), "-o", os.devnull], cwd=CSRC, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            res[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur:
            res[cur][m.group(1).strip()] = int(m.group(2))
    return res


@pytest.mark.skipif(not os.path.exists(HIPCC) or shutil.which("make") is None, reason="no hipcc")
def test_hot_kernels_reach_their_designed_occupancy():
    by_file = {}
    for key, (src, _, _) in DESIGNED.items():
        by_file.setdefault(src, []).append(key)
    for src, keys in by_file.items():
        res = _resources(src)
        for key in keys:
            names = [n for n in res if key in n]
            assert len(names) == 1, (key, names)
            r = res[names[0]]
            _, occ_min, scratch_max = DESIGNED[key]
            assert r["Occupancy"] >= occ_min, (key, r)
            assert r["ScratchSize"] <= scratch_max, (key, r)
