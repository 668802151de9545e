"""Register budget of the hot kernels (no GPU needed: hipcc's kernel-resource-usage remarks for gfx950, N = 5).

Round 5 found out the hard way that this is a CORRECTNESS gate, not a performance note: an opt-in code path added
inside handle_leader pushed rgb_tick_classes_kernel<5> over its 128-register budget (66 VGPRs spilled to scratch), and
on the MI355X that build returned wrong decisions for ~100 of 285 000 messages per tick -- always lanes 0-15 of a
wavefront, always in the general append_entries_rpc path, a different set on every run -- while the CPU emulation of
the same sources stayed bit-exact (tools/parity_tick0.py, profiles/EXPERIMENTS.md "Round 5").  The kernels place
their own s_waitcnt (LDS-DMA copies, the publish step); scratch traffic inside them is not something that code was
written for.  So: the per-tick class kernel and both train kernels must compile without scratch, and the class kernel
must keep its four wavefronts per SIMD."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc", path="/opt/rocm/bin") or shutil.which("hipcc")


@pytest.fixture(scope="module")
def usage():
    if HIPCC is None:
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "ra_amd", "csrc", "rgb_kernels.hip")
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-mllvm",
                        "-disable-machine-licm", "-DRGB_X_ONLY_N=5", "-Rpass-analysis=kernel-resource-usage", "-c", src,
                        "-o", os.devnull], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1); out[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur:
            out[cur][m.group(1).strip()] = int(m.group(2))
    return out


def _kernel(usage, name):
    hits = [v for k, v in usage.items() if name in k]
    assert len(hits) == 1, f"{name}: {len(hits)} kernels match"
    return hits[0]


@pytest.mark.parametrize("name,max_vgprs,min_occupancy", [("rgb_tick_classes_kernelILi5E", 128, 4),
                                                           ("rgb_train_dealt_kernelILi5E", 168, 3),
                                                           ("rgb_train_kernelILi5E", 168, 3)])
def test_hot_kernels_use_no_scratch(usage, name, max_vgprs, min_occupancy):
    k = _kernel(usage, name)
    assert k["ScratchSize"] == 0 and k["VGPRs Spill"] == 0, f"{name} spills to scratch: {k}"
    assert k["VGPRs"] <= max_vgprs, k
    assert k["Occupancy"] >= min_occupancy, k
