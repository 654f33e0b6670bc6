"""The build records registers / scratch / occupancy per kernel (csrc/build.sh: hipcc's kernel-resource-usage remarks ->
csrc/build/resources.txt). The GEMM and weight-gradient kernels must not spill: round 4 measured what 116 bytes of scratch do to
igemm8<3, relu> (the generator's u2: 77.5 -> 101 us) after an epilogue change that every test passed."""
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RES = os.path.join(ROOT, "gan-inpainting_amd", "csrc", "build", "resources.txt")
NO_SCRATCH = ("igemm3_kernel", "igemm5_kernel", "igemm6_kernel", "igemm7_kernel", "igemm8_kernel", "wgrad3_kernel", "wgrad2_kernel",
              "c1_gather_mfma_kernel", "c1_scatter_fused_kernel", "c1_wgrad_mfma_kernel", "bn_apply_kernel", "act_bn_bwd")


def test_gemm_kernels_have_no_scratch():
    if not os.path.exists(RES):
        pytest.skip("no build record (the library was not built in this tree)")
    rows = [line.rstrip("\n").split("\t") for line in open(RES)]
    assert rows, "empty build record"
    seen = set()
    bad = []
    for src, name, *kv in rows:
        d = dict(x.split("=", 1) for x in kv)
        for k in NO_SCRATCH:
            if k in name:
                seen.add(k)
                if d.get("scratch") not in ("0", None):
                    bad.append((src, name, d["scratch"]))
    assert not bad, f"kernels with scratch: {bad}"
    assert {"igemm8_kernel", "igemm7_kernel", "wgrad3_kernel"} <= seen, f"expected kernels missing from the record: {sorted(seen)}"
