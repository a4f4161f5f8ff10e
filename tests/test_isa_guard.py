"""DESIGN.md §9 as an enforceable rule: the kernels that issue f16 MFMA chains hold no packed-FP32 VALU instruction.

CPU test (hipcc cross-compiles here): the linked library's gfx950 code objects are disassembled and scanned; `build.build()` runs the
same check after every link and refuses the library on a hit."""
import re

from waveflow_amd import build as wf_build
from waveflow_amd import isa_guard


def test_mfma_kernels_hold_no_packed_fp32_instruction():
    lib = wf_build.build()
    hits, n_guarded = isa_guard.scan(lib)
    assert n_guarded >= 40, n_guarded            # every k_mfma / k_etile_* instantiation was looked at
    assert not hits, {k: v[:2] for k, v in hits.items()}


def test_the_scanner_sees_packed_fp32_code_where_it_exists(monkeypatch):
    """The reverse wave sweeps are compiled without the rule and are full of v_pk_fma_f32: pointed at them, the scanner must report it
    (a scanner that finds nothing anywhere would pass the test above for the wrong reason)."""
    monkeypatch.setattr(isa_guard, "GUARDED", re.compile(r"k_wave_bwd"))
    hits, n_guarded = isa_guard.scan(wf_build.build())
    assert n_guarded > 0 and hits and sum(len(v) for v in hits.values()) > 100


def test_mfma_translation_units_are_built_with_the_rule():
    assert "-fno-slp-vectorize" in wf_build.MFMA_FLAGS and "-packed-fp32-ops" in wf_build.MFMA_FLAGS


def test_every_translation_unit_is_built_with_preallocated_sgpr_spill_vgprs():
    """DESIGN.md section 9 (round 3): SGPRs spilled into lazily allocated VGPR lanes lost values at random in k_sample<2, 64>; the flag that ends it
    belongs to the flags of every translation unit, and the recorded flags of the built objects say so."""
    import os
    flags = wf_build.FLAGS
    assert "-amdgpu-prealloc-sgpr-spill-vgprs" in flags and flags[flags.index("-amdgpu-prealloc-sgpr-spill-vgprs") - 1] == "-mllvm"
    wf_build.build()
    assert os.path.exists(wf_build.STAMP) and "-amdgpu-prealloc-sgpr-spill-vgprs" in open(wf_build.STAMP).read()
