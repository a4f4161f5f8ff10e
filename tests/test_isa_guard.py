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


def test_no_mfma_result_is_read_inside_its_window_behind_a_branch():
    """behind the wait loop of k_ebwd<., 2> hipcc let a vector read of an MFMA result follow the MFMA by 4 instructions
    at a join (11 wait states are due): run-to-run differences in one gradient block.  The shipped library has no such site in any MFMA kernel."""
    hits = isa_guard.scan_mfma_joins(wf_build.build())
    assert not hits, {k: v[:2] for k, v in hits.items()}


_JOIN_BUG = """
0000000000001000 <k_ebwd_demo>:
\tv_mfma_f32_32x32x16_f16 a[0:15], v[218:221], v[8:11], a[0:15]   // 000000001000: D3D40000
\ts_and_saveexec_b64 s[0:1], vcc                                  // 000000001008: BE80206A
\ts_cbranch_execz 7                                               // 00000000100C: BF880007 <k_ebwd_demo+0x2c>
\ts_mov_b64 s[50:51], 0                                           // 000000001010: BEB20180
\ts_sleep 1                                                       // 000000001014: BF8E0001
\tds_read_b32 v1, v255 offset:584                                 // 000000001018: D86C0248
\ts_waitcnt lgkmcnt(0)                                            // 000000001020: BF8CC07F
\ts_cbranch_execnz 65532                                          // 000000001024: BF89FFFC <k_ebwd_demo+0x14>
\ts_nop 0                                                         // 000000001028: BF800000
\ts_or_b64 exec, exec, s[0:1]                                     // 00000000102C: 87FE007E
\tv_mfma_f32_32x32x16_f16 a[28:43], v[162:165], v[242:245], 0     // 000000001030: D3D4001C
\tv_accvgpr_read_b32 v145, a15                                    // 000000001038: D3D84091
\ts_endpgm                                                        // 000000001040: BF810000
"""


def test_the_join_rule_sees_the_sequence_that_failed_and_accepts_the_padded_one():
    guarded = re.compile("k_ebwd_demo")
    hits = isa_guard.join_hits_in_text(_JOIN_BUG, guarded)
    assert list(hits) == ["k_ebwd_demo"] and "v_accvgpr_read_b32 v145, a15" in hits["k_ebwd_demo"][0] and "of 11 wait states" in hits["k_ebwd_demo"][0]
    # the same code with the pipe drained in front of the branch (acc_add's fix): clean
    padded = _JOIN_BUG.replace("\ts_and_saveexec_b64 s[0:1], vcc ", "\ts_nop 15                                                       // 000000001004: BF80000F\n\ts_and_saveexec_b64 s[0:1], vcc ")
    assert isa_guard.join_hits_in_text(padded, guarded) == {}
    # straight-line code is the compiler's own count: not second-guessed
    straight = "\n".join(l for l in _JOIN_BUG.splitlines() if "s_cbranch" not in l)
    assert isa_guard.join_hits_in_text(straight, guarded) == {}


def test_every_kernel_that_issues_mfma_is_guarded():
    """A new matrix-core kernel under a name the guard's pattern does not match would escape both rules."""
    assert isa_guard.unguarded_mfma_kernels(wf_build.build()) == []
