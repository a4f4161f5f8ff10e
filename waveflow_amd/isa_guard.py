"""Build-time guard for DESIGN.md §9: no packed-FP32 VALU instruction may appear in a kernel that issues f16 MFMA chains.

The round-1 corruption of `k_mfma` needed `v_pk_fma_f32` / `v_pk_add_f32` / `v_pk_mul_f32` in the kernel (hipcc's SLP vectorizer, or
its instruction selection for two-element float vectors) together with VALU work scheduled into an MFMA chain at >= 3 waves per SIMD.
`build.py` compiles those translation units with `-fno-slp-vectorize` and with the target feature `packed-fp32-ops` switched off; this
module checks the result: it extracts the gfx950 code objects from the linked library (`llvm-objdump --offloading`), disassembles them
and counts the three opcodes inside every function whose name matches GUARDED.  `build.build()` calls `check()` after every link and
refuses the library on a hit; `tests/test_isa_guard.py` runs the same check in the CPU suite.
"""
import os
import re
import shutil
import subprocess
import tempfile

LLVM_BIN = os.environ.get("WF_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
GUARDED = re.compile(r"k_mfma|k_etile|k_efused|k_ebwd|k_ewgrad")                      # kernels of the translation units built with MFMA_FLAGS
FORBIDDEN = re.compile(r"\bv_pk_(fma|add|mul)_f32\b")
_SYM = re.compile(r"^[0-9a-f]+ <(.+)>:\s*$")


def _objdump():
    p = os.path.join(LLVM_BIN, "llvm-objdump")
    return p if os.path.exists(p) else shutil.which("llvm-objdump")


def scan(lib):
    """-> {kernel symbol: [offending disassembly lines]} over the guarded kernels of `lib`, and the number of guarded kernels seen."""
    od = _objdump()
    if od is None:
        raise RuntimeError("llvm-objdump not found (set WF_LLVM_BIN)")
    hits, n_guarded = {}, 0
    with tempfile.TemporaryDirectory(prefix="wf_isa_") as tmp:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([od, "--offloading", local], cwd=tmp, check=True, capture_output=True)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            # only code objects that define a guarded kernel are disassembled (the symbol table is cheap, the disassembly is not)
            syms = subprocess.run([od, "-t", os.path.join(tmp, f)], capture_output=True, text=True, check=True).stdout
            if not GUARDED.search(syms):
                continue
            text = subprocess.run([od, "-d", os.path.join(tmp, f)], capture_output=True, text=True, check=True).stdout
            cur, guarded = None, False
            for line in text.splitlines():
                m = _SYM.match(line)
                if m:
                    cur = m.group(1)
                    guarded = bool(GUARDED.search(cur))
                    n_guarded += guarded
                elif guarded and FORBIDDEN.search(line):
                    hits.setdefault(cur, []).append(line.strip())
    return hits, n_guarded


def check(lib):
    hits, n = scan(lib)
    if n == 0:
        raise RuntimeError(f"isa_guard: no guarded kernel found in {lib} (pattern {GUARDED.pattern})")
    if hits:
        lines = [f"  {k}: {len(v)} packed-FP32 instruction(s), e.g. {v[0]}" for k, v in sorted(hits.items())]
        raise RuntimeError("isa_guard: packed-FP32 VALU code in MFMA kernels (DESIGN.md §9):\n" + "\n".join(lines))
    return n


if __name__ == "__main__":
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    print(check(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "libwaveflow_hip.so")), "guarded kernels clean")
