"""Build-time guards for DESIGN.md §9.  Rule 1: no packed-FP32 VALU instruction may appear in a kernel that issues f16 MFMA chains.  Rule 2 (round 4, below):
no read of an MFMA result inside the MFMA's wait-state window behind a branch.

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
GUARDED = re.compile(r"k_mfma|k_etile|k_efused|k_ebwd|k_edir")                      # kernels of the translation units built with MFMA_FLAGS
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


# ---- second rule (round 4): no vector read of an MFMA result inside the MFMA's wait-state window ACROSS A BRANCH.
# hipcc counts the wait states between an MFMA and a VALU / memory instruction reading its destination in straight-line code; behind the wait loop of
# k_ebwd<true, 2> (a branch around a spin loop, the join right behind it) it let v_accvgpr_read follow the last MFMA of a product by 4 instructions where
# 11 are due: when the branch was taken straight away the last rows of the product were read before the matrix pipe had written them (DESIGN.md 9,
# "Round 4").  The rule walks every path of at most `need` instructions from each MFMA through the control flow of the disassembly and reports a non-MFMA
# instruction that reads the destination registers sooner than `need` wait states later IF the path crossed a branch or a branch target (straight-line
# code is the compiler's own count and is not second-guessed).  Every instruction counts as one wait state, s_nop N as N + 1.
_INS = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_TGT = re.compile(r"<[^>]*\+0x([0-9a-fA-F]+)>\s*$")
_REG = re.compile(r"\b([av])(?:\[(\d+):(\d+)\]|(\d+))")


def _regs(text):
    out = set()
    for m in _REG.finditer(text):
        lo, hi = (int(m.group(2)), int(m.group(3))) if m.group(2) is not None else (int(m.group(4)), int(m.group(4)))
        out.update((m.group(1), r) for r in range(lo, hi + 1))
    return out


def _mfma_need(mn):
    if "32x32x16" in mn or "16x16x32" in mn:      # 8 (4) passes
        return 11
    return 19                                      # 16 passes (v_mfma_f32_32x32x2_f32 and anything unlisted: the longest window)


def join_hits_in_text(text, guarded=None):
    """The rule on llvm-objdump -d output: -> {kernel: [descriptions]} (guarded: regex of the kernels to look at, default GUARDED)."""
    guarded = guarded or GUARDED
    hits, kernels, cur = {}, {}, None
    for line in text.splitlines():
        m = _SYM.match(line)
        if m:
            cur = m.group(1) if guarded.search(m.group(1)) else None
            if cur:
                kernels[cur] = {"base": int(line.split()[0], 16), "ins": []}
            continue
        if cur is None:
            continue
        m = _INS.match(line)
        if m:
            kernels[cur]["ins"].append((int(m.group(3), 16), m.group(1), m.group(2), line))
    for name, k in kernels.items():
        ins = k["ins"]
        index = {a: i for i, (a, _, _, _) in enumerate(ins)}
        target, is_target = {}, set()
        for i, (a, mn, ops, line) in enumerate(ins):
            if mn.startswith(("s_cbranch", "s_branch")):
                t = _TGT.search(line)
                if t and k["base"] + int(t.group(1), 16) in index:
                    target[i] = index[k["base"] + int(t.group(1), 16)]
                    is_target.add(target[i])
        for i, (a, mn, ops, line) in enumerate(ins):
            if not mn.startswith("v_mfma"):
                continue
            need = _mfma_need(mn)
            dst = _regs(ops.split(",")[0])
            # depth-first over paths: (instruction index, wait states so far, crossed a branch / join)
            stack, seen = [(i + 1, 0, False)], set()
            while stack:
                j, ws, crossed = stack.pop()
                if j >= len(ins) or ws >= need or (j, crossed) in seen:
                    continue
                seen.add((j, crossed))
                a2, mn2, ops2, line2 = ins[j]
                crossed2 = crossed or j in is_target
                parts = ops2.split(",")
                is_store = mn2.startswith(("ds_write", "ds_store", "global_store", "scratch_store", "buffer_store", "flat_store"))
                srcs = _regs(ops2 if is_store else ",".join(parts[1:]))
                if not mn2.startswith("v_mfma") and (mn2.startswith("v_") or is_store) and srcs & dst and crossed2:
                    hits.setdefault(name, []).append(f"{mn} at {a:#x} -> {mn2} {ops2} at {a2:#x} after {ws} of {need} wait states")
                    continue
                if _regs(parts[0]) >= dst and not mn2.startswith("v_mfma"):
                    continue            # destination overwritten by something else: the window is that instruction's business
                step = 1
                if mn2 == "s_nop":
                    step = int(ops2.strip() or 0) + 1
                if mn2 == "s_endpgm":
                    continue
                if j in target:
                    stack.append((target[j], ws + step, True))
                    if mn2.startswith("s_branch"):
                        continue
                stack.append((j + 1, ws + step, crossed2))
    return hits


def scan_mfma_joins(lib):
    """-> {kernel: [description of a read of an MFMA destination inside its window behind a branch]} over the guarded kernels of `lib`."""
    od = _objdump()
    hits = {}
    with tempfile.TemporaryDirectory(prefix="wf_isa_") as tmp:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([od, "--offloading", local], cwd=tmp, check=True, capture_output=True)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            syms = subprocess.run([od, "-t", os.path.join(tmp, f)], capture_output=True, text=True, check=True).stdout
            if not GUARDED.search(syms):
                continue
            hits.update(join_hits_in_text(subprocess.run([od, "-d", os.path.join(tmp, f)], capture_output=True, text=True, check=True).stdout))
    return hits


def unguarded_mfma_kernels(lib):
    """Kernels of `lib` that issue MFMA instructions but are not matched by GUARDED (the rules above would not see them)."""
    od = _objdump()
    out = set()
    with tempfile.TemporaryDirectory(prefix="wf_isa_") as tmp:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([od, "--offloading", local], cwd=tmp, check=True, capture_output=True)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            cur = None
            for line in subprocess.run([od, "-d", "--no-show-raw-insn", os.path.join(tmp, f)], capture_output=True, text=True, check=True).stdout.splitlines():
                m = _SYM.match(line)
                if m:
                    cur = m.group(1)
                elif cur and "v_mfma" in line and not GUARDED.search(cur):
                    out.add(cur)
    return sorted(out)


def check(lib):
    joins = scan_mfma_joins(lib)
    if joins:
        lines = [f"  {k}: {len(v)} site(s), e.g. {v[0]}" for k, v in sorted(joins.items())]
        raise RuntimeError("isa_guard: MFMA result read inside its wait-state window behind a branch (DESIGN.md §9, round 4):\n" + "\n".join(lines))
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
