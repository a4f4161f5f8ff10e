"""Builds waveflow_amd/libwaveflow_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libwaveflow_hip.so")
OBJ = os.path.join(CSRC, "_obj")

SOURCES = ["wf_tables.cpp", "wf_model.cpp", "wf_kernels_scalar.hip", "wf_scalar_inst_d2.hip", "wf_scalar_inst_d3.hip", "wf_scalar_inst_d4.hip", "wf_scalar_inst_d56.hip",
           "wf_scalar_inst_d78.hip", "wf_scalar_inst_n64.hip", "wf_kernels_mfma.hip", "wf_mfma_inst_d2.hip", "wf_mfma_inst_d2t2.hip", "wf_mfma_inst_d34.hip",
           "wf_mfma_inst_d567.hip", "wf_mfma_inst_d8.hip", "wf_mfma_inst_k2.hip", "wf_kernels_rqs.hip", "wf_kernels_grad.hip", "wf_kernels_wave.hip", "wf_kernels_etile.hip", "wf_etile_bwd_k2.hip", "wf_kernels_etile_dir.hip"]
# -ffp-contract=off: the index arithmetic and the table lerp keep the reference's separate
# multiply / add roundings; dot products that may fuse say so with explicit fmaf / MFMA.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]
# SGPRs that do not fit are spilled into the lanes of VGPRs (v_writelane_b32 / v_readlane_b32).  With the compiler's default, lazily allocated spill
# VGPRs the one-lane-per-walker sampler of 33 .. 64 bases (k_sample<2, 64>: 427 such spills, a divergent rejection loop) lost values at random --
# NaN draws for 2 - 40 % of the walkers, other ones from run to run, the code object byte-identical to a build that had worked.
# Pre-allocated spill VGPRs end it; every translation unit is built this way (scratch/sgpr_spill_scan.py lists the kernels that spill).
FLAGS += ["-mllvm", "-amdgpu-prealloc-sgpr-spill-vgprs"]
FLAGS += os.environ.get("WF_CXXFLAGS", "").split()  # experiment switches (-DWF_...)
# No packed-FP32 VALU code (v_pk_fma_f32 ...) in the translation units of the MFMA kernels: wf_mfma_impl.h, DESIGN.md §9.  The SLP
# vectorizer is one source of it, instruction selection of two-element float vectors another (the box transform's differences came out as
# v_pk_add_f32 in the D >= 3 builds): the target feature is switched off for these units (the host pass does not know the feature and says
# so on stderr; harmless), and isa_guard.check() disassembles the linked library and refuses it if one is left.
MFMA_FLAGS = ["-fno-slp-vectorize", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
# per translation unit: the two-row-block reverse kernels under the max-ilp scheduling strategy (-6 % for them, +1 % for the one-row-block form)
EXTRA_FLAGS = {"wf_etile_bwd_k2.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _deps(src):
    d = [os.path.join(CSRC, src), os.path.join(CSRC, "wf_internal.h"), os.path.join(HERE, "..", "include", "waveflow_hip.h")]
    if "mfma" in src or "etile" in src:
        d.append(os.path.join(CSRC, "wf_mfma_impl.h"))
    if "etile" in src:
        d += [os.path.join(CSRC, "wf_etile_common.h"), os.path.join(CSRC, "wf_etile_adjoint.h")]
    if src == "wf_etile_bwd_k2.hip":
        d.append(os.path.join(CSRC, "wf_kernels_etile.hip"))
    if "grad" in src or "wave" in src:
        d.append(os.path.join(CSRC, "wf_ring.h"))
    if "scalar" in src or "wave" in src or "rqs" in src:   # (the wave sampler shares Philox and the box reverse with the one-lane kernels)
        d.append(os.path.join(CSRC, "wf_scalar_impl.h"))
    return [p for p in d if os.path.exists(p)]


STAMP = os.path.join(HERE, "libwaveflow_hip.flags")   # next to the library (it travels with it; csrc/_obj does not)


def _flags_text():
    return " ".join(FLAGS) + " | " + " ".join(MFMA_FLAGS) + " | " + " ".join(f"{k}: {' '.join(v)}" for k, v in sorted(EXTRA_FLAGS.items()))


def _library_current():
    """The library is newer than every source / header and was built with the current flags."""
    try:
        if open(STAMP).read() != _flags_text():
            return False
        t = os.path.getmtime(LIB)
    except OSError:
        return False
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h"))]
    deps.append(os.path.join(HERE, "..", "include", "waveflow_hip.h"))
    return all(os.path.getmtime(d) <= t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    if not force and _library_current():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    try:
        force = force or open(STAMP).read() != _flags_text()   # changed flags (FLAGS / WF_CXXFLAGS / MFMA_FLAGS): everything again
    except OSError:
        force = True
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    for s in srcs:
        o = os.path.join(OBJ, s + ".o")
        if force or not os.path.exists(o) or any(os.path.getmtime(o) < os.path.getmtime(d) for d in _deps(s)):
            cmd = [_hipcc()] + FLAGS + (MFMA_FLAGS if ("mfma" in s or "etile" in s) else []) + EXTRA_FLAGS.get(s, []) + (["-x", "hip"] if s.endswith(".cpp") else [])
            cmd += ["-c", os.path.join(CSRC, s), "-o", o]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s + ".o") for s in srcs]
    if jobs or force or not os.path.exists(LIB) or any(os.path.getmtime(LIB) < os.path.getmtime(o) for o in objs):
        tmp = LIB + ".tmp"
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs)
        if os.environ.get("WF_SKIP_ISA_GUARD") != "1":   # (scratch builds of deliberately packed variants only)
            import importlib.util   # (by path: this file also runs as a script)
            spec = importlib.util.spec_from_file_location("wf_isa_guard", os.path.join(HERE, "isa_guard.py"))
            isa_guard = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(isa_guard)
            try:
                isa_guard.check(tmp)
            except Exception:
                os.remove(tmp)
                raise
        os.replace(tmp, LIB)
    with open(STAMP, "w") as f:
        f.write(_flags_text())
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
