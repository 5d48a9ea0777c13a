"""Build libmsam_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build() and by hand:
    python -m micro_sam_amd.build
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmsam_hip.so")
SOURCES = ["gemm.hip", "norm.hip", "attention.hip", "decoder.hip", "postprocess.hip", "encoder.hip", "segment.hip", "wsgemm.hip", "declayer.hip", "decfold.hip", "decfold_tok.hip", "upfused.hip", "image.hip", "train.hip", "amgselect.hip", "watershed.hip", "strict.hip"]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "msam_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def variant_path(variant: str = "") -> str:
    """Library file of a build variant: "" = default (fp16 mask decoder), "decbf16" = all-bf16 decoder of round 1
    (-DMSAM_DEC_F16=0; ablation only, selected at run time with MSAM_LIB_VARIANT=decbf16)."""
    return LIB_PATH if not variant else os.path.join(LIB_DIR, f"libmsam_hip_{variant}.so")


def build(force: bool = False, verbose: bool = True, variant: str = "", experiments: bool = False) -> str:
    """``experiments``: -DMSAM_EXPERIMENTS=1 (gemm.hip: the timing knobs and rejected kernel forms tools/gemm_probe.py drives; forces a rebuild,
    and the next default build is forced as well)."""
    os.makedirs(LIB_DIR, exist_ok=True)
    marker = os.path.join(LIB_DIR, ".experiments")
    if experiments or os.path.exists(marker):
        force = True
        if experiments:
            open(marker, "w").close()
        else:
            os.remove(marker)
    if variant:
        return _build_variant(variant, force, verbose)
    if not force and not needs_build():
        return LIB_PATH
    objs = []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(
                os.path.getmtime(path), os.path.getmtime(os.path.join(CSRC, "common.h")),
                os.path.getmtime(os.path.join(HERE, "..", "include", "msam_hip.h"))):
            cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", path, "-o", obj]
            if experiments:
                cmd.insert(-4, "-DMSAM_EXPERIMENTS=1")
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_PATH


def _build_variant(variant: str, force: bool, verbose: bool) -> str:
    if variant != "decbf16":
        raise ValueError(f"unknown build variant {variant!r}")
    out = variant_path(variant)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    objs = []
    for src in SOURCES:
        obj = os.path.join(LIB_DIR, src.replace(".hip", f".{variant}.o"))
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-DMSAM_DEC_F16=0", "-c", os.path.join(CSRC, src),
               "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    subprocess.check_call([_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", out])
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, variant="decbf16" if "--dec-bf16" in sys.argv else "", experiments="--experiments" in sys.argv))
