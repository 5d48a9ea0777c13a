#!/usr/bin/env python3
"""up_fused_kernel laboratory: named VARIANTS of micro_sam_amd/csrc/upfused.hip as small text patches of the shipped source, each built as
its own shared object (the library itself is untouched) and timed on one GPU against the shipped kernel at the benchmark's launch
(1024 prompts, 3 masks, blocked stream).

    python tools/uf_lab.py                      # on the GPU box: builds every variant with hipcc, times it, compares results with the base
    python tools/uf_lab.py --list               # anywhere: names, kinds and what each variant is for
    python tools/uf_lab.py --only base,R_permlane --prompts 256

Two kinds of variants (profiles/r03_experiments.md section 10 is why they exist: the tile loop's MFMA and VALU issue times are ~30 % of the
measured time each, so the first thing to learn is what the waves WAIT for):

  timing   the kernel with one ingredient taken out (a barrier, the LayerNorm exchange, the GELUs, the output path ...).  Results are
           WRONG by construction; the time difference to the base is that ingredient's share of the tile time.
  exact /  candidates that compute the same function: `exact` must reproduce the base bit for bit, `close` within the decoder's parity
  close    tolerance.  Both are checked on the CPU as well (tests/test_uf_lab_variants_host.py runs the patched source behind the host shim).

Output: a table on stdout and gpurun_out/uf_lab.json (ms per launch, difference to the base, VGPRs / occupancy as the compiler reports
them - a timing variant that needs fewer registers may run at a higher occupancy, the table says so)."""
import argparse
import collections
import ctypes
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "micro_sam_amd", "csrc", "upfused.hip")

_HELPER_ANCHOR = "// erf-GELU of two values in PACKED fp16 arithmetic (G16 instantiation, fp16 decoder build only)."
_ROWS_SUM = '''// sum over the wave's four 16-lane rows, in every lane, without the LDS crossbar: v_permlane16_swap exchanges the odd rows of its first
// operand with the even rows of its second (rows r0 r1 r2 r3 -> (r0 r0 r2 r2), (r1 r1 r3 r3)), v_permlane32_swap the same for 32-lane
// halves; the additions are the ones the xor-16 / xor-32 exchange performs ((r0 + r1) + (r2 + r3)), so the result is the same bits.
MSAM_DEVINL float wave_rows_sum(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float t = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
'''
_SUM_S = "        s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);\n"
_SUM_SS = "        ss += __shfl_xor(ss, 16); ss += __shfl_xor(ss, 32);\n"
_PERMLANE = [(_HELPER_ANCHOR, _ROWS_SUM + _HELPER_ANCHOR), (_SUM_S, "        s = wave_rows_sum(s);\n"),
             (_SUM_SS, "        ss = wave_rows_sum(ss);\n")]

_LN_TWO_PASS = '''        float s = 0.f;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) s += (u[rt][0] + u[rt][1]) + (u[rt][2] + u[rt][3]);
        s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
        const float mean = s * (1.f / 64.f);
        float ss = 0.f;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { u[rt][r] -= mean; ss += u[rt][r] * u[rt][r]; }
        ss += __shfl_xor(ss, 16); ss += __shfl_xor(ss, 32);
        const float rstd = rsqrtf(ss * (1.f / 64.f) + a.eps);
'''
# both sums formed together (their two exchanges are independent: one latency instead of two in a row), variance = E[u^2] - mean^2,
# the centring folded into the normalisation (one fma per value instead of a subtraction and a multiplication)
_LN_ONE_PASS = '''        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            s += (u[rt][0] + u[rt][1]) + (u[rt][2] + u[rt][3]);
#pragma unroll
            for (int r = 0; r < 4; ++r) ss += u[rt][r] * u[rt][r];
        }
        s = wave_rows_sum(s); ss = wave_rows_sum(ss);
        const float mean = s * (1.f / 64.f);
        const float rstd = rsqrtf(fmaxf(ss * (1.f / 64.f) - mean * mean, 0.f) + a.eps);
        const float shift = -mean * rstd;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) u[rt][r] = u[rt][r] * rstd + shift;
'''
_AFFINE = [("gelu_pk_h<G16>(u[rt][0] * rstd * g4.x + b4.x, u[rt][1] * rstd * g4.y + b4.y)", "gelu_pk_h<G16>(u[rt][0] * g4.x + b4.x, u[rt][1] * g4.y + b4.y)"),
           ("gelu_pk_h<G16>(u[rt][2] * rstd * g4.z + b4.z, u[rt][3] * rstd * g4.w + b4.w)", "gelu_pk_h<G16>(u[rt][2] * g4.z + b4.z, u[rt][3] * g4.w + b4.w)"),
           ("gelu_erf2(f32x2_t{u[rt][0] * rstd * g4.x + b4.x, u[rt][1] * rstd * g4.y + b4.y})", "gelu_erf2(f32x2_t{u[rt][0] * g4.x + b4.x, u[rt][1] * g4.y + b4.y})"),
           ("gelu_erf2(f32x2_t{u[rt][2] * rstd * g4.z + b4.z, u[rt][3] * rstd * g4.w + b4.w})", "gelu_erf2(f32x2_t{u[rt][2] * g4.z + b4.z, u[rt][3] * g4.w + b4.w})")]

_BARRIER = "        __syncthreads();                                 // next tile staged; output patch of this tile complete\n"
_CVT = "    const h16x2_t x = __builtin_convertvector(xf, h16x2_t);\n"
_STAGE1 = "            for (int rt = 0; rt < 4; ++rt) u[rt] = mfma16d(w1f[rt][ks], kf, u[rt]);\n"
_PATCH_WRITE = "        if (fg == 0) {                                   // rows = masks r, column = token fr\n"
_STORE = "        if (tid < 64 * a.nmask) {\n"
_GRID = "    const int grid = a.nitems < 2 * cus ? a.nitems : 2 * cus;\n"
_KS = "    while (P * ks < 2 * cus && ks < 16) ks *= 2;\n"

V = collections.OrderedDict()
V["base"] = dict(kind="base", doc="the shipped source, unchanged", patches=[])
V["R_permlane"] = dict(kind="exact", doc="LayerNorm2d sums through v_permlane16/32_swap instead of four ds_bpermute round trips", patches=_PERMLANE)
V["R_ln_one_pass"] = dict(kind="close", doc="both LayerNorm sums in one exchange (variance = E[u^2] - mean^2), centring folded into one fma per value; permlane sums",
                          patches=[_PERMLANE[0], (_LN_TWO_PASS, _LN_ONE_PASS)] + _AFFINE)
V["R_pipelined"] = dict(kind="exact", doc="tile q + 1's stage-1 MFMAs under tile q's LayerNorm2d + GELU inside every wave (tools/uf_lab_pipelined.inc); biases from LDS",
                        patches=[_PERMLANE[0], ("template <int UF_PRIO, int G16>\n__global__", "}  // namespace\n\nextern \"C\" int msam_upscale_fused_layout(",
                                                open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "uf_lab_pipelined.inc")).read())])
# v_exp_f16 has no packed form: the compiler emits one plain and one SDWA (source half select) instruction and packs the two results.  With
# the destination half selected as well (dst_unused:UNUSED_PRESERVE) the two results land in one register and the v_pack_b32_f16 goes: one
# vector instruction less per GELU pair (24 per tile).  gfx940-family hazard: a write with a destination select needs one wait state before
# the register is read again (the second instruction preserves - reads - the other half).  Inline asm: not exercised by the host build
# (the whole packed-fp16 GELU is restated there); the GPU run compares bit for bit with the shipped kernel.
_EXP_PAIR = "        e = h16x2_t{(_Float16)__builtin_exp2f16(q.x), (_Float16)__builtin_exp2f16(q.y)};\n"
_EXP_SDWA = ('        uint32_t ew_;\n'
             '        asm("v_exp_f16_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0\\n\\ts_nop 0\\n\\t"\n'
             '            "v_exp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\\n\\ts_nop 0"\n'
             '            : "=&v"(ew_) : "v"(__builtin_bit_cast(uint32_t, q)));\n'
             '        e = __builtin_bit_cast(h16x2_t, ew_);\n')
V["R_exp_sdwa"] = dict(kind="exact", host_checked=False, doc="both v_exp_f16 of a GELU pair write their half of one register (SDWA destination select): no v_pack_b32_f16",
                       patches=[(_EXP_PAIR, _EXP_SDWA)])
# Two workgroups share a CU (one wave of each per SIMD); they are launched together and run the same stage sequence at the same speed, so
# they stay IN PHASE: both in the MFMA phase, then both in the VALU phase.  The SQ counters of the shipped kernel (profiles/r03_pmc_sq_counters.md)
# say exactly that: per tile and wave 1770 VALU-active + 896 MFMA-busy cycles = 2666 of the 2725 the SIMD spends - a sum, not a maximum.  A start
# offset of half a tile period for one workgroup of each pair is preserved by the same argument (equal speeds), and lets one workgroup's MFMA
# phase run under the other's VALU phase.  Which workgroups share a CU is the dispatcher's choice: both pairings are here.
_LOOP_START = "    int q = 0, buf = 0;\n"
def _dephase(cond, sleeps):
    return [(_LOOP_START, f"    if ({cond}) {{\n        _Pragma(\"unroll\") for (int i_ = 0; i_ < {sleeps}; ++i_) __builtin_amdgcn_s_sleep(7);   // 7 x 64 cycles each\n    }}\n" + _LOOP_START)]
V["R_dephase_half"] = dict(kind="exact", doc="second half of the grid starts ~1350 cycles late (half the tile period of a de-phased pair): co-resident workgroups i, i + grid / 2",
                           patches=_dephase("(int)blockIdx.x >= ((int)gridDim.x >> 1)", 3))
V["R_dephase_odd"] = dict(kind="exact", doc="odd workgroups start ~1350 cycles late: de-phases co-resident workgroups 2 j, 2 j + 1",
                          patches=_dephase("(int)blockIdx.x & 1", 3))
V["R_dephase_half_long"] = dict(kind="exact", doc="as R_dephase_half with ~2700 cycles (half of the 5350-cycle tile period the in-phase pair shows today)",
                                patches=_dephase("(int)blockIdx.x >= ((int)gridDim.x >> 1)", 6))
# combinations (the candidates are independent: in-wave overlap, one vector instruction less per GELU pair, a start offset)
_PIPE_LOOP_START = "    int q = 0;\n"
V["R_pipe_sdwa"] = dict(kind="exact", host_checked=False, doc="R_pipelined + R_exp_sdwa", patches=V["R_pipelined"]["patches"] + V["R_exp_sdwa"]["patches"])
V["R_pipe_sdwa_dephase"] = dict(kind="exact", host_checked=False, doc="R_pipelined + R_exp_sdwa + the second half of the grid ~700 cycles late",
                                patches=V["R_pipe_sdwa"]["patches"] + [(_PIPE_LOOP_START, "    if ((int)blockIdx.x >= ((int)gridDim.x >> 1)) {\n        __builtin_amdgcn_s_sleep(11);\n    }\n" + _PIPE_LOOP_START)])
_PIPE_BARRIER = "        __syncthreads();                                 // tile q + 2 staged; output patch of this tile complete\n"
V["T_pipe_no_barrier"] = dict(kind="timing", doc="R_pipelined without its per-tile barrier (racy): the barrier's share in the pipelined form",
                              patches=V["R_pipelined"]["patches"] + [(_PIPE_BARRIER, "        (void)0;\n")])
V["T_no_barrier"] = dict(kind="timing", doc="the per-tile workgroup barrier removed (racy)", patches=[(_BARRIER, "        (void)0;\n")])
V["T_no_ln_stats"] = dict(kind="timing", doc="no LayerNorm statistics (no sums, no exchange, no rsqrt); affine and GELU stay",
                          patches=[(_LN_TWO_PASS, "        const float rstd = a.eps + 1.f;\n")])
V["T_no_gelu"] = dict(kind="timing", doc="both GELUs reduced to their fp16 conversion",
                      patches=[(_CVT, _CVT + "    if (EXPM >= 0) return __builtin_bit_cast(uint32_t, x);\n")])
V["T_no_exp"] = dict(kind="timing", doc="GELU polynomial without the exponential (the library's up_gelu16 = 2 instantiation)",
                     patches=[("    else if (g_tune_up_gelu16) {\n        if (g_uf_prio) hipLaunchKernelGGL((up_fused_kernel<1, 1>)",
                               "    else if (g_tune_up_gelu16) {\n        if (g_uf_prio) hipLaunchKernelGGL((up_fused_kernel<1, 2>)")])
V["T_no_stage1_mfma"] = dict(kind="timing", doc="stage 1 without its 32 MFMAs (operands still read; W1 stays live through one MFMA per k-step)",
                             patches=[(_STAGE1, "            for (int rt = 0; rt < 4; ++rt) {\n"
                                                "                if (rt == (ks & 3)) u[rt] = mfma16d(w1f[rt][ks], kf, u[rt]);\n"
                                                "                else asm volatile(\"\" :: \"v\"(w1f[rt][ks].x), \"v\"(w1f[rt][ks].y), \"v\"(w1f[rt][ks].z), \"v\"(w1f[rt][ks].w));\n"
                                                "            }\n")])
V["T_no_output"] = dict(kind="timing", doc="no output patch and no global store (stage 3 results kept alive by one predicated store)",
                        patches=[(_PATCH_WRITE, "        if (fg == 0 && key0 < 0) {\n"), (_STORE, "        if (tid < 64 * a.nmask && key0 < 0) {\n")])
V["T_one_wg_per_cu"] = dict(kind="timing", doc="grid = number of CUs: what the second co-resident workgroup buys",
                            patches=[(_GRID, "    const int grid = a.nitems < cus ? a.nitems : cus;\n")])
V["T_prio0"] = dict(kind="timing", doc="no issue priority for the stage-1 MFMA phase (the library's msam_upscale_set_prio(0))",
                    patches=[("int g_uf_prio = 1;", "int g_uf_prio = 0;")])

STUBS = r'''
#include <cstdio>
void msam_set_error(const char* msg) { fprintf(stderr, "uf_lab: %s\n", msg); }
int msam_check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { fprintf(stderr, "uf_lab: %s: %s\n", what, hipGetErrorString(e)); return 2; } return 0; }
void msam_profile_mark2(void*, int, double, double, int) {}
'''


def variant_source(name: str) -> str:
    """The shipped source with the variant's patches applied: (old, new) - the text must occur exactly once - or (start, end, new) -
    everything from the start marker up to (not including) the end marker is replaced."""
    src = open(SRC).read()
    for patch in V[name]["patches"]:
        if len(patch) == 3:
            start, end, new = patch
            assert src.count(start) == 1, (name, start[:70], src.count(start))
            i = src.index(start)
            src = src[:i] + new + src[src.index(end, i):]
            continue
        old, new = patch
        assert src.count(old) == 1, (name, old[:70], src.count(old))
        src = src.replace(old, new)
    return src


def build(name: str, outdir: str):
    os.makedirs(outdir, exist_ok=True)
    src = variant_source(name)
    src = src.replace('#include "common.h"', f'#include "{ROOT}/micro_sam_amd/csrc/common.h"')
    src = src.replace('#include "../../include/msam_hip.h"', f'#include "{ROOT}/include/msam_hip.h"')
    import hashlib
    tag = hashlib.sha256((variant_source(name) + open(os.path.join(ROOT, "micro_sam_amd", "csrc", "common.h")).read()).encode()).hexdigest()[:10]
    path, so = os.path.join(outdir, name + ".hip"), os.path.join(outdir, f"{name}.{tag}.so")
    if os.path.exists(so) and os.path.exists(so + ".json"):          # built before from the same text (e.g. in the build container)
        return so, json.load(open(so + ".json"))
    with open(path, "w") as fh:
        fh.write(src + STUBS)
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Rpass-analysis=kernel-resource-usage", path, "-o", so]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.exit(f"{name}: build failed\n{r.stderr[-3000:]}")
    # resource usage of the <1, 1> (or, for T_no_exp, <1, 2>) instantiation
    res = {}
    blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)
    want = "ILi0ELi1E" if name == "T_prio0" else "ILi1ELi2E" if name == "T_no_exp" else "ILi1ELi1E"
    for b in blocks[1:]:
        if want in b.split("\n", 1)[0]:
            for key, pat in (("vgprs", r"VGPRs: (\d+)"), ("agprs", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                             ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
                m = re.search(pat, b)
                if m:
                    res[key] = int(m.group(1))
    with open(so + ".json", "w") as fh:
        json.dump(res, fh)
    return so, res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--list", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--prompts", type=int, default=1024)
    ap.add_argument("--launches", type=int, default=20)
    ap.add_argument("--build-only", action="store_true", help="compile every variant (works without a GPU) and print its register use")
    ap.add_argument("--out", default=os.path.join(ROOT, "tools", "lab_build"), help="objects are cached here by source hash (git-ignored, travels with gpurun)")
    a = ap.parse_args()
    names = [n for n in V if not a.only or n in a.only.split(",")]
    if a.list:
        for n in names:
            print(f"{n:18s} {V[n]['kind']:7s} {V[n]['doc']}")
        return
    built = {n: build(n, a.out) for n in names}
    if a.build_only:
        for n in names:
            print(f"{n:18s} {built[n][1]}")
        return
    import torch
    assert torch.cuda.is_available(), "uf_lab times kernels: it needs the GPU"
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(3)
    P = a.prompts
    keys = (torch.randn(64, 4096, 256, generator=g)).to(torch.float16).to(dev)
    keys = keys.repeat((P + 63) // 64, 1, 1)[:P].contiguous()                    # [P, 4096, 256] fp16, read as the blocked stream
    w1 = (torch.randn(256, 256, generator=g) / 16).to(torch.float16).to(dev)
    b1 = torch.randn(256, generator=g).to(dev)
    lnw = (torch.randn(64, generator=g) * 0.2 + 1).to(dev); lnb = (torch.randn(64, generator=g) * 0.3).to(dev)
    w2 = (torch.randn(128, 64, generator=g) / 8).to(torch.float16).to(dev)
    b2 = torch.randn(32, generator=g).to(dev)
    hyper = torch.randn(P, 4, 128, generator=g).to(dev)
    vp, i32 = ctypes.c_void_p, ctypes.c_int32
    stream = torch.cuda.current_stream().cuda_stream
    results, base_out = {}, None
    for n in names:
        so, res = built[n]
        lib = ctypes.CDLL(so)
        fn = lib.msam_upscale_fused_layout
        fn.restype = i32
        fn.argtypes = [vp, i32, i32, vp, vp, vp, vp, ctypes.c_float, vp, vp, vp, i32, i32, i32, vp, vp]
        out = torch.full((P, 3, 256, 256), float("nan"), device=dev)

        def launch():
            rc = fn(keys.data_ptr(), 1, P, w1.data_ptr(), b1.data_ptr(), lnw.data_ptr(), lnb.data_ptr(), 1e-6, w2.data_ptr(), b2.data_ptr(),
                    hyper.data_ptr(), 128, 1, 3, out.data_ptr(), stream)
            assert rc == 0, (n, rc)
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.launches):
            launch()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.launches
        rec = dict(kind=V[n]["kind"], doc=V[n]["doc"], ms_per_launch=round(ms, 4), **res)
        if n == "base":
            base_out = out.clone()
            rec["finite"] = bool(torch.isfinite(out).all())
        elif base_out is not None and V[n]["kind"] in ("exact", "close"):
            d = (out - base_out).abs()
            rec["max_abs_diff_vs_base"] = float(d.max()); rec["mean_abs_diff_vs_base"] = float(d.mean())
            rec["scale"] = float(base_out.abs().max())
            rec["bit_identical"] = bool(torch.equal(out, base_out))
            rec["ok"] = rec["bit_identical"] if V[n]["kind"] == "exact" else bool(d.max() <= 2e-3 * rec["scale"])
        results[n] = rec
        del out
    base_ms = results.get("base", {}).get("ms_per_launch")
    print(f"{'variant':18s} {'kind':7s} {'ms':>8s} {'vs base':>8s} {'VGPRs':>6s} {'occ':>4s}  check")
    for n, r in results.items():
        dv = f"{r['ms_per_launch'] - base_ms:+.3f}" if base_ms and n != "base" else ""
        chk = "" if "ok" not in r else ("ok" if r["ok"] else "MISMATCH") + (" (bit-identical)" if r.get("bit_identical") else f" (max {r['max_abs_diff_vs_base']:.2e})")
        print(f"{n:18s} {r['kind']:7s} {r['ms_per_launch']:8.3f} {dv:>8s} {r.get('vgprs', '?'):>6} {r.get('occupancy', '?'):>4}  {chk}")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "uf_lab.json"), "w") as fh:
        json.dump(dict(prompts=P, launches=a.launches, results=results), fh, indent=1)


if __name__ == "__main__":
    main()
