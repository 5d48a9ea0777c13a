"""What csrc/upfused.hip's hand-counted `s_waitcnt lgkmcnt(n)` rely on, checked on the compiled ISA (hipcc cross-compiles gfx950 without a GPU):

  * no scalar memory load inside the tile loop - SMEM shares the lgkm counter and returns out of order, which would void "at most n younger
    LDS reads in flight" (the compiler itself only ever waits lgkmcnt(0) while one is pending);
  * the shipped instantiations keep their registers (no scratch) at two waves per SIMD.

The kernel's results are checked elsewhere (tests/test_upfused_kernel_host_emulation.py on the CPU, tests/test_gpu_kernels.py on the device)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if not (os.path.exists(HIPCC) or shutil.which("hipcc")):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("uf_isa")
    src = os.path.join(ROOT, "micro_sam_amd", "csrc", "upfused.hip")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-Rpass-analysis=kernel-resource-usage", src, "-o", str(out / "uf.s")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return open(out / "uf.s").read(), r.stderr


def _kernels(text):
    """name -> instruction lines of every up_fused_kernel instantiation"""
    res = {}
    for m in re.finditer(r"^(_ZN\S*up_fused_kernel\S*):", text, re.M):
        body = text[m.end():text.index("s_endpgm", m.end())]
        res[m.group(1)] = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
    return res


def test_no_scalar_memory_load_in_the_tile_loop(isa):
    text, _ = isa
    ks = _kernels(text)
    assert len(ks) >= 6
    for name, lines in ks.items():
        hdr = [i for i, l in enumerate(lines) if "Inner Loop Header" in l or (l.startswith(".LBB") and "Loop Header" in l)]
        assert hdr, name
        loop = lines[hdr[-1]:]                             # the tile loop is the kernel's last loop (the earlier ones fill the LDS parameter block)
        assert any(l.startswith("s_barrier") for l in loop) and sum(1 for l in loop if l.startswith("v_mfma")) >= 56, name
        smem = [l for l in loop if l.startswith("s_load_") or l.startswith("s_buffer_load") or l.startswith("s_scratch_load")]
        assert not smem, (name, smem[:3])
        if "ILi1ELi1ELi0ELi" in name:                      # the shipped instantiations: the hand-placed reads and waits are there
            assert sum(1 for l in loop if l.startswith("ds_read_b128")) >= 40, name
            assert any(re.match(r"s_waitcnt lgkmcnt\(3\)", l) for l in loop) and any(re.match(r"s_waitcnt lgkmcnt\(1\)", l) for l in loop), name


def test_shipped_instantiations_do_not_spill(isa):
    _, remarks = isa
    blocks = remarks.split("Function Name: ")[1:]
    seen = 0
    for b in blocks:
        head = b.split("\n", 1)[0]
        if "up_fused_kernelILi1ELi1ELi0ELi" not in head and "up_fused_kernelILi1ELi0ELi0ELi" not in head:
            continue
        seen += 1
        assert int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1)) == 0, head
        assert int(re.search(r"Occupancy \[waves/SIMD\]: (\d+)", b).group(1)) == 2, head
        assert int(re.search(r" VGPRs: (\d+)", b).group(1)) <= 256, head
    assert seen >= 4
