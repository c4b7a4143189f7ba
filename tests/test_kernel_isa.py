"""Checks on the gfx950 code object inside libsvohip.so (no GPU needed: llvm-objdump disassembles it here).

Why: round 3's closing commit removed a #define and with it the 12 LDS stores that zero a lane's stack column when a ray
is set up (DescWalk::fresh_stack).  Nothing failed -- a POP to a level its ray never pushed is so rare that no golden, no
fuzz case and none of 900 000 random hostile cameras on the CPU oracle produces one (`cold_pops` of the oracle's
statistics) -- so the property "the kernel that ships has the stores" is checked on the instructions themselves."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.fixture(scope="module")
def disassembly(tmp_path_factory):
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump not installed")
    d = tmp_path_factory.mktemp("isa")
    # --offloading writes the unbundled code objects NEXT TO ITS INPUT: work on a copy (two such files were once committed)
    so = shutil.copy(os.path.join(ROOT, "svo-raytracer_amd", "csrc", "libsvohip.so"), d)
    subprocess.check_call([OBJDUMP, "--offloading", so], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=d)
    co = [f for f in os.listdir(d) if "gfx950" in f]
    assert len(co) == 1, os.listdir(d)
    txt = subprocess.check_output([OBJDUMP, "-d", os.path.join(d, co[0])], text=True)
    kernels = {}
    name = None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m and m.group(1).startswith("L") and not m.group(1).startswith("_Z"):
            continue   # a label of the inline assembly (Ltrip1, LnoD1, ...): still the same kernel
        if m:
            name = m.group(1)
            kernels[name] = []
        elif name and line.strip():
            kernels[name].append(line.split()[0])
    return kernels


def _kernel(kernels, *parts):
    hit = [k for k in kernels if all(p in k for p in parts)]
    assert len(hit) == 1, (parts, hit)
    return kernels[hit[0]]


@pytest.mark.parametrize("tab", [0, 1])
@pytest.mark.parametrize("cams", [0, 1])
@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_descriptor_walk_zeroes_a_new_ray_s_stack_column(disassembly, mode, cams, tab):
    """trav_loop2's POP reads {descriptor, t_max} of its level unconditionally (no pushed-levels mask): every ray must
    start on a zeroed column = the reference's zero-initialised octstack (svotrace.comp:227).  12 levels of 8 bytes per
    lane: six ds_write2st64_b64 (or twelve ds_write_b64) next to the loop's one PUSH (ds_write2_b32)."""
    # (tab: the kernels that read the launch's row / column tables -- what a launch without folded samples runs since round 5)
    ins = _kernel(disassembly, "persist_kernelILi%dE" % mode, "DescWalkELb%dELb%dE" % (cams, tab))
    levels = 2 * ins.count("ds_write2st64_b64") + ins.count("ds_write_b64")
    assert levels >= 12, {k: ins.count(k) for k in set(ins) if k.startswith("ds_")}
    assert ins.count("ds_write2_b32") >= 1   # the PUSH


def test_persistent_kernels_use_no_scratch_beyond_the_known_spills(disassembly):
    """a tripwire, not a target: the descriptor walk's mode-0 kernel must keep its traversal loop free of scratch traffic
    (spills live in round code only) -- a `scratch_` / `buffer_..._offen` spill inside the loop body would show up as a
    jump in these counts"""
    for tab in (0, 1):
        ins = _kernel(disassembly, "persist_kernelILi0E", "DescWalkELb0ELb%dE" % tab)
        spills = sum(1 for i in ins if i.startswith("scratch_"))
        assert spills < 120, (tab, spills)


def test_descriptor_walk_stack_is_the_kernel_s_only_lds_object(tmp_path):
    """trav_loop2 forms a level's stack address from the scale alone (svo_travloop2.h: SVO_PUSH_ADDR / SVO_POP_ADDR without the
    record walk's clamp): a push below level 12 -- only reachable from the phantom state behind a POP to a never-pushed level, which
    no input produces -- lands below the lane's column, i.e. below LDS offset 0, and is dropped by the hardware, a pop there reads
    zero.  That holds only while the stack is the kernel's SOLE shared object, starting at LDS offset 0: the code object must say
    6 144 bytes of LDS (12 levels x 64 lanes x 8 bytes) for every descriptor-walk kernel, not a byte more."""
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(OBJDUMP) and os.path.exists(readelf)):
        pytest.skip("llvm tools not installed")
    so = shutil.copy(os.path.join(ROOT, "svo-raytracer_amd", "csrc", "libsvohip.so"), tmp_path)
    subprocess.check_call([OBJDUMP, "--offloading", so], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp_path)
    co = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    notes = subprocess.check_output([readelf, "--notes", os.path.join(tmp_path, co[0])], text=True)
    lds, seen = None, 0
    for line in notes.splitlines():
        m = re.match(r"\s*\.group_segment_fixed_size:\s*(\d+)", line)
        if m:
            lds = int(m.group(1))
        m = re.match(r"\s*\.name:\s*(\S+)", line)
        if m and "persist_kernel" in m.group(1) and "DescWalk" in m.group(1):
            assert lds == 12 * 64 * 8, (m.group(1), lds)
            seen += 1
    assert seen >= 20, seen
