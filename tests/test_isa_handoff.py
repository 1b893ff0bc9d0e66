"""Disassembly check of the in-launch hand-off protocol: on gfx950 hipcc's `__syncthreads()` is a
bare `s_barrier` -- no `s_waitcnt vmcnt(0)` -- so a hand-off word set behind it could overtake another wave's `sc1`
stores.  The producers of `tgp_prep.hpp` use `handoff_barrier()` (inline asm: `s_waitcnt vmcnt(0) lgkmcnt(0)`, then
`s_barrier`); this test reads the ISA of the built `k_prep_a` and `k_bwd` and holds every publication of a progress
word (SY_TILES = status[4]; the backward launch's column / row block counts = status[6]) to that form.
CPU-only: it disassembles the object the Makefile built (skipped when the build directory or llvm-objdump is absent)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "tgp", "pytorch_amd", "csrc", "build", "tgp_mm.o")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _device_isa(tmp_path):
    if not (os.path.exists(OBJ) and os.path.exists(OBJDUMP)):
        pytest.skip("needs the in-tree build (make -C tgp/pytorch_amd/csrc) and llvm-objdump")
    o = os.path.join(str(tmp_path), "tgp_mm.o")
    shutil.copy(OBJ, o)
    subprocess.check_call([OBJDUMP, "--offloading", o], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dev = [f for f in os.listdir(str(tmp_path)) if "gfx950" in f]
    assert dev, "no gfx950 code object in tgp_mm.o"
    return subprocess.check_output([OBJDUMP, "-d", os.path.join(str(tmp_path), dev[0])], text=True)


def _function(isa, mangled_prefix):
    lines = isa.split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <%s" % mangled_prefix, l))
    end = next((i for i in range(start + 1, len(lines)) if re.match(r"^[0-9a-f]+ <", lines[i])), len(lines))
    return lines[start:end]


def test_prepare_launch_publishes_behind_a_store_drain(tmp_path):
    body = _function(_device_isa(tmp_path), "_ZN3tgp8k_prep_a")
    # SY_TILES = status[4]: the no-return agent-scope add at byte offset 16 of the status pointer
    pubs = [i for i, l in enumerate(body) if "global_atomic_add " in l and "offset:16" in l]
    assert pubs, "k_prep_a no longer publishes SY_TILES with a global atomic add: update this check"
    for i in pubs:
        window = [l.split("//")[0] for l in body[max(0, i - 40):i]]
        bar = max((k for k, l in enumerate(window) if "s_barrier" in l), default=-1)
        assert bar >= 0, "SY_TILES is published with no workgroup barrier in front of it"
        wait = [k for k, l in enumerate(window[:bar]) if "s_waitcnt" in l and "vmcnt(0)" in l]
        assert wait and bar - wait[-1] <= 2, ("the barrier in front of the SY_TILES add is not preceded by "
                                              "s_waitcnt vmcnt(0): a wave's sc1 stores may still be in flight")
        # nothing that stores to global memory sits between the drain and the publication
        assert not any("global_store" in l for l in window[wait[-1]:]), "a global store between the drain and the add"


def test_backward_launch_publishes_behind_a_store_drain(tmp_path):
    body = _function(_device_isa(tmp_path), "_ZN3tgp5k_bwdE")
    # status[6] (byte offset 24): +1 by a finished column block (its Q tiles are out), +0x10000 by a finished row block (PP)
    pubs = [i for i, l in enumerate(body) if "global_atomic_add " in l and "offset:24" in l]
    assert len(pubs) >= 2, "k_bwd no longer publishes its progress counts with global atomic adds at status[6]: update this check"
    for i in pubs:
        window = [l.split("//")[0] for l in body[max(0, i - 40):i]]
        bar = max((k for k, l in enumerate(window) if "s_barrier" in l), default=-1)
        assert bar >= 0, "a progress count is published with no workgroup barrier in front of it"
        wait = [k for k, l in enumerate(window[:bar]) if "s_waitcnt" in l and "vmcnt(0)" in l]
        assert wait and bar - wait[-1] <= 2, "the barrier in front of a progress add is not preceded by s_waitcnt vmcnt(0)"
        assert not any("global_store" in l for l in window[wait[-1]:]), "a global store between the drain and the add"
