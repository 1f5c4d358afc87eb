"""Compiler-made stalls of the MP3 back half, held down by reading the compiled code (hipcc cross-compiles without a GPU).

Round 5 found two of them in the ISA, each a memory round trip per pipeline step of the requantisation wave and neither visible
in the source: a `v_mad_u64_u32` whose unused upper addend was allocated to a register of the prefetch just
issued, and a load merged with an earlier one and waited for on the spot.  Both show as an `s_waitcnt vmcnt(..)` between the
prefetch of a step -- the run of spectra loads -- and the step's barrier, where nothing may wait for memory.  Likewise the Huffman
kernel must not touch scratch memory (a lambda with pointer out-parameters once put its bit reader there)."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if not Path(HIPCC).exists():
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / "rg_mp3dev.s"
    subprocess.run([HIPCC, "-O3", "-Wno-missing-braces", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only", "-S",
                    str(ROOT / "mp3rgain_amd" / "csrc" / "rg_mp3dev.hip"), "-o", str(out)], check=True, capture_output=True, timeout=600)
    return out.read_text()


def _kernel(isa: str, name: str):
    m = re.search(r"^(_Z\d+%s\w*):" % name, isa, re.M)
    assert m, name
    body = isa[m.end():]
    body = body[:body.index(".end_amdhsa_kernel")]
    return [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]


def test_nothing_waits_for_memory_between_a_steps_prefetch_and_its_barrier(isa):
    lines = _kernel(isa, "rg_mp3_backhalf_kernel")
    checked = 0
    for i, l in enumerate(lines):
        if not l.startswith("global_load_dwordx4"):  # the units' load opens the prefetch of a step
            continue
        seg = lines[i:i + 80]
        loads = [j for j, x in enumerate(seg) if x.startswith("global_load_dword ") or x.startswith("global_load_dwordx2")]
        if len(loads) < 6:  # (the loop's prefetch, not the prologue's few loads)
            continue
        # the gains of the step follow the prefetch in a straight line (a few hundred instructions before the rounds begin):
        # both stalls sat within the first hundred
        tail = lines[i + loads[-1] + 1:i + loads[-1] + 1 + 250]
        if any(x.startswith("global_load") for x in tail[:40]):  # a prologue: more loads follow at once
            continue
        checked += 1
        waits = [x for x in tail if "vmcnt" in x and not x.startswith("global_")]
        assert not waits, f"the requantisation wave waits for memory behind its prefetch: {waits[:3]}"
    assert checked >= 2, "no prefetch sequence recognised: the kernel's shape changed, rewrite this test"


def test_huffman_kernel_keeps_its_bit_reader_in_registers(isa):
    lines = _kernel(isa, "rg_mp3_huffman_kernel")
    bad = [x for x in lines if x.startswith("scratch_") or x.startswith("flat_")]
    assert not bad, f"scratch / flat accesses in the Huffman kernel: {bad[:3]}"
    meta = isa[isa.index("rg_mp3_huffman_kernel"):]
    m = re.search(r"\.private_segment_fixed_size:\s*(\d+)", isa[isa.index(".amdhsa_kernel") if ".amdhsa_kernel" in isa else 0:])
    assert len(lines) < 5000, "the Huffman kernel has outgrown the instruction cache's comfort (was 66 KB once, with unrolled scalefactor loops)"
