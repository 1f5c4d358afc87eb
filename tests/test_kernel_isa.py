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


def _descriptors(isa: str):
    """kernel name -> {directive: value} of its .amdhsa_kernel block"""
    out = {}
    for m in re.finditer(r"^\s*\.amdhsa_kernel (\S+)\n(.*?)^\s*\.end_amdhsa_kernel", isa, re.M | re.S):
        out[m.group(1)] = {k: v for k, v in re.findall(r"^\s*\.amdhsa_(\w+)\s+(\S+)", m.group(2), re.M)}
    return out


def test_huffman_kernel_keeps_its_bit_reader_in_registers(isa):
    lines = _kernel(isa, "rg_mp3_huffman_kernel")
    bad = [x for x in lines if x.startswith("scratch_") or x.startswith("flat_")]
    assert not bad, f"scratch / flat accesses in the Huffman kernel: {bad[:3]}"
    desc = [v for k, v in _descriptors(isa).items() if "rg_mp3_huffman_kernel" in k]
    assert len(desc) == 1 and int(desc[0]["private_segment_fixed_size"]) == 0, desc
    assert len(lines) < 5000, "the Huffman kernel has outgrown the instruction cache's comfort (was 66 KB once, with unrolled scalefactor loops)"


def test_no_kernel_of_the_decoder_touches_scratch_memory(isa):
    """Every kernel of rg_mp3dev.hip: no private segment, no scratch instruction.  (Round 5 shipped the frame parser with
    208 bytes of scratch per thread -- the host's record writer fills a private array through a running index; the records and
    the frame's bytes wait in LDS now -- and round 6's window wave spilled address registers until its tables moved to LDS: a
    reload of a spilled register in a loop is a wait for every store in flight.)"""
    desc = _descriptors(isa)
    assert len(desc) >= 8, sorted(desc)
    for name, d in desc.items():
        assert int(d["private_segment_fixed_size"]) == 0, f"{name}: {d['private_segment_fixed_size']} bytes of scratch per thread"
    assert "scratch_" not in isa
    back = [d for k, d in desc.items() if "rg_mp3_backhalf_kernel" in k][0]
    assert int(back["next_free_vgpr"]) <= 128 and int(back["group_segment_fixed_size"]) <= 40960, "four blocks per CU: 128 VGPRs, 40 KB of LDS"


def test_back_half_runs_its_dct_on_the_matrix_cores_beside_the_window_sums(isa):
    """Wave 3's step is ONE straight line: 24 v_mfma_f32_16x16x4_f32, each followed by twelve fused multiply-adds of the
    window sums, PCM leaving through global (not flat: a flat store counts as an LDS access at the block's barrier) stores."""
    lines = _kernel(isa, "rg_mp3_backhalf_kernel")
    mf = [i for i, x in enumerate(lines) if x.startswith("v_mfma_f32_16x16x4_f32")]
    assert len(mf) == 72, len(mf)  # three instantiations with a granule to transform (window: none / history only / sums)
    assert not [x for x in lines if x.startswith("flat_store") or x.startswith("flat_load")]
    # the steady-state instantiation: between two consecutive matrix instructions at least eight v_fma / v_fmac (of its twelve: the compiler folds a sum's first term)
    best = 0
    for k in range(0, 72, 24):
        seg = mf[k:k + 24]
        gaps = [sum(1 for x in lines[a:b] if x.startswith("v_fma")) for a, b in zip(seg, seg[1:])]
        best = max(best, min(gaps))
    assert best >= 8, best
