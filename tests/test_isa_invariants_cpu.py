"""What the hand-issued refill load of the K2S round kernel relies on, checked in the ISA hipcc emits (no GPU needed).

uba_next_word (jpeglibrary_amd/csrc/kernels.hip) loads the next 16 bytes of a lane's stream with an inline-asm
global_load_dword x 4 into the registers of the queue itself and does NOT wait: the compiler does not know a load is in
flight.  That is only right as long as the compiler does not touch those registers before the next hand-written
s_waitcnt -- with one 128-bit load it did (it gathered the queue into a register tuple and copied it back out of registers
the load had not reached yet).  The parity tests on the GPU would see the stale words; this test sees the copy itself."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "jpeglibrary_amd", "csrc", "kernels.hip")


def _regs(text):
    """vector registers an operand string names: v7, v[4:7]"""
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(x) for x in re.findall(r"\bv(\d+)\b", text))
    return out


@pytest.mark.timeout(600)
def test_nothing_reads_the_refill_registers_behind_the_hand_issued_load(tmp_path):
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    flags = open(os.path.join(ROOT, "jpeglibrary_amd", "csrc", "Makefile")).read()
    m = re.search(r"^CXXFLAGS\s*[:?]?=\s*(.*)$", flags, re.M)
    cxxflags = m.group(1).split() if m else ["-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math"]
    asm = tmp_path / "kernels.s"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", *[f for f in cxxflags if not f.startswith("-W")], "-S", "--cuda-device-only", "-o", str(asm), SRC],
                          stderr=subprocess.DEVNULL)
    lines = asm.read_text().splitlines()
    start = next(i for i, ln in enumerate(lines) if ln.startswith("_ZN5jpgpu19subseq_round_kernel"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    blocks = [i for i, ln in enumerate(body) if "ASMSTART" in ln and i + 4 < len(body) and all("global_load_dword " in body[i + k] for k in (1, 2, 3, 4))]
    assert len(blocks) == 1, "the hand-issued refill (four global_load_dword in one asm statement) should appear exactly once"
    i = blocks[0]
    loaded = set()
    for k in (1, 2, 3, 4):
        loaded |= _regs(body[i + k].split(",")[0])  # the destination operand
    assert len(loaded) == 4
    # in front of it: the hand-written wait, then the queue takes the old chunk out of these registers
    before = [ln.strip() for ln in body[max(0, i - 12):i] if ln.strip() and not ln.strip().startswith(";")]
    assert any(ln.startswith("s_waitcnt vmcnt(0)") for ln in before), before
    # behind it, up to the end of the basic block: nobody reads (or overwrites) them
    j = i + 5
    assert "ASMEND" in body[j]
    for ln in body[j + 1:]:
        t = ln.strip()
        if not t or t.startswith(";"):
            continue
        if t.endswith(":") or t.startswith("s_branch") or t.startswith("s_cbranch") or t.startswith("s_endpgm"):
            break
        assert not (_regs(t) & loaded), "touches a register the refill load is still writing: " + t
    # ... and anywhere in the loop around it (the registers are loop-carried: nothing else may live in them), whoever touches
    # them does so in a basic block that has waited for ALL outstanding loads first and has not issued one since
    header = max(k for k in range(i) if "=>This Loop Header: Depth=1" in body[k])
    name = re.match(r"\.L(BB\d+_\d+):", body[header]).group(1)  # e.g. BB8_59, as the block comments name the loop
    last = max(k for k in range(header, len(body)) if ("Header=" + name) in body[k] or ("Parent Loop " + name) in body[k])
    stop = next((k for k in range(last + 1, len(body)) if re.match(r"\.LBB\d+_\d+:", body[k])), len(body))
    waited = False
    for k in range(header, stop):
        t = body[k].strip()
        if not t or t.startswith(";"):
            continue
        if re.match(r"\.LBB\d+_\d+:", t) or t.startswith("s_branch") or t.startswith("s_cbranch"):
            waited = False
            continue
        if t.startswith("s_waitcnt vmcnt(0)"):
            waited = True
            continue
        if t.startswith("global_load") or t.startswith("buffer_load") or t.startswith("flat_load"):
            if i < k < i + 5:
                waited = False  # the hand-issued loads themselves
                continue
            waited = False
        if _regs(t) & loaded:
            assert waited, "line %d of the kernel touches a refill register without a full wait in its block: %s" % (k, t)
