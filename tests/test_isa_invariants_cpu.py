"""Properties of the ISA hipcc emits for the hot kernels that the parity claims rest on, checked without a GPU.

* the IDCT butterfly (FastFloatingPointDCT.TransformIDCT, FastFloatingPointDCT.cs:54-127) must not be contracted: every
  multiply and add of the reference is one IEEE operation (`-ffp-contract=off`); the headline variant is 256 v_pk_add_f32 +
  128 v_pk_mul_f32 + 64 v_rndne_f32 per block and no fused multiply-add but the one of the address set-up's division;
* no hot kernel spills vector registers or uses scratch;
* the K2S round kernel's burst of symbol steps is straight-line code: no branch between its first and last ds_add;
* the encoder's fused E1 kernel (fdct_fused_kernel) keeps the FDCT un-contracted too: its only fused multiply-adds are the
  exact ones (colour conversion on 24-bit operands, the quotient's correction steps that reproduce the hardware division,
  the chroma sample's t / 4 - 127.5), it stays under 168 registers (three waves per SIMD) and does not spill."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "jpeglibrary_amd", "csrc")
# the decode kernels' translation units (round 5: kernels.hip split by stage)
SRCS = [os.path.join(CSRC, f) for f in ("k1_markers.hip", "k2_huffman.hip", "k2s_subseq.hip", "k3_idct.hip")]


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    flags = open(os.path.join(ROOT, "jpeglibrary_amd", "csrc", "Makefile")).read()
    m = re.search(r"^CXXFLAGS\s*[:?]?=\s*(.*)$", flags, re.M)
    cxxflags = m.group(1).split() if m else ["-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math"]
    out = tmp_path_factory.mktemp("isa")
    procs = []
    for src in SRCS:
        asm = out / (os.path.basename(src) + ".s")
        procs.append((asm, subprocess.Popen([hipcc, "--offload-arch=gfx950", *[f for f in cxxflags if not f.startswith("-W")], "-S", "--cuda-device-only",
                                             "-o", str(asm), src], stderr=subprocess.DEVNULL)))
    text = []
    for asm, p in procs:
        assert p.wait() == 0, asm
        text.append(asm.read_text())
    return "\n".join(text)


def _body(text, mangled_prefix):
    lines = text.splitlines()
    start = next(i for i, ln in enumerate(lines) if ln.startswith(mangled_prefix) and ln.rstrip().split(";")[0].strip().endswith(":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    return [ln.strip() for ln in lines[start:end] if ln.strip() and not ln.strip().startswith(";")]


@pytest.mark.timeout(600)
def test_the_idct_butterfly_is_not_contracted(isa):
    body = _body(isa, "_ZN5jpgpu18idct_output_kernelILi0ELi3EEE")  # INTERLEAVED_U8, 4:2:0: the headline variant
    count = lambda op: sum(1 for ln in body if re.sub(r"_(e32|e64|sdwa|dpp)$", "", ln.split()[0]) == op)
    assert count("v_pk_add_f32") == 256
    assert count("v_pk_mul_f32") == 128
    assert count("v_rndne_f32") == 64
    assert count("v_fma_f32") + count("v_fmac_f32") + count("v_pk_fma_f32") <= 1  # the reciprocal of the address set-up's integer division
    assert not any(ln.split()[0].startswith("v_mfma") for ln in body)


@pytest.mark.timeout(600)
def test_no_hot_kernel_spills_or_uses_scratch(isa):
    names = re.findall(r"\.name:\s+(\S+)", isa)
    spills = dict(zip(names, re.findall(r"\.vgpr_spill_count:\s+(\d+)", isa)))
    scratch = dict(zip(names, re.findall(r"\.private_segment_fixed_size:\s+(\d+)", isa)))
    hot = [n for n in names if any(k in n for k in ("idct_output_kernel", "huffman_decode_kernel", "subseq_round_kernel", "subseq_final_kernel",
                                                    "marker_count_kernel", "marker_write_kernel"))]
    assert len(hot) >= 20
    for n in hot:
        assert spills[n] == "0", (n, spills[n])
        assert scratch[n] == "0", (n, scratch[n])


@pytest.mark.timeout(600)
def test_the_round_kernels_burst_is_straight_line_code(isa):
    body = _body(isa, "_ZN5jpgpu19subseq_round_kernel")
    adds = [i for i, ln in enumerate(body) if ln.startswith("ds_add_u32")]
    # the burst's steps (one ds_add into the lane's DC sum each) come first; the exact path's commit has one more
    assert len(adds) >= 8
    burst = body[adds[0]:adds[7] + 1]
    assert not any(ln.startswith(("s_cbranch", "s_branch", "s_setpc")) for ln in burst), [ln for ln in burst if ln.startswith("s_")]


@pytest.fixture(scope="module")
def enc_isa(tmp_path_factory):
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    flags = open(os.path.join(ROOT, "jpeglibrary_amd", "csrc", "Makefile")).read()
    m = re.search(r"^CXXFLAGS\s*[:?]?=\s*(.*)$", flags, re.M)
    cxxflags = m.group(1).split() if m else ["-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math"]
    asm = tmp_path_factory.mktemp("isa") / "encode_kernels.s"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", *[f for f in cxxflags if not f.startswith("-W")], "-S", "--cuda-device-only", "-o", str(asm),
                           os.path.join(ROOT, "jpeglibrary_amd", "csrc", "encode_kernels.hip")], stderr=subprocess.DEVNULL)
    return asm.read_text()


@pytest.mark.timeout(600)
def test_the_fused_encoder_kernel_keeps_the_fdct_in_ieee_steps(enc_isa):
    """FastFloatingPointDCT.TransformFDCT (FastFloatingPointDCT.cs:194-362) has 14 multiplications and 26 additions per
    8-point butterfly and no fused step; the kernel runs it on packed pairs.  Per luma round it holds one pass-1 and one
    pass-2 butterfly, per chroma round the same: every v_pk_fma_f32 must belong to one of the three exact uses."""
    body = _body(enc_isa, "_ZN5jpgpu17fdct_fused_kernel")
    op = lambda ln: re.sub(r"_(e32|e64|sdwa|dpp)$", "", ln.split()[0])
    count = lambda name: sum(1 for ln in body if op(ln) == name)
    # no scalar fused multiply-add outside quant_pair's two (the refined reciprocal, computed once per table entry)
    assert count("v_fma_f32") + count("v_fmac_f32") <= 2 * 2
    # butterflies: 4 packed ones in the edge variant + fast variant of pass 1 (2), pass 2 of luma (1) and chroma (2 in the loop body: pass 1 + pass 2)
    n_butterflies = 2 + 1 + 2
    assert count("v_pk_mul_f32") >= 14 * n_butterflies
    assert count("v_pk_add_f32") >= 26 * n_butterflies
    names = re.findall(r"\.name:\s+(\S+)", enc_isa)
    vgprs = dict(zip(names, re.findall(r"\.vgpr_count:\s+(\d+)", enc_isa)))
    spills = dict(zip(names, re.findall(r"\.vgpr_spill_count:\s+(\d+)", enc_isa)))
    sspills = dict(zip(names, re.findall(r"\.sgpr_spill_count:\s+(\d+)", enc_isa)))
    scratch = dict(zip(names, re.findall(r"\.private_segment_fixed_size:\s+(\d+)", enc_isa)))
    lds = dict(zip(names, re.findall(r"\.group_segment_fixed_size:\s+(\d+)", enc_isa)))
    n = next(x for x in names if "fdct_fused_kernel" in x)
    assert int(vgprs[n]) <= 168 and spills[n] == "0" and sspills[n] == "0" and scratch[n] == "0"
    assert int(lds[n]) * 12 <= 160 * 1024  # twelve waves per CU


@pytest.mark.timeout(600)
def test_the_look_back_chains_neither_write_back_nor_invalidate_the_l2(isa, enc_isa):
    """bits_emit_kernel (encoder, E2 + E3 in one pass) and marker_onepass_kernel (K1 in one pass) hand their chain records from
    workgroup to workgroup as RELAXED device-scope atomics.  A release / acquire pair at that scope is `buffer_wbl2` / `buffer_inv` on
    this part -- the XCD's whole L2 written back at every publish and invalidated at every poll, with every other workgroup's output
    in it: the first versions of bits_emit_kernel took 18-57 ms instead of 2.5 (LAB_NOTEBOOK 8).  Neither instruction may appear;
    the records must travel as device-scope (sc1) accesses, and neither kernel may spill."""
    for text, prefix, name in ((enc_isa, "_ZN5jpgpu16bits_emit_kernel", "bits_emit_kernel"), (isa, "_ZN5jpgpu21marker_onepass_kernel", "marker_onepass_kernel")):
        body = _body(text, prefix)
        ops = [ln.split()[0] for ln in body]
        assert not any(o.startswith("buffer_wbl2") or o.startswith("buffer_inv") for o in ops), name
        assert any(("global_load" in ln or "global_atomic" in ln) and "sc1" in ln for ln in body), name   # the polls
        assert any("global_store" in ln and "sc1" in ln for ln in body), name                              # the publishes
        names = re.findall(r"\.name:\s+(\S+)", text)
        spills = dict(zip(names, re.findall(r"\.vgpr_spill_count:\s+(\d+)", text)))
        scratch = dict(zip(names, re.findall(r"\.private_segment_fixed_size:\s+(\d+)", text)))
        n = next(x for x in names if name in x)
        assert spills[n] == "0" and scratch[n] == "0", name
