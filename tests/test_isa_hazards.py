"""CPU: a scan of the gfx950 ISA hipcc emits for the kernels that pack bf16 operands with inline-asm `v_cvt_pk_bf16_f32` (no builtin on
gfx950) and feed them to matrix instructions.  hipcc pads no hazard whose producer sits inside an asm string; a VALU-written VGPR
needs two wait states before an MFMA reads it as its A / B operand.  Round 5 found the unpadded pattern giving stale operands on
~4 % of the tiles of the new attention forward (csrc/sra_attention.hip) and, latent, 7 times in the weight-gradient kernels
(csrc/gemm_tn.hip): both now end their pack statements with `s_nop 1`.  This test compiles the sources to assembly (hipcc
cross-compiles without a GPU) and requires ZERO `v_cvt_pk -> v_mfma` pairs closer than two wait states."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "combo-avs_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def hazards(asm_text):
    """-> (number of v_mfma instructions, [(line, mfma, cvt)] pairs with < 2 wait states between a v_cvt_pk_bf16_f32 and an MFMA
    reading its destination as A or B).  An instruction between the two counts as one wait state, `s_nop N` as N + 1."""
    lines = [ln.strip() for ln in asm_text.splitlines()]
    lines = [ln for ln in lines if ln and not ln.startswith((";", ".", "//")) and not ln.endswith(":")]
    found, n_mfma = [], 0
    for i, ln in enumerate(lines):
        if not ln.startswith("v_mfma"):
            continue
        n_mfma += 1
        ops = [t.strip() for t in ln.split(None, 1)[1].split(",")]
        src = _regs(ops[1]) | _regs(ops[2])
        states = 0
        for j in range(i - 1, max(i - 4, -1), -1):
            p = lines[j]
            if p.startswith(("v_cvt_pk_bf16_f32", "v_cvt_pk_f16_f32")):  # (fp16 pieces: round 6, csrc/gemm_nt3.hip F16)
                if _regs(p.split(None, 1)[1].split(",")[0].strip()) & src and states < 2:
                    found.append((i, ln, p))
                    break
            states += int(p.split()[1]) + 1 if p.startswith("s_nop") else 1
    return n_mfma, found


def test_the_scanner_sees_an_unpadded_pair_and_accepts_a_padded_one():
    bad = "v_cvt_pk_bf16_f32 v7, v38, v39\ns_waitcnt lgkmcnt(1)\nv_mfma_f32_32x32x16_bf16 v[32:47], v[0:3], v[4:7], 0\n"
    good = "v_cvt_pk_bf16_f32 v7, v38, v39\ns_nop 1\nv_mfma_f32_32x32x16_bf16 v[32:47], v[0:3], v[4:7], 0\n"
    other = "v_cvt_pk_bf16_f32 v9, v38, v39\nv_mfma_f32_32x32x16_bf16 v[32:47], v[0:3], v[4:7], 0\n"
    assert len(hazards(bad)[1]) == 1 and not hazards(good)[1] and not hazards(other)[1]


@pytest.mark.parametrize("src", ["sra_attention.hip", "gemm_tn.hip", "gemm_nt3.hip"])
def test_no_cvt_pk_to_mfma_pair_closer_than_two_wait_states(src, tmp_path):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    out = tmp_path / (src + ".s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
           "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    n_mfma, found = hazards(out.read_text())
    assert n_mfma > 50, n_mfma  # (the kernels are in there)
    assert not found, found[:5]
