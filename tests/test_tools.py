"""CPU checks of the analysis tools the design leans on (no GPU)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_bank_model_separates_the_cell_pitches():
    """tools/conv_bank_sim.py models gfx950's ds_read_b128 lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, + 32): the
    48- / 80-byte cells of rounds 2-4 are two-way conflicted on every A-operand read, the pitches csrc/common.h bf16_cell_bytes
    returns (32 bytes mod 64) are not, and the fused conv-GRU cell's regions have to sit whole bank rows apart."""
    import conv_bank_sim as B

    assert B.conv_reads(CI=16, CS=48, PX=34)[0] > 7.0
    assert B.conv_reads(CI=16, CS=32, PX=34)[0] < 4.5
    assert B.conv_reads(CI=32, CS=80, PX=66)[0] == 8.0
    assert B.conv_reads(CI=32, CS=96, PX=66)[0] == 4.0
    assert B.conv_reads(CI=32, CS=32, PX=66, split_planes=16, plane_stride=66 * 10 * 32 + 64)[0] == 4.0
    assert B.gru_gate_reads(HID=8, XC=16, PITCH=66, REG=12 * 66 * 16)[0] == 8.0
    assert B.gru_gate_reads(HID=8, XC=16, PITCH=66, REG=12800)[0] == 4.0
    # the helper's formula (common.h): 16 bytes up to 8 channels, else the smallest pitch >= 2 C that is 32 mod 64
    cell = lambda C: 16 if C <= 8 else ((2 * C - 32 + 63) // 64) * 64 + 32
    assert [cell(c) for c in (8, 16, 24, 32, 40, 48, 64)] == [16, 32, 96, 96, 96, 96, 160]


def test_asm_mix_counts_a_listing(tmp_path):
    """tools/asm_mix.py: instruction classes of a kernel body, whole and after the first s_barrier."""
    import subprocess

    listing = tmp_path / "k.s"
    listing.write_text("_Zkern:\n\tv_add_f32 v0, v1, v2\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n.LBB0_1:\n\tv_mfma_f32_16x16x32_bf16 v[0:3], v[4:7], v[8:11], v[0:3]\n"
                       "\tds_read_b128 v[4:7], v12\n\tbuffer_load_dwordx4 v[8:11], v13, s[0:3], 0 offen\n\ts_cbranch_scc1 .LBB0_1\n\ts_endpgm\n")
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "asm_mix.py")
    out = subprocess.run([sys.executable, tool, str(listing), "kern"], capture_output=True, text=True, check=True).stdout
    assert "kernel: 2 vector (1 mfma) 3 scalar 1 ds 1 global" in out
    assert "after the first barrier: 1 vector (1 mfma) 2 scalar 1 ds 1 global" in out
