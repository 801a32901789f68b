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


def test_store_hazard_scan_finds_the_pattern_and_the_library_is_clean():
    """tools/store_hazard_scan.py (round 5): gfx950 still reads the data registers of a 12- / 16-byte buffer store with an SGPR
    soffset when the next vector instructions issue, LLVM guards only an immediate soffset, and an overwrite right behind such a
    store stored garbage in lanes 12-15 (the stride-2 fused conv-GRU cell with 8-row tiles; csrc/common.h
    buffer_store_b128_guarded).  The scanner flags the sequence in a listing -- not the guarded, the immediate-soffset or the
    unrelated ones -- and the production library does not contain it in any kernel."""
    import pytest
    import store_hazard_scan as H

    bad = ["0000 <_Zk1>:", "\tbuffer_store_dwordx4 v[106:109], v110, s[24:27], s66 offen", "\tv_pk_add_f32 v[106:107], v[142:143], v[200:201]", "\ts_endpgm"]
    n, hits = H.scan_listing(bad)
    assert n == 1 and len(hits) == 1 and hits[0][0] == "_Zk1" and hits[0][3] == 0
    one_slot = ["0000 <_Zk2>:", "\tbuffer_store_dwordx3 v[4:6], v9, s[0:3], s7 offen offset:16", "\tv_mov_b32_e32 v20, v21", "\tv_exp_f32_e32 v5, v30"]
    assert len(H.scan_listing(one_slot)[1]) == 1 and H.scan_listing(one_slot)[1][0][3] == 1
    fine = ["0000 <_Zk3>:",
            "\tbuffer_store_dwordx4 v[106:109], v110, s[24:27], s66 offen", "\ts_nop 1", "\tv_pk_add_f32 v[106:107], v[142:143], v[200:201]",   # guarded
            "\tbuffer_store_dwordx4 v[10:13], v14, s[24:27], 0 offen", "\tv_mov_b32_e32 v10, 0",                # immediate soffset: LLVM's own wait states
            "\tbuffer_store_dwordx4 v[20:23], v24, s[24:27], s5 offen", "\tv_mov_b32_e32 v30, 0", "\tv_mov_b32_e32 v31, 0", "\tv_mov_b32_e32 v20, 0",
            "\tbuffer_store_dwordx2 v[40:41], v24, s[24:27], s5 offen", "\tv_mov_b32_e32 v40, 0",              # 8 bytes: no hazard
            "\tbuffer_store_dwordx4 v[50:53], v54, s[24:27], s5 offen", "\tds_write_b64 v50, v[60:61]", "\tv_cmp_gt_u32_e32 vcc, v50, v51", "\ts_endpgm"]
    n, hits = H.scan_listing(fine)
    assert n == 3 and hits == []
    if not os.path.exists(H.OBJDUMP):
        pytest.skip("llvm-objdump not in this image")
    n, hits = H.scan_library()
    assert n > 0 and hits == [], hits
