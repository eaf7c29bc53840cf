"""tools/mfma_branch_hazard.py (run by csrc/build.sh over the device assembly of every translation unit): the sequence hipcc produced
in round 4 -- an MFMA, a taken branch, a read of the MFMA's result one instruction later -- is flagged; the same read behind enough
wait states, behind independent MFMAs, as the next MFMA's accumulator input, or on a path cut off by an unconditional branch is not."""
import os
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _scan(tmp_path, body):
    import mfma_branch_hazard as H

    p = tmp_path / "k.s"
    p.write_text("_ZN2ac6kernelEv:\n" + textwrap.dedent(body))
    return H.scan(str(p))


def test_the_round4_sequence_is_flagged(tmp_path):
    hits = _scan(tmp_path, """
        v_mfma_f32_16x16x32_f16 a[0:3], v[70:73], v[190:193], a[0:3]
        s_cbranch_vccnz .LBB0_25
        global_load_dwordx4 v[166:169], v[126:127], off
        s_branch .LBB0_26
    .LBB0_25:
        v_mov_b32_e32 v194, v124
    .LBB0_26:
        v_accvgpr_read_b32 v129, a3
        s_endpgm
    """)
    assert len(hits) == 1 and "v_accvgpr_read_b32 v129, a3" in hits[0][3]


def test_enough_wait_states_or_independent_work_pass(tmp_path):
    assert not _scan(tmp_path, """
        v_mfma_f32_16x16x32_f16 a[0:3], v[70:73], v[190:193], a[0:3]
        s_cbranch_vccnz .LBB0_25
    .LBB0_25:
        s_nop 4
        v_accvgpr_read_b32 v129, a3
        s_endpgm
    """)
    assert not _scan(tmp_path, """
        v_mfma_f32_16x16x32_f16 a[0:3], v[126:129], a[40:43], a[0:3]
        v_mfma_f32_16x16x32_f16 a[4:7], v[158:161], a[40:43], a[4:7]
        v_mfma_f32_16x16x32_f16 a[8:11], v[190:193], a[40:43], a[8:11]
        s_cbranch_vccnz .LBB0_55
    .LBB0_55:
        v_accvgpr_read_b32 v33, a3
        s_endpgm
    """)


def test_accumulation_and_dead_paths_pass(tmp_path):
    assert not _scan(tmp_path, """
        v_mfma_f32_16x16x4_f32 a[12:15], v81, v77, a[12:15]
        s_cbranch_vccnz .LBB0_9
    .LBB0_9:
        v_mfma_f32_16x16x4_f32 a[12:15], v26, v18, a[12:15]
        s_endpgm
    """)
    assert not _scan(tmp_path, """
        v_mfma_f32_16x16x4_f32 a[0:3], v165, v129, a[20:23]
        s_cbranch_execnz .LBB0_110
        s_branch .LBB0_213
    .LBB0_108:
        v_accvgpr_mov_b32 a1, a0
    .LBB0_110:
        s_nop 7
        s_nop 1
    .LBB0_213:
        s_endpgm
    """)
