"""tools/tap8_pipeline_scan.py (build gate, csrc/build.sh) on synthetic assembly: the shapes it must accept and the failures it
exists for -- a load sunk into a branch some waves skip (the round-4 race, profiles/r4_tapgemm8.md section 2.1), an LDS-DMA request behind
the activation loads, a barrier with no covering wait, a wait counting the wrong number of loads -- and, when the in-tree build's
assembly is present, the real kernels (0 findings)."""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import tap8_pipeline_scan as S  # noqa: E402

NAME = "_ZN2ac16tap_gemm8_kernelILi2ELi4ELi2ELi2ELb0ELb0EEEvNS_13TapGemmParamsEPKDF16b"      # 128-row tile: A_SLOTS = 3
DMA = "\t;;#ASMSTART\n\ts_mov_b32 s1, m0\n\ts_mov_b32 m0, s0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 v[6:7], off\n\ts_mov_b32 m0, s1\n\t;;#ASMEND\n"
LOAD = "\tbuffer_load_dwordx4 v[{0}:{1}], v9, s[44:47], 0 offen\n"
WAIT = "\t;;#ASMSTART\n\ts_waitcnt vmcnt({0})\n\t;;#ASMEND\n\t;;#ASMSTART\n\ts_waitcnt lgkmcnt(0)\n\t;;#ASMEND\n\ts_barrier\n"
MFMA = "\tv_mfma_f32_32x32x16_f16 v[100:115], v[20:23], v[24:27], v[100:115]\n"


def loads(n, first=10):
    return "".join(LOAD.format(first + 4 * i, first + 4 * i + 3) for i in range(n))


def kernel(body):
    return f"\t.text\n{NAME}:\n\ts_load_dwordx4 s[44:47], s[0:1], 0x0\n{body}\ts_endpgm\n.Lfunc_end0:\n"


def run(tmp_path, body):
    p = tmp_path / "k.s"
    p.write_text(kernel(body))
    ks, labs = S.parse(str(p))
    assert list(ks) == [NAME]
    return S.scan_kernel(NAME, ks[NAME], labs[NAME])


def stage(dma=2, arms=None, wait=3):
    """One main-loop stage: LDS-DMA requests, the activation loads (`arms`: text, default 3 plain loads), MFMAs, wait + barrier."""
    return DMA * dma + (arms if arms is not None else loads(3)) + MFMA * 4 + WAIT.format(wait)


def loop(body):
    return DMA + loads(3) + WAIT.format(3) + ".LBB0_1:\n" + body + "\ts_cmp_eq_u32 s5, 0\n\ts_cbranch_scc0 .LBB0_1\n"


def test_accepts_the_intended_pipeline(tmp_path):
    assert run(tmp_path, loop(stage())) == []


def test_accepts_two_exclusive_arms_of_one_if(tmp_path):
    # hipcc's structurised `if (interior) loads A else loads B`: a flag register carries the condition to the second arm
    arms = ("\ts_mov_b64 s[0:1], -1\n\ts_and_b64 vcc, exec, s[22:23]\n\ts_cbranch_vccnz .LBB0_5\n" + loads(3) +
            "\ts_mov_b64 s[0:1], 0\n.LBB0_5:\n\ts_and_b64 vcc, exec, s[0:1]\n\ts_cbranch_vccz .LBB0_6\n" + loads(3) + ".LBB0_6:\n")
    assert run(tmp_path, loop(stage(arms=arms))) == []


def test_rejects_an_if_whose_arms_can_both_be_skipped(tmp_path):
    # the same shape WITHOUT the flag: nothing says the second arm runs when the first did not
    arms = ("\ts_and_b64 vcc, exec, s[22:23]\n\ts_cbranch_vccnz .LBB0_5\n" + loads(3) +
            ".LBB0_5:\n\ts_and_b64 vcc, exec, s[30:31]\n\ts_cbranch_vccz .LBB0_6\n" + loads(3) + ".LBB0_6:\n")
    f = run(tmp_path, loop(stage(arms=arms)))
    assert f and "LDS-DMA request" in f[0]


def test_rejects_a_load_sunk_into_a_branch_some_waves_skip(tmp_path):
    # round 4's race: the last slot's load moved into `if (row < A_ROWS)` -- only one wave issues it
    arms = loads(2) + "\ts_and_saveexec_b64 s[2:3], s[10:11]\n\ts_cbranch_execz .LBB0_7\n" + loads(1, 30) + ".LBB0_7:\n\ts_or_b64 exec, exec, s[2:3]\n"
    f = run(tmp_path, loop(stage(arms=arms)))
    assert f and "among the 3 youngest" in f[0]


def test_rejects_an_lds_dma_request_behind_the_loads(tmp_path):
    f = run(tmp_path, loop(DMA + loads(3) + DMA + MFMA * 4 + WAIT.format(3)))
    assert f and "LDS-DMA request" in f[0]


def test_rejects_a_wait_that_counts_the_wrong_number(tmp_path):
    f = run(tmp_path, loop(stage(wait=2)))
    assert f and "A_SLOTS = 3" in f[0]


def test_rejects_a_wait_that_lost_its_barrier(tmp_path):
    body = loop(DMA * 2 + loads(3) + MFMA * 4 + "\t;;#ASMSTART\n\ts_waitcnt vmcnt(3)\n\t;;#ASMEND\n\tds_read_b128 v[20:23], v5\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n")
    f = run(tmp_path, body)
    assert any("not followed by its barrier" in x for x in f), f


def test_rejects_requests_in_flight_at_the_end_of_the_kernel(tmp_path):
    # the last stage still issues weight requests and nothing drains them before the (LDS-reusing) epilogue / s_endpgm
    body = loop(stage()) + DMA + MFMA * 4 + "\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n"
    f = run(tmp_path, body)
    assert any("s_endpgm" in x and "no covering wait" in x for x in f), f


def test_accepts_requests_in_flight_across_segment_barriers(tmp_path):
    # the ping-pong loop: requests at the top of a stage, two segment barriers, THEN the counted wait and its barrier
    seg = "\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n"
    body = loop(DMA * 2 + loads(3) + seg + MFMA * 4 + "\ts_barrier\n" + MFMA * 0 + WAIT.format(3) + MFMA * 4 + "\ts_barrier\n")
    assert run(tmp_path, body) == []


def test_other_vector_memory_traffic_in_the_window_is_rejected(tmp_path):
    f = run(tmp_path, loop(DMA * 2 + loads(2) + "\tglobal_load_dword v40, v[2:3], off\n" + loads(1, 30) + MFMA + WAIT.format(3)))
    assert f and "global_load_dword" in f[0]


def test_the_in_tree_build_is_clean():
    asm = glob.glob(os.path.join(ROOT, "audiocodecs_amd", "csrc", "build", "*-hip-amdgcn-amd-amdhsa-gfx950.s"))
    if not asm:
        import pytest
        pytest.skip("no in-tree build directory (csrc/build.sh has not run here)")
    total, findings = 0, []
    for p in asm:
        ks, labs = S.parse(p)
        for n, ins in ks.items():
            total += 1
            findings += S.scan_kernel(n, ins, labs[n])
    assert total >= 12 and findings == []
