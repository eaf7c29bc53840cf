"""tools/loop_waits.py (the report that found the staged epilogue's latent waits in round 5): a loop that issues loads and waits with
vmcnt(0) is listed with its counts; a loop without vector-memory waits and code outside loops are not."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ASM = """
\t.text
_ZN2ac6kernelAEv:                        ; @_ZN2ac6kernelAEv
\ts_load_dwordx2 s[0:1], s[4:5], 0x0
\ts_waitcnt vmcnt(0)
.LBB0_1:                                ; =>This Inner Loop Header: Depth=1
\tglobal_load_dwordx4 v[0:3], v[4:5], off
\ts_waitcnt vmcnt(0)
\tglobal_store_dwordx4 v[4:5], v[0:3], off
\ts_cbranch_scc1 .LBB0_1
.LBB0_2:                                ; %exit
\ts_endpgm
_ZN2ac6kernelBEv:                        ; @_ZN2ac6kernelBEv
.LBB1_1:                                ; =>This Inner Loop Header: Depth=1
\tv_add_f32_e32 v0, v0, v1
\ts_cbranch_scc1 .LBB1_1
\ts_endpgm
_ZN2ac6kernelCEv:                        ; @_ZN2ac6kernelCEv
.LBB2_1:                                ; =>This Inner Loop Header: Depth=1
\tbuffer_load_dwordx4 v[0:3], v4, s[0:3], 0 offen
.LBB2_2:                                ;   in Loop: Header=BB2_1 Depth=1
\ts_waitcnt vmcnt(3)
\tglobal_atomic_umax v[4:5], v0, off
\ts_cbranch_scc1 .LBB2_1
\ts_endpgm
"""


def run(tmp_path, *pats):
    f = tmp_path / "k.s"
    f.write_text(ASM)
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "loop_waits.py"), str(f), *pats], capture_output=True, text=True, check=True).stdout


def test_lists_loops_with_vector_memory_waits(tmp_path):
    out = run(tmp_path)
    assert "_ZN2ac6kernelAEv" in out and "load=1" in out and "store=1" in out and "vmcnt(0)=1" in out
    assert "_ZN2ac6kernelBEv" not in out                      # a loop without waits
    assert "_ZN2ac6kernelCEv" in out and "vmcnt(3)=1" in out and "atomic=1" in out
    assert out.count("vmcnt(0)=1") == 1                       # the wait in front of kernel A's loop is not in a loop


def test_kernel_filter(tmp_path):
    out = run(tmp_path, "kernelC")
    assert "kernelC" in out and "kernelA" not in out
