"""Hand-issued loads (inline assembly the compiler's wait-count pass does not see) in k_pm_walk: the device assembly
must hold no read of such a load's target before a wait the load cannot have survived, and the ring registers the loads
land in must occur in hand-written assembly only (tools/isa_hazards.py; the first form of the kernel kept its in-flight
values in compiler-allocated registers, the compiler copied one at the loop's end -- before its load had landed -- and
cfg4 came out different from run to run)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_read_of_a_register_whose_load_is_in_flight():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_hazards.py"), "k_pm_walk"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]


def test_the_lint_sees_a_planted_hazard():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_hazards
    asm = """
_ZN4qmcp6k_fakeEv:
	;;#ASMSTART
	global_load_dword v100, v1, s[0:1]
	;;#ASMEND
	v_mov_b32_e32 v3, v100
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	s_endpgm
	.amdhsa_kernel _ZN4qmcp6k_fakeEv
"""
    found = isa_hazards.hazards(asm, "k_fake", ring_only_in_asm=True)
    assert len(found) >= 1
    ok = asm.replace("\tv_mov_b32_e32 v3, v100\n", "")
    assert isa_hazards.hazards(ok, "k_fake", ring_only_in_asm=True) == []
