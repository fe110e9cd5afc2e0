"""Build-time ISA check of the sweeps whose loads are issued by hand (tools/isa_check.py): the register ring
of a load in flight is touched by nothing, and nothing in the loop drains the queue.  Cross-compiles; no GPU."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mod():
    spec = importlib.util.spec_from_file_location("isa_check", os.path.join(ROOT, "tools", "isa_check.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_the_checker_sees_a_use_before_the_wait_and_a_drain():
    m = _mod()
    good = """.LBB0_1:
	global_load_dwordx2 v[10:11], v2, s[4:5] nt
	global_load_dwordx2 v[12:13], v2, s[6:7] nt
	s_waitcnt vmcnt(2)
	v_add_f64 v[20:21], v[14:15], v[16:17]
	global_load_dwordx2 v[14:15], v2, s[8:9] nt
	global_load_dwordx2 v[16:17], v2, s[10:11] nt
	s_waitcnt vmcnt(2)
	v_add_f64 v[22:23], v[10:11], v[12:13]
	s_cbranch_scc1 .LBB0_1
	s_endpgm""".splitlines()
    errs, _ = m.check("k", good)
    assert errs == [], errs
    early = [l.replace("v_add_f64 v[22:23], v[10:11], v[12:13]", "v_add_f64 v[22:23], v[14:15], v[12:13]") for l in good]
    errs, _ = m.check("k", early)
    assert any("while a load into it is in flight" in e for e in errs), errs
    drain = [l.replace("s_waitcnt vmcnt(2)\n", "") for l in good]
    drain[3] = "\ts_waitcnt vmcnt(0)"
    errs, _ = m.check("k", drain)
    assert any("vmcnt(0)" in e for e in errs), errs


@pytest.mark.skipif(shutil.which("/opt/rocm/bin/hipcc") is None, reason="no hipcc")
def test_default_path_sweeps_pass_the_isa_check():
    assert _mod().main() == 0
