"""The switch table (common.h `enum TuneKey`, model.hip `g_tune_table`) is indexed by the enum: the two lists must name the same switches
in the same order -- a row added to one and not the other would silently re-label every switch behind it."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_enum_and_table_name_the_same_switches_in_the_same_order():
    e = open(os.path.join(ROOT, "fastllm_amd", "csrc", "common.h")).read()
    t = open(os.path.join(ROOT, "fastllm_amd", "csrc", "model.hip")).read()
    keys = [k for k in re.findall(r"^\s*(TK_\w+)", re.search(r"enum TuneKey \{(.*?)\};", e, re.S).group(1), re.M) if k != "TK_COUNT"]
    names = re.findall(r'\{"(\w+)",', re.search(r"g_tune_table\[TK_COUNT\] = \{(.*?)\};", t, re.S).group(1))
    assert len(keys) == len(names) and len(keys) > 50
    assert [k[3:].lower() for k in keys] == names
    assert len(set(names)) == len(names)


def test_default_build_refuses_experimental_switches_and_timing_probes(monkeypatch):
    """CPU-only (fl_tune touches no GPU).  The default library does not contain the kernels that measured slower (decode engine, fused
    attention + o_proj, attention prefetch workgroups, loader waves): their switches are refused by fl_tune and ignored in the
    environment; the wrong-results timing probe behind bit 16 of "h4_pf"  cannot be switched on; the shipped sources of the
    default build carry no timing-experiment branch."""
    import glob
    import pytest
    import fastllm_amd as fa
    try:
        fa.tune("experimental", 0)
        pytest.skip("this is the EXPERIMENTAL build (FL_LIB_PATH)")
    except fa.FastLLMError as e:
        assert e.code == -10
    for key in ("engine", "fuse_oproj", "attn_prefetch", "skinny_loaders", "engine_grid", "ao_waves", "engine_timeout_ms", "debug_tp_loopback"):
        with pytest.raises(fa.FastLLMError) as e:
            fa.tune(key, 1)
        assert e.value.code == -10 and "EXPERIMENTAL" in str(e.value).upper(), (key, str(e.value))
    with pytest.raises(fa.FastLLMError) as e:
        fa.tune("no_such_switch", 1)
    assert e.value.code == -8
    monkeypatch.setenv("FL_ENGINE", "1")                 # (the conftest fixture re-reads the environment)
    monkeypatch.setenv("FL_H4_PF", str((1 << 16) + 6))
    fa.tune("h4_pf", (1 << 16) + 6)                      # accepted, bit 16 dropped: nothing to assert from here but that it is no error
    fa.tune("gemv_r", 2); fa.tune("gemv_u", 0)           # table rows now: "reload_env" reaches them
    fa.reload_env()
    srcs = glob.glob(os.path.join(ROOT, "fastllm_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "fastllm_amd", "csrc", "*.h"))
    assert not [s for s in srcs if "TIMING EXPERIMENT" in open(s).read()]
