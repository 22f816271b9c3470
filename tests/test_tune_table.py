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
