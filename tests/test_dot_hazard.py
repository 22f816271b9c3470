"""The decode GEMV's arithmetic is `v_dot2c_f32_bf16` through inline asm (hipcc 7.2 cannot select a builtin for it).
gfx940+ lets an instruction that is not the same dot touch a dot's result only three wait states later; hipcc inserts
such waits for its own instructions but cannot see into inline asm.  The kernels end every run of dots with an `s_nop 2`
tied to the accumulator -- this test checks the SHIPPED code objects instruction by instruction, so that a compiler that
one day puts a register copy, a spill or the wave reduction right behind a dot fails here and not as a wrong sum."""
import os
import re
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "fastllm_amd", "lib", "libfastllm_mi355x.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def gfx950_code_objects(blob):
    """every gfx950 entry of every offload bundle in the library's .hip_fatbin"""
    out, i = [], blob.find(MAGIC)
    while i >= 0:
        n = struct.unpack_from("<Q", blob, i + 24)[0]
        off = i + 32
        for _ in range(n):
            o, s, tl = struct.unpack_from("<QQQ", blob, off)
            off += 24
            triple = blob[off:off + tl].decode()
            off += tl
            if "gfx950" in triple and s:
                out.append(blob[i + o:i + o + s])
        i = blob.find(MAGIC, i + 1)
    return out


REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def touches(operands, d):
    for m in REG.finditer(operands):
        if m.group(1) is not None:
            if int(m.group(1)) == d:
                return True
        elif int(m.group(2)) <= d <= int(m.group(3)):
            return True
    return False


def hazards(disasm):
    """[(function, line)] where something other than the accumulating dot touches a dot's result inside three wait states"""
    insns, func = [], None
    for line in disasm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            func = m.group(1)
            insns.append((func, None, None, line))                   # function boundary
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//", line)
        if m:
            insns.append((func, m.group(1), m.group(2), line))
    bad, ndots = [], 0
    for k, (fn, op, args, line) in enumerate(insns):
        if not op or not op.startswith("v_dot2c_f32_bf16"):
            continue
        ndots += 1
        d = int(re.match(r"v(\d+)", args).group(1))
        wait = 0
        for fn2, op2, args2, line2 in insns[k + 1:]:
            if wait >= 3 or op2 is None or op2 == "s_endpgm":
                break
            if op2.startswith("s_nop"):
                wait += int(args2.split()[0], 0) + 1
                continue
            if op2.startswith("v_dot2c_f32_bf16"):
                dst, srcs = args2.split(",", 1)
                if touches(srcs, d):
                    bad.append((fn, line.strip(), line2.strip()))     # a dot result as the next dot's multiplicand
                if touches(dst, d):
                    break                                            # the next link of the accumulation chain: its own window
                wait += 1
                continue
            if op2.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc")):
                bad.append((fn, line.strip(), line2.strip()))         # control flow inside the window: not provable here
                break
            if touches(args2, d):
                bad.append((fn, line.strip(), line2.strip()))
                break
            wait += 1
    return bad, ndots


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain not found")
def test_no_instruction_touches_a_dot_result_too_early(tmp_path):
    blob = open(LIB, "rb").read()
    cos = gfx950_code_objects(blob)
    assert cos, "no gfx950 code object in the library"
    total, bad = 0, []
    for n, co in enumerate(cos):
        if b"v_dot2c" not in co and n >= 0:
            pass                                                     # (mnemonics are not in the binary: every object is disassembled)
        p = tmp_path / ("co%d.elf" % n)
        p.write_bytes(co)
        txt = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", str(p)], capture_output=True, text=True, check=True).stdout
        if "v_dot2c_f32_bf16" not in txt:
            continue
        b, nd = hazards(txt)
        total += nd
        bad += b
    assert total > 1000, "expected the GEMV kernels' dots in the library, found %d" % total
    assert not bad, "dot results touched inside three wait states:\n" + "\n".join("%s\n    %s\n    %s" % x for x in bad[:20])
