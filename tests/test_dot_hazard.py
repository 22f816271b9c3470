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


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain not found")
@pytest.mark.parametrize("kernel,min_mfma,min_found", [("gemm_4w_kernel", 2 * 256, 2), ("gemm_h4_kernel", 4 * 64, 4), ("gemm_w14_kernel", 2 * 112, 1)])
def test_four_wave_gemm_owns_its_accumulator_registers(tmp_path, kernel, min_mfma, min_found):
    """k_gemm_8p.hip's gemm_4w_kernel keeps its 64 accumulator tiles in a[0:255] through inline asm (k_gemm_h4.hip's gemm_h4_kernel its
    32 in a[0:127]); hipcc does not know.  The shipped code must therefore hold no scratch access (a spill could land in those
    registers' shadow), no compiler copy INTO an accumulator register (v_accvgpr_write from a VGPR, v_accvgpr_mov), and every read of
    the accumulators after an MFMA must come behind the kernel's own s_nop padding (>= 16 wait states: the 4-pass MFMA's result latency)."""
    blob = open(LIB, "rb").read()
    found = 0
    for n, co in enumerate(gfx950_code_objects(blob)):
        p = tmp_path / ("g%d.elf" % n)
        p.write_bytes(co)
        txt = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", str(p)], capture_output=True, text=True, check=True).stdout
        if kernel not in txt:
            continue
        func, since_mfma, nmfma, recent = None, None, 0, []
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                func = m.group(1) if kernel in m.group(1) else None
                if func:
                    found += 1
                since_mfma = None
                continue
            m = re.match(r"^\s+(\S+)\s*(.*?)\s*//", line)
            if not func or not m:
                continue
            op, args = m.group(1), m.group(2)
            if op.startswith("v_") and not op.startswith(("v_mfma", "v_accvgpr", "v_cmp", "v_readlane", "v_readfirstlane")):
                d = re.match(r"v\[(\d+):(\d+)\]|v(\d+)", args)
                recent = (recent + [(op, (int(d.group(1)), int(d.group(2))) if d and d.group(1) else (int(d.group(3)), int(d.group(3))) if d else None)])[-2:]
            elif not op.startswith("v_mfma"):
                recent = (recent + [(op, None)])[-2:]
            assert not op.startswith("scratch_"), "%s: scratch access\n%s" % (func, line)
            assert not op.startswith("v_accvgpr_mov"), "%s: compiler copy between accumulator registers\n%s" % (func, line)
            if op.startswith("v_accvgpr_write"):
                assert re.search(r",\s*0$", args), "%s: a VGPR written into an accumulator register\n%s" % (func, line)
            if op.startswith("v_mfma"):
                assert re.match(r"a\[\d+:\d+\]", args), "%s: MFMA outside the accumulator file\n%s" % (func, line)
                # its A / B operands arrive by ds_read (hipcc waits for those); a VALU instruction writing one of them right in
                # front of the asm MFMA would need wait states hipcc does not know to insert
                srcs = [(int(a), int(b)) for a, b in re.findall(r"v\[(\d+):(\d+)\]", args)]
                for pop, pdst in recent:
                    for lo, hi in srcs:
                        assert not (pdst and lo <= pdst[1] and pdst[0] <= hi), "%s: %s writes an MFMA operand two slots ahead\n%s" % (func, pop, line)
                since_mfma, nmfma = 0, nmfma + 1
                recent = (recent + [(op, None)])[-2:]
            elif since_mfma is not None:
                if op.startswith("s_nop"):
                    since_mfma += int(args.split()[0], 0) + 1
                elif op.startswith("v_accvgpr_read"):
                    assert since_mfma >= 16, "%s: accumulator read %d wait states behind an MFMA\n%s" % (func, since_mfma, line)
                    since_mfma = None                                  # the rest of this epilogue is covered
                else:
                    since_mfma += 1
        assert nmfma >= min_mfma, "expected the instantiations' K loops, found %d MFMAs" % nmfma
    assert found >= min_found, "%s not found in the library" % kernel


READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


@pytest.mark.skipif(not os.path.exists(READELF), reason="llvm-readelf of the ROCm toolchain not found")
def test_no_dispatched_kernel_spills(tmp_path):
    """No kernel the default dispatch can launch holds spilled VGPRs (a spill turns a weight stream into scratch traffic: the fp32
    mode's QKV GEMV ran at 1.4 TB/s instead of 4.5 with one, round 4).  Exempt: instantiations only a tuning switch reaches -- the fp32
    GEMV with 4 / 7 / 8 chunks per row or four rows per wave (FL_GEMV_U / FL_GEMV_R; fp32 defaults to two chunks, two rows) and the
    32-row prefill attention for groups of five query heads (no model here has them)."""
    blob = open(LIB, "rb").read()
    bad, seen = [], 0
    for n, co in enumerate(gfx950_code_objects(blob)):
        p = tmp_path / ("n%d.elf" % n)
        p.write_bytes(co)
        md = subprocess.run([READELF, "--notes", str(p)], capture_output=True, text=True, check=True).stdout
        for blk in md.split(".args:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            spill = re.search(r"\.vgpr_spill_count:\s+(\d+)", blk)
            if not name or not spill:
                continue
            seen += 1
            if int(spill.group(1)) == 0:
                continue
            k = name.group(1)
            if re.search(r"gemv_kernelIffLi(2ELi[478]|4ELi2)E", k) or "attn_prefill32_kernelILi128ELi5ELi1E" in k:
                continue
            bad.append("%s: %s VGPRs spilled" % (k, spill.group(1)))
    assert seen > 100, "kernel metadata not found (%d entries)" % seen
    assert not bad, "\n".join(bad)
