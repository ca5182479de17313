#!/usr/bin/env python3
"""Where did hipcc put a wait of its own?  (diagnostic; tests/test_isa_cpu.py uses `compiler_waits_in_loops`)

hipcc's wait-count pass knows nothing of the memory operations inside inline asm and everything about its own: a builtin
LDS-DMA (`__builtin_amdgcn_global_load_lds`) is a pending LDS write to it, and the next LDS access that "may alias" gets
`s_waitcnt vmcnt(0)` in front -- which also waits for every request and store the kernel meant to leave in flight.  The
same happens at the header of a spin loop whose compare may be fed by the poll of the back edge, and at the next write of
a register a polled load may still be writing.  Round 5 found four hot loops serialised that way ("waits the
compiler adds"); this script is how:

    python tools/isa_waits.py scan  wavenet_amd/csrc/w16_gemm.hip            # compiler-made vmcnt waits inside loops, per kernel
    python tools/isa_waits.py trace wavenet_amd/csrc/w16_gemm.hip k16_wgradILb0  # the kernel's memory operations, waits and barriers in order
                                                                              # (lower case + '*': inside inline asm)
Compiles the file to gfx950 assembly first (device only); a `.s` file is taken as it is."""
import os
import re
import subprocess
import sys
import tempfile

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def assembly(path):
    if path.endswith(".s"):
        return open(path).read().split("\n")
    out = os.path.join(tempfile.mkdtemp(), "k.s")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-inline-asm", "-I", os.path.join(ROOT, "include"),
                        "-S", "--cuda-device-only", "-o", out, path], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-3000:])
    return open(out).read().split("\n")


_LOOP = re.compile(r"^\.LBB\d+_\d+:.*(Loop Header|in Loop|Inner Loop)")


def compiler_waits_in_loops(lines):
    """{kernel symbol: [(line, N), ...]}: every `s_waitcnt vmcnt(N)` outside ;;#ASMSTART/;;#ASMEND in a block hipcc marks as part of a loop"""
    out, name, in_asm, in_loop = {}, None, False, False
    for ln, line in enumerate(lines, 1):
        if re.match(r"^_Z\w+:", line):
            name, in_loop = line.split(":")[0], False
            out[name] = []
            continue
        if name is None:
            continue
        if "#ASMSTART" in line:
            in_asm = True
        elif "#ASMEND" in line:
            in_asm = False
        elif re.match(r"^\.LBB", line):
            in_loop = bool(_LOOP.match(line))
        elif ".end_amdhsa_kernel" in line:
            name = None
        else:
            m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", line)
            if m and not in_asm and in_loop:
                out[name].append((ln, int(m.group(1))))
    return out


_CATS = [("DMA", r"global_load_lds"), ("GLOAD", r"global_load_dword|buffer_load|flat_load"), ("GSTORE", r"global_store|buffer_store"),
         ("GATOM", r"global_atomic"), ("SCRATCH", r"scratch_"), ("DSR", r"ds_read"), ("DSW", r"ds_write"), ("MFMA", r"v_mfma"),
         ("BAR", r"s_barrier"), ("WAIT", r"s_waitcnt")]


def trace(lines, sub):
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and sub in l)
    end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i])
    prev, cnt, first, in_asm = None, 0, 0, False
    for ln in range(start + 1, end):
        line = lines[ln]
        if "#ASMSTART" in line:
            in_asm = True
            continue
        if "#ASMEND" in line:
            in_asm = False
            continue
        body = line.split(";")[0]
        tag = next((c for c, r in _CATS if re.search(r, body)), None)
        if tag is None:
            continue
        if tag == "WAIT":
            tag = ("wait* " if in_asm else "WAIT  ") + body.strip().replace("s_waitcnt ", "")
        elif in_asm:
            tag = tag.lower() + "*"
        if tag == prev:
            cnt += 1
        else:
            if prev:
                print("%7d  %s x%d" % (first, prev, cnt))
            prev, cnt, first = tag, 1, ln + 1
    if prev:
        print("%7d  %s x%d" % (first, prev, cnt))


if __name__ == "__main__":
    if len(sys.argv) < 3 or sys.argv[1] not in ("scan", "trace"):
        sys.exit(__doc__)
    text = assembly(sys.argv[2])
    if sys.argv[1] == "scan":
        for k, v in compiler_waits_in_loops(text).items():
            if v:
                print(k[:100], v[:16])
    else:
        trace(text, sys.argv[3])
