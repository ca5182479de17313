"""Build-time check on the ISA of the one kernel whose correctness rests on how hipcc treats inline-asm load outputs
(ADVICE r3, low): `k_colgemm_h2q` (wavenet_amd/csrc/mfma_gemm_b3.hip) keeps its X ring as the OUTPUTS of inline-asm
`global_load_dwordx4` that are still in flight across loop iterations, waited for by a hand-counted `s_waitcnt vmcnt`.
hipcc believes an asm output is ready when the statement ends: if it ever copies or spills such a register between the
request and the wait, the copy reads a register the load has not written yet (exactly this miscompiled in
`k_wgrad_h2p`'s first form: intermittently wrong sums at full size).  The committed build is fine -- the ring sits in
fixed registers, never moved -- and this test keeps it so across compiler / flag changes: it compiles the file to gfx950
assembly (device only, ~35 s) and requires, in every instantiation of the kernel,
  * no scratch (spill) instruction at all,
  * every asm-issued ring load writes one of at most 8 fixed 4-register slots (a rotating ring would show more), and
  * with the vector-memory queue modelled (in-order retirement, `s_waitcnt vmcnt(N)` leaves the N youngest in flight; the
    pipelined region scanned twice for the loop-carried requests): no `v_mov_b32` / `v_accvgpr_write` / `v_pk_mov` /
    `v_swap` / `ds_write` / `global_store` reads a ring register whose load is still in the queue.
No GPU needed (hipcc cross-compiles)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "wavenet_amd", "csrc", "mfma_gemm_b3.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _regs(tok):
    """registers named by an operand token: v12 -> {12}; v[130:133] -> {130..133}"""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_the_pipelined_skip_gemm_never_moves_its_in_flight_ring_registers(tmp_path):
    out = tmp_path / "b3.s"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", str(out), SRC],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    text = out.read_text().splitlines()
    starts = [i for i, l in enumerate(text) if re.match(r"^_ZN2wn13k_colgemm_h2q\w+:", l)]
    assert len(starts) >= 3, "k_colgemm_h2q instantiations not found"
    for s0 in starts:
        end = next(i for i in range(s0, len(text)) if ".end_amdhsa_kernel" in text[i])
        body = text[s0:end]
        assert not any(re.search(r"\bscratch_(load|store)", l) for l in body), text[s0]
        ring, in_asm, slots, load_lines = set(), False, set(), []
        for i, l in enumerate(body):                     # pass 1: the ring = every destination of an asm-issued dwordx4 load
            ins = l.strip()
            if ins.startswith(";;#ASMSTART"):
                in_asm = True
            elif ins.startswith(";;#ASMEND"):
                in_asm = False
            elif in_asm and ins.startswith("global_load_dwordx4"):
                dst = _regs(re.split(r"[ ,]+", ins)[1])
                slots.add(min(dst))
                ring |= dst
                load_lines.append(i)
        assert 1 <= len(slots) <= 8 and len(ring) == 4 * len(slots), (sorted(slots), text[s0][:60])
        # pass 2: model the vector-memory queue (operations retire in issue order; `s_waitcnt vmcnt(N)` leaves the N youngest in
        # flight) over the pipelined part of the kernel -- first ring request to last -- twice, the second time starting
        # from the queue the first pass ended with (the loop-carried requests): no copy / spill / store may read a ring
        # register whose load is still in the queue
        def scan(lines, queue):
            for l in lines:
                ins = l.strip().split(";")[0].strip()
                if not ins or ins.startswith("."):
                    continue
                ops = [o for o in re.split(r"[ ,]+", ins) if o]
                if ops[0] == "s_waitcnt":
                    m = re.search(r"vmcnt\((\d+)\)", ins)
                    if m:
                        n = int(m.group(1))
                        queue = queue[len(queue) - n:] if n else []
                    continue
                if ops[0].startswith(("global_load", "global_store", "global_atomic", "buffer_", "scratch_")):
                    dst = _regs(ops[1]) & ring if ops[0].startswith("global_load_dword") else set()
                    queue = queue + [dst]
                    if ops[0].startswith(("global_store", "scratch_store")):
                        srcs = set().union(*[_regs(o) for o in ops[1:]])
                        busy = set().union(*queue[:-1]) if len(queue) > 1 else set()
                        assert not (srcs & busy), "a ring register in flight is stored: %s  [%s]" % (ins, text[s0][:60])
                    continue
                if ops[0].startswith(("v_mov_b32", "v_accvgpr_write", "v_pk_mov", "v_swap", "ds_write")):
                    srcs = set().union(*[_regs(o) for o in (ops[1:] if ops[0].startswith("ds_write") else ops[2:])])
                    busy = set().union(*queue) if queue else set()
                    assert not (srcs & busy), "a ring register in flight is copied: %s  [%s]" % (ins, text[s0][:60])
            return queue
        region = body[load_lines[0]:load_lines[-1] + 1]
        q1 = scan(region, [])
        scan(region, q1)


def test_every_inline_asm_store_carries_its_hazard_wait_states():
    """A store of more than 8 bytes reads its data registers late: gfx940+ needs two wait states before a VALU instruction
    overwrites them.  hipcc's hazard recogniser inserts them behind its own stores, never behind inline asm -- the bf16
    forward's ragged-tile path once computed an address into v[0:1] right behind an asm store of v[0:3] (NaNs in z).  Every
    asm store of the library therefore ends in `s_nop 1`; this keeps a new one from being added without it."""
    csrc = os.path.join(ROOT, "wavenet_amd", "csrc")
    found = 0
    for fn in sorted(os.listdir(csrc)):
        if not fn.endswith((".hip", ".hpp")):
            continue
        for ln, line in enumerate(open(os.path.join(csrc, fn)), 1):
            if "asm" in line and re.search(r"(global|buffer|flat)_store_dwordx[234]", line):
                found += 1
                assert re.search(r"s_nop\s+[1-9]", line), "%s:%d: asm store without its s_nop" % (fn, ln)
    assert found >= 2


def _kernel_disassembly(tmp_path):
    """{kernel symbol: [instruction text, ...]} of every gfx950 code object inside the built library (llvm-objdump extracts the
    offload bundles next to its input, so it works on a copy)."""
    import shutil
    from wavenet_amd import _lib
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(objdump) and os.path.exists(_lib.LIB_PATH)):
        pytest.skip("llvm-objdump or the built library is missing")
    lib = tmp_path / "lib.so"
    shutil.copy(_lib.LIB_PATH, lib)
    subprocess.run([objdump, "--offloading", str(lib)], capture_output=True, text=True, timeout=300)
    objs = [p for p in tmp_path.iterdir() if "gfx950" in p.name]
    assert objs, "no gfx950 code object extracted from the library"
    kernels, cur = {}, None
    for co in objs:
        r = subprocess.run([objdump, "-d", "--no-show-raw-insn", str(co)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-500:]
        for line in r.stdout.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\w+)>:", line)
            if m:
                cur = kernels.setdefault(m.group(1), [])
            elif cur is not None and line.startswith("\t"):
                cur.append(line.split("//")[0].strip())
    return kernels


def test_no_vector_memory_instruction_reads_a_scalar_a_vector_instruction_has_just_written(tmp_path):
    """gfx9 hazard: a VALU instruction that writes an SGPR (v_readlane / v_readfirstlane / a compare / a carry-out) must be
    five wait states ahead of a vector-memory instruction that reads that SGPR (its `saddr` base).  hipcc keeps that
    distance for its own instructions and cannot see inside inline asm; the decoder's chain loads, the bf16 forward's requests
    and stores and the pipelined GEMMs' requests are inline asm with an SGPR base.  Checked on the disassembly of the library
    as built: every vector-memory instruction with an s[a:b] operand, in every kernel."""
    kernels = _kernel_disassembly(tmp_path)
    assert any("k_decode_fast3" in k for k in kernels) and any("k16_fwd" in k for k in kernels)
    checked = 0
    for name, ins in kernels.items():
        for i, text in enumerate(ins):
            if not re.match(r"(global|buffer|scratch)_(load|store|atomic)", text):
                continue
            m = re.search(r"\bs\[(\d+):(\d+)\]", text)
            if not m:
                continue
            base = set(range(int(m.group(1)), int(m.group(2)) + 1))
            checked += 1
            states, k = 0, i - 1
            while k >= 0 and states < 5:
                prev = ins[k]
                if prev.startswith("v_"):
                    ops = re.split(r",\s*", prev.split(None, 1)[1]) if " " in prev else []
                    # scalar destinations of a VALU instruction: operand 0 (v_readlane, compares), and operand 1 of the
                    # instructions with a carry-out / scale flag
                    dsts = ops[:2] if prev.startswith(("v_add_co", "v_sub_co", "v_addc_co", "v_subb_co", "v_subrev_co",
                                                       "v_div_scale", "v_mad_u64", "v_mad_i64")) else ops[:1]
                    written = set()
                    for d in dsts:
                        m1, m2 = re.fullmatch(r"s(\d+)", d), re.fullmatch(r"s\[(\d+):(\d+)\]", d)
                        if m1:
                            written.add(int(m1.group(1)))
                        elif m2:
                            written |= set(range(int(m2.group(1)), int(m2.group(2)) + 1))
                    assert not (written & base), "%s: `%s` reads s%s %d wait state(s) after `%s`" % (
                        name, text, sorted(base), states, prev)
                nop = re.match(r"s_nop (\d+)", prev)
                states += int(nop.group(1)) + 1 if nop else 1
                k -= 1
    assert checked > 100


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("src,kernels", [
    ("w16_gemm.hip", ("k16_wgrad", "k16_cgemm256")),
    ("w16_layer.hip", ("k16_fwd", "k16_gate_bwd", "k16_dx")),
])
def test_bf16_stage_loops_carry_no_wait_of_the_compilers(src, kernels):
    """Round 5 ("waits the compiler adds"): with a builtin LDS-DMA in a loop hipcc puts `s_waitcnt vmcnt(0)` in front
    of the next LDS read that may alias it -- in k16_wgrad that stood at the top of the stage loop, right behind the requests of
    the stage after next (5 us per 64 KB stage instead of 1), in k16_dx in front of the dWp reads.  These kernels request through
    inline asm and wait by their own counts; this keeps a builtin request (or a plain load consumed inside the loop) from coming
    back: behind the kernel's first request, no block of a loop holds a vmcnt wait that is not the kernel's own."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_waits
    text = isa_waits.assembly(os.path.join(ROOT, "wavenet_amd", "csrc", src))
    waits = isa_waits.compiler_waits_in_loops(text)
    seen = 0
    for sym, ws in waits.items():
        if not any(("%d%s" % (len(k), k)) in sym for k in kernels):
            continue
        seen += 1
        s0 = next(i for i, l in enumerate(text) if l.startswith(sym + ":"))
        first_req = next((i + 1 for i in range(s0, len(text)) if "global_load_lds" in text[i]), None)
        assert first_req is not None, sym
        late = [w for w in ws if w[0] > first_req]
        assert not late, "%s: compiler-made vmcnt wait(s) inside a loop behind the first request: %s" % (sym, late)
    assert seen >= len(kernels)
