"""Checks on the device code hipcc generates for libggq (gfx950 ISA text and the compiler's resource remarks).

The library's bit-exactness and part of its speed rest on properties of the generated code that no source-level
test sees:
  * no packed-f32 arithmetic (v_pk_add/mul/fma_f32) and no mixed-precision fused multiply-add (v_fma_mix*): the
    first has given wrong results on these machines, the second turns a -0.0 product into +0.0 and fuses a
    multiply with an add that the reference (ggml-quants/src/structs/q4_1.rs:49-60 and every other decoder)
    rounds separately;
  * fused multiply-adds only inside the compiler's correctly rounded division sequence (v_div_scale ...
    v_div_fixup) of the quantizers, where the reference divides too (ggml-quants/src/structs/q4_0.rs:31);
  * the group start of the single-tensor decode kernels: a scalar one-dword load of the destination that is
    issued, waited for and folded into the source address before the first tile load;
  * no scratch memory, 8 waves per SIMD and exactly the LDS bytes of the tile geometry for every decode kernel.
This module is the parser and the checks; tests/test_codegen.py runs them on the shipped flags and on two
sabotaged flag sets.  CPU only: hipcc cross-compiles.
"""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gguf_amd", "csrc")


def compile_device(out_s, flags_override=None, timeout=600):
    """`make asm` of gguf_amd/csrc into out_s with the compiler's kernel-resource-usage remarks; returns the remarks
    text.  flags_override replaces the Makefile's FLAGS (to show that the checks notice a missing flag)."""
    cmd = ["make", "-s", "-C", CSRC, "asm", f"ASM_OUT={out_s}", "EXTRA=-Rpass-analysis=kernel-resource-usage"]
    if flags_override is not None:
        cmd.append(f"FLAGS={flags_override}")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    if r.returncode != 0:
        raise RuntimeError(f"make asm failed:\n{r.stderr[-2000:]}")
    return r.stderr


def makefile_flags():
    """The FLAGS the Makefile builds with, as one string (make -p expands them)."""
    r = subprocess.run(["make", "-C", CSRC, "-pn", "asm"], capture_output=True, text=True, timeout=60)
    for line in r.stdout.splitlines():
        if line.startswith("FLAGS := ") or line.startswith("FLAGS = "):
            return line.split("=", 1)[1].strip()
    raise RuntimeError("FLAGS not found in make -p output")


def functions(asm_text):
    """{mangled name: [instruction lines]} for every function of the assembly text."""
    out, name, body = {}, None, []
    for line in asm_text.splitlines():
        m = re.match(r"^(_Z[\w$.]+):", line)
        if m and name is None:
            name, body = m.group(1), []
            continue
        if name is not None:
            if re.match(r"^\.Lfunc_end\d+:", line):
                out[name] = body
                name = None
                continue
            s = line.strip()
            if s and not s.startswith((";", ".", "//")) and not s.endswith(":"):
                ins = s.split(";")[0].strip()
                # the quantizers' short division sequences carry a tag in the asm comment
                body.append(ins + " ; ggq:fastdiv" if "ggq:fastdiv" in s else ins)
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, r.stdout.splitlines()))


FORBIDDEN = re.compile(r"^(v_pk_(add|mul|fma)_f32|v_fma_mix\w*)\b")
FMA = re.compile(r"^(v_fma_f32|v_fmac_f32|v_fma_f64|v_fmac_f64|v_pk_fma_f16|v_fma_f16|v_fmac_f16)\b")


# fused steps of the short divisions per encode call (gguf_amd/csrc/quantize.h: div_const_fast 2, recip_fast 4);
# Q4_0 and Q5_0 divide by a power of two first, which is a multiplication
FASTDIV_STEPS = {"EncQ4_0": 4, "EncQ5_0": 4, "EncQ4_1": 6, "EncQ5_1": 6, "EncQ8_0": 6, "EncQ8_1": 6, "EncQ8K": 6}


def check_arithmetic(funcs, names=None):
    """Returns a list of complaints (empty = fine).  Fused multiply-adds may appear in two places only: inside the
    compiler's division sequence, and as the tagged steps of the quantizers' short divisions -- there exactly
    FASTDIV_STEPS per encode call (one or two calls per thread), and only in quantizer kernels and the self-test."""
    bad = []
    names = names or {}
    for name, body in funcs.items():
        open_div, tagged = 0, 0
        for ins in body:
            if FORBIDDEN.match(ins):
                bad.append(f"{name}: {ins}")
            elif ins.startswith("v_div_scale_f32"):
                open_div += 1
            elif ins.startswith("v_div_fixup_f32"):
                open_div = max(0, open_div - 2)
            elif ins.endswith("ggq:fastdiv"):
                tagged += 1
            elif FMA.match(ins) and open_div == 0:
                bad.append(f"{name}: fused multiply-add outside a division sequence: {ins}")
        pretty = names.get(name, name)
        m = re.match(r"^void ggq::quant_(?:one|batch)<ggq::(Enc\w+),", pretty)
        if m:
            per_call = FASTDIV_STEPS[m.group(1)]
            if tagged not in (per_call, 2 * per_call):
                bad.append(f"{pretty}: {tagged} tagged fused steps, expected {per_call} per encode call")
            if any(ins.startswith("v_cmpx") for ins in body):
                bad.append(f"{pretty}: a vector compare writes EXEC in a kernel with hand-placed DPP instructions")
        elif tagged and "fast_division_selftest" not in pretty:
            bad.append(f"{pretty}: tagged fused steps outside the quantizers")
    return bad


def _regs(tok):
    """VGPR numbers named by an operand token: v7, |v7|, -v7, v[4:7]."""
    tok = tok.strip().strip("|").lstrip("-").strip("|")
    m = re.match(r"^v(\d+)$", tok)
    if m:
        return {int(m.group(1))}
    m = re.match(r"^v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def _operands(ins):
    """(mnemonic, [operand tokens]) of an instruction line; modifiers such as quad_perm:[..] stay out."""
    ins = ins.split(";")[0].strip()
    parts = ins.split(None, 1)
    if len(parts) < 2:
        return parts[0], []
    ops, depth, cur = [], 0, ""
    for ch in parts[1]:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur)
            cur = ""
        else:
            cur += ch
    ops.append(cur)
    ops = [o.strip().split(" ")[0] if not o.strip().startswith("v[") else o.strip().split("]")[0] + "]" for o in ops]
    return parts[0], ops


TRANS = re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)(_iflag|_legacy)?_f(16|32|64)")


def check_hand_placed_hazards(funcs):
    """The library places instructions by hand (asm) whose hazards the compiler cannot see, so they are checked on
    the generated code of every function:
      * the result of a transcendental instruction (v_rcp_f32 ...) is not read by the very next instruction
        (one wait state on this hardware; an s_nop in between provides it);
      * a DPP instruction or a v_permlane*_swap does not read a VGPR that one of the vector instructions within the
        two wait states in front of it wrote (s_nop N counts N + 1) -- the compiler pads for writers it can see, not
        for a hand-placed one;
      * the result of a dot-product instruction (v_dot4 ...) is not read by a vector instruction within three wait
        states (found the hard way: a hand-placed DPP add two states behind a v_dot4c read the old register);
      * v_readfirstlane does not read a VGPR that the instruction right in front of it wrote (the group start's
        opaque zero goes from a hand-placed v_and_b32 into a v_readfirstlane);
      * a hand-placed v_cndmask_b32 does not take its lane mask from scalar registers that a VECTOR instruction
        wrote within two wait states (the masks of max_by_abs come out of scalar mask arithmetic)."""
    bad = []
    for name, body in funcs.items():
        for i, ins in enumerate(body):
            mn, ops = _operands(ins)
            if TRANS.match(mn) and ops and i + 1 < len(body):
                dst = _regs(ops[0])
                nmn, nops = _operands(body[i + 1])
                if not nmn.startswith("s_") and any(_regs(o) & dst for o in nops[1:] if nmn.startswith("v_")):
                    bad.append(f"{name}: {body[i + 1]!r} reads the result of {ins!r} without a wait state")
            if mn.startswith("v_dot") and ops:              # a dot product's result: three wait states before a vector reader
                dst = _regs(ops[0])
                states, j = 0, i + 1
                while j < len(body) and states < 3:
                    nmn, nops = _operands(body[j])
                    if nmn == "s_nop":
                        states += int(nops[0]) + 1 if nops else 1
                    else:
                        # (the accumulate chain of the same kind of instruction is exempt: dot -> dot on its own dst)
                        if nmn.startswith("v_") and not nmn.startswith("v_dot") and any(_regs(o) & dst for o in nops[1:]):
                            bad.append(f"{name}: {body[j]!r} reads the result of {ins!r} after {states} wait states")
                        if nmn.startswith("v_") and nops and _regs(nops[0]) & dst:
                            break                           # overwritten: later readers see another value
                        states += 1
                    j += 1
            if mn.startswith("v_readfirstlane") and len(ops) == 2 and i > 0:
                pmn, pops = _operands(body[i - 1])
                if pmn.startswith("v_") and pops and _regs(pops[0]) & _regs(ops[1]):
                    bad.append(f"{name}: {ins!r} reads what {body[i - 1]!r} wrote without a wait state")
            if mn == "v_cndmask_b32" and len(ops) == 4:      # no _e32 / _e64 suffix: placed by hand (asm)
                m = re.match(r"^s\[(\d+):(\d+)\]$", ops[3])
                mask = set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()
                states, j = 0, i - 1
                while j >= 0 and states < 2:
                    pmn, pops = _operands(body[j])
                    if pmn == "s_nop":
                        states += int(pops[0]) + 1 if pops else 1
                    else:
                        pm = re.match(r"^s\[(\d+):(\d+)\]$", pops[0]) if pops else None
                        if pmn.startswith("v_") and pm and set(range(int(pm.group(1)), int(pm.group(2)) + 1)) & mask:
                            bad.append(f"{name}: {ins!r} reads a mask that {body[j]!r} (a vector instruction) wrote "
                                       f"{states} wait states earlier")
                        states += 1
                    j -= 1
            if ("_dpp" in mn or mn.startswith("v_permlane")) and len(ops) >= 2:
                # (the row / half swaps read and write both operands, and need the same two wait states)
                src = _regs(ops[1]) | (_regs(ops[0]) if mn.startswith("v_permlane") else set())
                states, j = 0, i - 1
                while j >= 0 and states < 2:
                    pmn, pops = _operands(body[j])
                    if pmn == "s_nop":
                        states += int(pops[0]) + 1 if pops else 1
                    else:
                        if pmn.startswith("v_") and pops and _regs(pops[0]) & src:
                            bad.append(f"{name}: {ins!r} reads a register {body[j]!r} wrote {states} wait states earlier")
                        states += 1
                    j -= 1
    return bad


def check_group_start(funcs, names):
    """Every deq_one<F, O, ...> and quant_one<Q, In, ...>: before its first tile load (global_load_dwordx4) it must
    contain, in this order, a scalar ONE-dword load from memory other than the kernel arguments (the probe of the
    destination), a wait for it, the opaque `v_and_b32 v, 0, v` and the v_readfirstlane that feeds the result into
    the source address."""
    bad, seen = [], 0
    for name, body in funcs.items():
        if not names[name].startswith(("void ggq::deq_one<", "void ggq::quant_one<")):
            continue
        seen += 1
        try:
            first_tile = next(i for i, s in enumerate(body) if s.startswith("global_load_dwordx4"))
        except StopIteration:
            bad.append(f"{names[name]}: no tile load found")
            continue
        head = body[:first_tile]
        try:
            # (s[0:1] holds the kernel argument pointer: a one-dword ARGUMENT load, e.g. quant_one's group size, is
            # not the probe)
            i_probe = next(i for i, s in enumerate(head) if re.match(r"^s_load_dword\s+s\d+, (?!s\[0:1\])", s))
            i_wait = next(i for i, s in enumerate(head) if i > i_probe and s.startswith("s_waitcnt") and "lgkmcnt(0)" in s)
            i_and = next(i for i, s in enumerate(head) if i > i_wait and re.match(r"^v_and_b32\s+v\d+, 0, v\d+", s))
            next(i for i, s in enumerate(head) if i > i_and and s.startswith("v_readfirstlane_b32"))
        except StopIteration:
            bad.append(f"{names[name]}: the group start's scalar probe does not precede the first tile load")
            continue
        if any(s.startswith(("global_load_dword ", "flat_load_dword ", "buffer_load_dword ")) for s in head[:i_and]):
            bad.append(f"{names[name]}: the probe went to the vector memory path")
    if seen < 40:
        bad.append(f"only {seen} deq_one / quant_one kernels found")
    return bad


def resources(remarks):
    """{mangled name: {field: int}} from -Rpass-analysis=kernel-resource-usage."""
    out, cur = {}, None
    for line in remarks.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return out


# elements and packed bytes per block (gguf_amd/csrc/formats.h)
BLOCK = {"Q4_0": (32, 18), "Q4_1": (32, 20), "Q5_0": (32, 22), "Q5_1": (32, 24), "Q8_0": (32, 34), "Q8_1": (32, 36),
         "Q2K": (256, 84), "Q3K": (256, 110), "Q4K": (256, 144), "Q5K": (256, 176), "Q6K": (256, 210),
         "Q8K": (256, 290), "IQ4NL": (32, 18), "IQ4XS": (256, 136)}


def tile_lds_bytes(fmt, out):
    """TileGeom<F, O>::LDS_SLOTS * 16 (gguf_amd/csrc/kernels.h): 512 threads x 2 stores of 16 bytes."""
    epl = 4 if out == "OutF32" else 8
    elems = 512 * epl * 2
    n, b = BLOCK[fmt]
    tbytes = elems // n * b
    return ((tbytes + 15) // 16 + 2) * 16


def check_resources(res, names):
    bad, seen = [], 0
    for name, r in res.items():
        d = names.get(name, name)
        m = re.match(r"^void ggq::(deq_one|deq_batch)<ggq::(\w+), ggq::(\w+),", d)
        if r.get("ScratchSize", 0) != 0 or r.get("VGPRs Spill", 0) != 0 or r.get("SGPRs Spill", 0) != 0:
            bad.append(f"{d}: scratch {r.get('ScratchSize')} B/lane, spills {r.get('VGPRs Spill')}/{r.get('SGPRs Spill')}")
        if re.match(r"^void ggq::(quant_one|quant_batch|narrow_vec|widen_vec|permute_qk_rows|scale_vec)<", d) and \
                r.get("Occupancy") != 8:
            bad.append(f"{d}: {r.get('Occupancy')} waves/SIMD (VGPRs {r.get('VGPRs')}), 8 expected")
        if not m:
            continue
        seen += 1
        if r.get("Occupancy") != 8:
            bad.append(f"{d}: {r.get('Occupancy')} waves/SIMD (VGPRs {r.get('VGPRs')}), 8 expected")
        want = tile_lds_bytes(m.group(2), m.group(3))
        if r.get("LDS Size") != want:
            bad.append(f"{d}: LDS {r.get('LDS Size')} bytes, tile geometry says {want}")
    if seen < 2 * 14 * 3:
        bad.append(f"only {seen} decode kernels in the resource report")
    return bad
