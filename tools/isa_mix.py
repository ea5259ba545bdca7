#!/usr/bin/env python
"""Instruction-class mix of the chain kernels' loops from the SHIPPED code object (no GPU needed):

    python tools/isa_mix.py [smcpp_amd/libsmcpp_engine.so] > profiles/r06_isa_mix.json

Why: a SIMD of gfx950 issues a plain 32-bit VALU instruction of a wave64 in 2 cycles and a DPP, 64-bit (fp64 / cvt / 64-bit
integer) or cross-lane (v_readlane ...) instruction in 4 (tools/dpp_lab.hip on the box: v_add_f32 982 - 1002 G/s, v_fmac_f32_dpp /
v_mov_b32_dpp 587, v_fma_f64 583, eight wavefronts per SIMD; MI355X_MICROARCH.md: 157.3 TFLOP/s FP32 vector = 2-cycle issue).
bench.py's issue roofline therefore prices the kernel's instructions per CLASS: peak = 1024 SIMDs x 2.4 GHz / mean cycles per
instruction of the mix.  The mix is STATIC: every VALU instruction inside a loop of the kernel that contains DPP instructions (the
row / position loops of the chains), each loop weighted by its length - the loops are what executes ~all of the kernel's dynamic
instructions, and their mixes differ little from each other.  Transcendentals (v_rcp_f32: quarter rate) are listed separately.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(so):
    d = tempfile.mkdtemp()
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           f"--input={fat}", f"--output={co}", "--unbundle"])
    return subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", co], text=True)


def classify(mn, ops):
    """-> class of one VALU instruction: 'dpp' | 'w64' | 'lane' | 'trans' | 'plain32'"""
    if "_dpp" in mn or "row_" in ops or "wave_sh" in ops or "quad_perm" in ops:
        return "dpp"
    if mn.startswith(("v_readlane", "v_writelane", "v_readfirstlane", "v_permlane")):
        return "lane"
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_", mn):
        return "trans"
    if re.search(r"(f64|_u64|_i64|_b64)", mn):
        return "w64"
    return "plain32"


CYCLES = {"dpp": 4, "w64": 4, "lane": 4, "trans": 8, "plain32": 2}


def kernel_mix(text, symbol_re):
    out = {}
    cur = None
    body = {}
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = m.group(1)
            body[cur] = []
            continue
        if cur is None:
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", line)
        if m:
            body[cur].append((int(m.group(3), 16), m.group(1), m.group(2)))
    for sym, ins in body.items():
        if not re.search(symbol_re, sym) or not ins:
            continue
        addr_index = {a: i for i, (a, _, _) in enumerate(ins)}
        loops = []
        for i, (a, mn, ops) in enumerate(ins):
            if mn.startswith(("s_cbranch", "s_branch")):
                try:
                    off = int(ops.split()[0])
                except (ValueError, IndexError):
                    continue
                if off >= 32768:                       # backward branch: simm16, in dwords from the next instruction
                    tgt = a + 4 + 4 * (off - 65536)
                    if tgt in addr_index:
                        loops.append((addr_index[tgt], i))
        # innermost first; count each instruction once (in the innermost loop that holds it)
        loops.sort(key=lambda l: l[1] - l[0])
        owner = [None] * len(ins)
        for k, (lo, hi) in enumerate(loops):
            for j in range(lo, hi + 1):
                if owner[j] is None:
                    owner[j] = k
        tot = {c: 0 for c in CYCLES}
        salu = 0
        nloop = 0
        for k, (lo, hi) in enumerate(loops):
            mine = [ins[j] for j in range(lo, hi + 1) if owner[j] == k]
            # the chain loops: an outer row loop holds the inner position loop, so a loop counts when IT or any loop inside it has DPP
            span = [ins[j] for j in range(lo, hi + 1)]
            if not any("_dpp" in mn for _, mn, _ in span):
                continue
            nloop += 1
            for _, mn, ops in mine:
                if mn.startswith("v_") and not mn.startswith("v_cmpx_nop"):
                    tot[classify(mn, ops)] += 1
                elif mn.startswith("s_"):
                    salu += 1
        n = sum(tot.values())
        if n == 0:
            continue
        cyc = sum(tot[c] * CYCLES[c] for c in tot) / n
        out[sym] = {"loops_with_dpp": nloop, "valu_in_those_loops": n, "salu_in_those_loops": salu, "by_class": tot,
                    "mean_issue_cycles_per_valu": cyc, "peak_ginstr_per_s": 1024 * 2.4 / cyc}
    return out


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "smcpp_amd", "libsmcpp_engine.so")
    text = disassemble(so)
    res = {"note": "static VALU class mix of the loops that contain DPP instructions, from the shipped gfx950 code object "
                   "(tools/isa_mix.py); issue cycles per wave64 instruction: " + json.dumps(CYCLES) +
                   " (tools/dpp_lab.hip; MI355X_MICROARCH.md); peak = 1024 SIMDs x 2.4 GHz / mean cycles",
           "kernels": kernel_mix(text, r"k_chain_ss|k_span_scan")}
    json.dump(res, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
