#!/usr/bin/env python3
"""Static VALU opcode mix of every kernel of liborbx against the measured issue cost of each opcode class.

tools/ubench_valu.hip (run on the MI355X, output committed as profiles/r04_valu_issue.txt) measures cycles per wave64 instruction per SIMD
for 27 opcodes: two classes come out, ~2.2 cycles (v_add_u32, v_sub_u32, v_and_b32, v_xor_b32, v_add_f32, v_fma_f32) and ~4.1 cycles
(shifts, min / max / med3, v_bcnt, v_perm, v_alignbyte, v_dot2 / v_dot4, v_sad, v_mad, v_bfe, the three-operand adds / logic ops, every
packed op, v_mul_lo).  This script compiles csrc/*.hip to ISA (hipcc -S, no GPU needed), counts the VALU opcodes of every kernel and
prices them with that table: opcodes the benchmark covers get their measured cost, the others the class of their closest relative
(listed in the output as `assumed`).  The mix is STATIC (every instruction counted once, loops not weighted): an estimate of the kernel's
average issue cost, good enough to say which of "2 cycles" (MI355X_MICROARCH.md) and "4 cycles" a kernel lives at.
   python tools/valu_mix.py [profiles/r05_valu_issue.txt] > profiles/r05_valu_mix.json      (round 5: 63 opcodes measured)"""
import json
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "self_commit_orb-slam2_amd" / "csrc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-S", "--cuda-device-only"]
# opcodes the benchmark does not cover, priced like their closest measured relative.  Round 5 measured 36 more opcodes: v_cndmask_b32, v_addc_co_u32, f32 min / max
# and v_cvt_f32_ubyte* turned out to be HALF rate (they were assumed full rate in round 4) and left this list; v_cmp_* is full rate (v_cmp + v_cndmask pair: 6.2 cycles);
# the two-operand 16-bit integer forms are full rate.
FULL_LIKE = ("v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_add_f32", "v_sub_f32", "v_mul_f32", "v_fma_f32",
             "v_fmac_f32", "v_cmp_", "v_add_co_u32", "v_sub_co_u32", "v_accvgpr", "v_readlane", "v_readfirstlane", "v_writelane",
             "v_lshrrev_b32", "v_min_u16", "v_max_u16", "v_min_i16", "v_max_i16", "v_add_u16", "v_sub_u16", "v_mul_lo_u16", "v_lshlrev_b16", "v_lshrrev_b16",
             "v_cvt_u32_f32", "v_cvt_i32_f32", "v_rcp_f32", "v_rndne_f32", "v_add_f64", "v_mul_f64", "v_fma_f64")


def measured(path):
    tab, clocks = {}, []
    for line in open(path):
        m = re.match(r"(v_\w+)\s+([0-9.]+) cycles per wave-instr per SIMD.*clock ([0-9.]+) GHz", line)
        if m:
            tab[m.group(1)] = float(m.group(2))
            clocks.append(float(m.group(3)))
    return tab, (sorted(clocks)[len(clocks) // 2] if clocks else 2.4)


def kernels_of(src):
    with tempfile.TemporaryDirectory() as d:
        out = Path(d) / "k.s"
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", str(out), str(src)], check=True, stderr=subprocess.DEVNULL)
        text = out.read_text()
    res, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m and "k_" in m.group(1):
            cur = m.group(1)
            lm = re.search(r"(\d+)(?=k_)", m.group(1))      # Itanium mangling: <length><name>; the digit run may start with the tail of "_N_1"
            if lm:
                st = lm.end()
                for cut in range(len(lm.group(1))):
                    n = int(lm.group(1)[cut:])
                    if 2 < n <= len(m.group(1)) - st and re.fullmatch(r"k_[a-z0-9_]+", m.group(1)[st:st + n]):
                        cur = m.group(1)[st:st + n]
                        break
            res.setdefault(cur, {})
            continue
        if line.strip().startswith("s_endpgm"):
            cur = None
            continue
        if cur:
            m = re.match(r"^\s+(v_[a-z0-9_]+)", line)
            if m:
                op = re.sub(r"_(e32|e64|sdwa|dpp)$", "", m.group(1))
                res[cur][op] = res[cur].get(op, 0) + 1
    return res


def main():
    path = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "profiles" / "r05_valu_issue.txt"
    tab, clock = measured(path)
    half = sorted(v for v in tab.values() if v > 3.0)
    full = sorted(v for v in tab.values() if v <= 3.0)
    c_half, c_full = half[len(half) // 2], full[len(full) // 2]
    out = {"source": str(path.relative_to(ROOT)), "clock_GHz_median": clock, "cycles_half_rate_class": c_half, "cycles_full_rate_class": c_full,
           "measured_opcodes": tab, "kernels": {}}
    for src in sorted(CSRC.glob("*.hip")):
        for k, ops in kernels_of(src).items():
            n = sum(ops.values())
            if n < 20:
                continue
            cyc, assumed = 0.0, {}
            for op, c in ops.items():
                if op in tab:
                    cyc += c * tab[op]
                else:
                    cls = c_full if op.startswith(FULL_LIKE) else c_half
                    cyc += c * cls
                    assumed[op] = "full" if cls == c_full else "half"
            top = sorted(ops.items(), key=lambda kv: -kv[1])[:8]
            out["kernels"].setdefault(k, {"file": src.name, "valu_static": 0, "cycles_per_instr_static_mix": 0.0, "top_opcodes": top, "assumed_class": assumed})
            e = out["kernels"][k]
            if n > e["valu_static"]:      # (template instances: keep the largest)
                e.update(valu_static=n, cycles_per_instr_static_mix=round(cyc / n, 3), top_opcodes=top, assumed_class=assumed, file=src.name)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
