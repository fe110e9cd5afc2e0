#!/usr/bin/env python3
"""Build-time ISA check of the sweeps that issue their loads by hand (RowLoad / row_wait, csrc/kernels_op2.hpp).

Those loads are inline asm: the compiler's wait-count pass does not see them, the kernels state the counts themselves.
What keeps that correct is (a) the register ring of a load is not used for anything else while the load is in flight
and (b) nothing in the loop drains the queue.  Bit-for-bit tests against the tile-map kernels guard the results on
today's compiler; this script guards the mechanism when ROCm changes: it compiles the default-path instantiations to
assembly (cross-compiles without a GPU, a few seconds) and asserts, for the main loop of each kernel:

  * no `s_waitcnt vmcnt(0)` (a drain: one per batch was the 11.0 ms form of round 3) and no scratch access;
  * vector memory operations return in order, so after `s_waitcnt vmcnt(N)` only the N youngest are outstanding: walking
    the loop (twice: the second trip starts with the loads the first left in flight), no instruction may read or
    write a VGPR that an outstanding load is still going to write -- no use before the wait, no reuse of a ring
    register as a temporary, no accvgpr / mov shuffling of it while the load flies;
  * vmcnt values fit their 6 bits.

  python tools/isa_check.py [-v]      exit code 0 = all checks hold; prints one summary line per kernel
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

OSR = "template __global__ void temx::sweep_osr_kernel<double, 7, 13, 2, 2, %d>(FieldPtrs<4>, int64_t, int, int, const double*, const int4*, const int2*, const double*, const double*, int, double*, double*, int, int);"
OS2 = "template __global__ void temx::sweep_os2_kernel<float, 7, 13, 2, 2, %d>(FieldPtrs<4>, int64_t, int, int, const double*, const int4*, const int4*, const int*, const int*, const int2*, const double*, const double*, int, double*, double*, int, int);"
OPR = "template __global__ void temx::sweep_opr_kernel<double, 7, 2, 0>(FieldPtrs<4>, int64_t, int, const double*, const int4*, const int2*, const double*, double*, int, int, double*);"


def regs(tok):
    """VGPR numbers named by an operand token: v12 -> {12}, v[10:13] -> {10..13}."""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def functions(asm):
    cur, name = None, None
    for line in asm.splitlines():
        m = re.match(r"^(_ZN4temx\w+):", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            cur.append(line)
            if "s_endpgm" in line:
                yield name, cur
                cur = None


def loops(body):
    """(start, end) line ranges of backward branches."""
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            labels[m.group(1)] = i
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\w+)", l) or re.search(r"s_branch\s+(\.LBB\w+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            yield labels[m.group(1)], i


VREG = re.compile(r"(?<![\w.\]])v\[(\d+):(\d+)\]|(?<![\w.\]])v(\d+)\b")


def vregs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def pk_used(op, rest):
    """Registers a packed-fp32 VALU instruction reads or writes: each source is a register pair of which op_sel /
    op_sel_hi pick one register per half (op_sel 0 -> the low register feeds the low result, op_sel_hi 1 -> the high
    register feeds the high result); a pair named with op_sel_hi 0 and op_sel 0 is only read in its low register."""
    ops = [o.strip() for o in rest.split(",")]
    mods = " ".join(o for o in ops if ":" in o and "[" in o and not o.startswith(("v[", "-v[", "|v[")))
    plain = [o for o in rest.replace(mods, "").split(",") if o.strip()]

    def sel(name, n, dflt):
        m = re.search(name + r":\[([01,]+)\]", rest)
        v = [int(x) for x in m.group(1).split(",")] if m else []
        return (v + [dflt] * n)[:n]
    toks = []
    for o in plain:
        t = o.strip().split()[0] if o.strip() else ""
        toks.append(t)
    toks = [t for t in toks if t and not t.startswith(("op_sel", "neg_"))]
    used = vregs(toks[0]) if toks else set()                  # destination pair: both written
    srcs = toks[1:]
    lo, hi = sel("op_sel", len(srcs), 0), sel("op_sel_hi", len(srcs), 1)
    for i, t in enumerate(srcs):
        r = sorted(vregs(t))
        if len(r) == 2:
            used.add(r[1] if lo[i] else r[0])
            used.add(r[1] if hi[i] else r[0])
        else:
            used |= set(r)
    return used


def check(name, body, verbose=False):
    hand = re.compile(r"^\s*global_load_dword(x2)?\s+(\S+),\s*v\d+,\s*s\[\d+:\d+\].*\bnt\b")
    best = None
    for a, b in loops(body):
        n = sum(1 for l in body[a:b] if hand.match(l))
        if n and (best is None or n > best[2]):
            best = (a, b, n)
    if best is None:
        return ["no loop with hand-issued loads found"], "?"
    a, b, nhand = best
    loop = [l.split(";")[0].rstrip() for l in body[a:b + 1]]
    loop = [l for l in loop if l.strip() and not l.strip().startswith(".") and not l.strip().endswith(":")]
    errs = []
    vm = []
    for l in loop:
        if "scratch_" in l:
            errs.append("scratch access in the loop: " + l.strip())
        for m in re.finditer(r"vmcnt\((\d+)\)", l):
            vm.append(int(m.group(1)))
    if 0 in vm:
        errs.append("s_waitcnt vmcnt(0) inside the loop (%d times): a drain of the load ring" % vm.count(0))
    if vm and max(vm) > 63:
        errs.append("vmcnt beyond 6 bits")
    # Vector memory operations return in order: after `s_waitcnt vmcnt(N)` only the N youngest are outstanding.  Walk
    # the loop twice (the second trip starts with what the first left in flight) and require that no instruction
    # touches a register an outstanding load is still going to write.
    fifo = []
    worst = 0
    for trip in range(2):
        for l in loop:
            t = l.strip()
            op = t.split()[0]
            rest = t[len(op):]
            if op.startswith(("global_load", "buffer_load")):
                dst = vregs(rest.split(",")[0])
                if trip == 1:
                    for r_ in fifo:
                        if r_ & (vregs(rest) - dst):
                            errs.append("address / data register of `%s` is the target of a load in flight" % t)
                    for r_ in fifo:
                        if r_ & dst:
                            errs.append("`%s` targets a register another load in flight targets" % t)
                fifo.append(dst)
                worst = max(worst, len(fifo))
                continue
            if op.startswith(("global_store", "buffer_store")):
                if trip == 1:
                    used = vregs(rest)
                    for r_ in fifo:
                        if r_ & used:
                            errs.append("`%s` reads a register a load in flight will write" % t)
                fifo.append(set())
                continue
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", t)
                if m:
                    n = int(m.group(1))
                    while len(fifo) > n:
                        fifo.pop(0)
                continue
            if trip == 1 and (op.startswith(("v_", "ds_")) or op.startswith("s_") is False):
                used = pk_used(op, rest) if op.startswith("v_pk_") and "f32" in op else vregs(rest)
                for r_ in fifo:
                    hit = r_ & used
                    if hit:
                        errs.append("`%s` touches v%d while a load into it is in flight" % (t, sorted(hit)[0]))
                        break
    summary = "loop of %d instructions: %d hand-issued loads per trip, at most %d memory operations in flight, vmcnt values %s, %d MFMAs" % (
        len(loop), nhand, worst, sorted(set(vm)), sum(1 for l in loop if "v_mfma" in l))
    return sorted(set(errs)), summary


def main():
    verbose = "-v" in sys.argv
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "isa_tu.hip")
        with open(src, "w") as fh:
            fh.write('#include <hip/hip_runtime.h>\n#include "%s"\nusing namespace temx;\n' % os.path.join(ROOT, "pytemdiags_amd", "csrc", "kernels_op2.hpp"))
            for k in (0, 1, 3):
                fh.write(OSR % k + "\n" + OS2 % k + "\n")
            fh.write(OPR + "\n")
        out = os.path.join(td, "isa_tu.s")
        subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", out, src],
                       check=True, stderr=subprocess.DEVNULL)
        asm = open(out).read()
    bad = 0
    n = 0
    for name, body in functions(asm):
        if not any(k in name for k in ("sweep_osr", "sweep_os2", "sweep_opr")):
            continue
        n += 1
        errs, summary = check(name, body, verbose)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        print("%-58s %s  %s" % (dem.replace("void temx::", ""), "OK  " if not errs else "FAIL", summary))
        for e in errs[:8]:
            print("      " + e)
        bad += bool(errs)
    print("%d kernels checked, %d failed" % (n, bad))
    return 1 if bad or n != 7 else 0


if __name__ == "__main__":
    sys.exit(main())
