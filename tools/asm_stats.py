#!/usr/bin/env python3
"""Static instruction mix of the kernels in a gfx950 assembly file (hipcc --cuda-device-only -S).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DFRAD_ASM_MARKS --cuda-device-only -S frad_p1_wave.hip -o /tmp/p1w.s
    python tools/asm_stats.py /tmp/p1w.s [name substring]

The wave kernels are fully unrolled inside their unit loop, so the static count of the loop body is the dynamic count
per unit (frame) up to the rare exact-path blocks.  Classes: valu f64 / f32+int / transcendental, salu, lds read / write,
vmem, scratch, waitcnt.  Prints the whole-kernel count and the count of the largest backward-branch loop."""
import re
import subprocess
import sys

TRANS = ("v_rcp", "v_sqrt", "v_rsq", "v_log", "v_exp", "v_sin", "v_cos")


def classify(op):
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("ds_"):
        return "lds_w" if ("write" in op or "store" in op or "add" in op or "_or_" in op) else "lds_r"
    if op.startswith(("global_", "buffer_", "flat_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_"):
        if op.startswith(TRANS): return "trans"
        if "f64" in op: return "v_f64"
        return "v_other"
    return "other"


def main():
    path = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    lines = open(path).read().splitlines()
    # kernel bodies: from "<sym>:" (a .globl'd _Z symbol followed by code) to s_endpgm
    i = 0
    names = {}
    while i < len(lines):
        m = re.match(r"^(_Z\w+):", lines[i])
        if m:
            sym = m.group(1)
            body = []
            j = i + 1
            while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
                body.append(lines[j]); j += 1
            names[sym] = body
            i = j
        i += 1
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    for sym, d in zip(names, dem):
        d = re.sub(r"\(.*$", "", d).replace("void frad::", "")
        if pat and pat not in d:
            continue
        body = names[sym]
        ins = []          # (index, op) ; labels: name -> position in ins
        labels = {}
        for ln in body:
            s = ln.strip()
            if not s or s.startswith((";", ".", "//")):
                m = re.match(r"^(\.LBB\w+):", s)
                if m: labels[m.group(1)] = len(ins)
                continue
            m = re.match(r"^(\.?\w+):", s)
            if m:
                labels[m.group(1)] = len(ins); continue
            ins.append(s.split()[0] if s.split() else "?")
            ins[-1] = (ins[-1], s)

        def mix(seq):
            c = {}
            for op, _ in seq:
                k = classify(op); c[k] = c.get(k, 0) + 1
            return c
        total = mix(ins)
        # largest backward branch
        best = (0, 0, 0)
        for idx, (op, s) in enumerate(ins):
            if op.startswith(("s_cbranch", "s_branch")):
                tgt = s.split()[-1]
                if tgt in labels and labels[tgt] <= idx and idx - labels[tgt] > best[0]:
                    best = (idx - labels[tgt], labels[tgt], idx)
        print(f"{d}: {len(ins)} instructions {total}")
        # sections between "; FRAD_MARK name" comments (inline-asm marks of the kernel source), in program order
        marks = []
        n = 0
        for ln in body:
            st = ln.strip()
            m = re.search(r"FRAD_MARK (\w+)", st)
            if m and not st.startswith("."):
                marks.append((m.group(1), n)); continue
            if st and not st.startswith((";", ".", "//")) and not re.match(r"^(\.?\w+):", st):
                n += 1
        for (name, a), (_, b) in zip(marks, marks[1:] + [("end", len(ins))]):
            sm = mix(ins[a:b])
            slots = sm.get("v_f64", 0) + sm.get("v_other", 0) + 4 * sm.get("trans", 0)
            print(f"    [{name:>20}] {b - a:5d} instr, VALU slots {slots:5d}  {sm}")
        if best[0]:
            lm = mix(ins[best[1]:best[2] + 1])
            issue = lm.get("v_f64", 0) + lm.get("v_other", 0) + 4 * lm.get("trans", 0)
            print(f"    main loop: {best[0]} instructions {lm}; VALU issue slots (trans x4) {issue}")


if __name__ == "__main__":
    main()
