#!/usr/bin/env python3
"""Loop spans of a kernel's ISA: every backward branch of one function in an `llvm-objdump -d` listing, with the bytes between the
target and the branch and the instruction mix inside -- is the steady-state loop larger than the instruction cache (64 KB per CU
pair on gfx950)?    llvm-objdump -d --no-show-raw-insn X.elf > X.s ; tools/isa_loops.py X.s 'pc_kernel<32, 4, 3, 2, 1>' """
import re
import subprocess
import sys


def functions(path):
    cur, body = None, {}
    for line in open(path):
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
        if m:
            cur = m.group(2)
            body[cur] = []
            continue
        if cur is None:
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)$", line)
        if m:
            body[cur].append((int(m.group(3), 16), m.group(1), m.group(2) + " " + m.group(4)))
    return body


def main():
    path, want = sys.argv[1], sys.argv[2]
    fns = functions(path)
    names = {n: subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip() for n in fns}
    for n, ins in fns.items():
        if want not in names[n] or not ins:
            continue
        print(names[n][:100], "--", ins[-1][0] - ins[0][0], "bytes,", len(ins), "instructions")
        addr_of = {a: i for i, (a, _, _) in enumerate(ins)}
        for i, (a, op, args) in enumerate(ins):
            if not op.startswith("s_cbranch") and op != "s_branch":
                continue
            m = re.search(r"<.*\+0x([0-9a-f]+)>", args)
            if not m:
                continue
            tgt = ins[0][0] + int(m.group(1), 16)
            if tgt >= a or a - tgt < 512:
                continue
            j = addr_of.get(tgt)
            inside = ins[j:i + 1] if j is not None else []
            mix = {}
            for _, o, _ in inside:
                k = ("mfma" if "mfma" in o else "ds_read" if o.startswith("ds_read") else "ds_write" if o.startswith("ds_write") else
                     "vmem" if o.startswith(("global_", "buffer_", "flat_")) else "barrier" if o == "s_barrier" else
                     "waitcnt" if o == "s_waitcnt" else "valu" if o.startswith("v_") else "salu")
                mix[k] = mix.get(k, 0) + 1
            print(f"  loop {tgt - ins[0][0]:#8x} .. {a - ins[0][0]:#8x}: {a - tgt:6d} bytes  {op:16s} {mix}")


main()
