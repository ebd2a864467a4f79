"""Instruction mix of one kernel in a hipcc -S listing (development aid).
usage: isa_mix.py listing.s <substring of the mangled kernel name> [--top N]
Prints the static instruction count per class for the whole kernel and for its prologue (everything before the
first v_mfma), so that device-side evaluation of layout tables (scalar loops with s_load in the prologue) shows up."""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 0
    body, inside = [], False
    for line in open(path):
        if re.match(r"^_Z\w+:", line):
            inside = key in line
            continue
        if inside and line.startswith("\t.end_amdhsa_kernel") or (inside and line.startswith(".Lfunc_end")):
            inside = False
        if inside:
            t = line.strip()
            if t and not t.startswith((";", ".")) and not t.endswith(":"):
                body.append(t.split()[0])
    first_mfma = next((i for i, op in enumerate(body) if op.startswith("v_mfma")), len(body))
    for name, ops in (("kernel", body), ("prologue", body[:first_mfma])):
        c = collections.Counter(classify(op) for op in ops)
        print(f"{name:9s} total {len(ops):6d}  " + "  ".join(f"{k} {v}" for k, v in sorted(c.items())))
    if top:
        for op, n in collections.Counter(body).most_common(top):
            print(f"  {n:5d} {op}")


if __name__ == "__main__":
    main()
