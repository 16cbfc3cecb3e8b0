"""Instruction histogram of one kernel's main loop from `hipcc -S --cuda-device-only` output.

    python tools/isa_hist.py file.s <kernel-name-substring> [loop-label-substring]

Counts opcodes between the kernel's entry and its s_endpgm; with the third argument only inside the basic blocks whose label
contains it (e.g. the hot loop found by reading the file once)."""
import collections
import re
import sys


def main():
    s = open(sys.argv[1]).read()
    pat = sys.argv[2]
    for m in re.finditer(r'^(_Z\S*):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M):
        if pat not in m.group(1):
            continue
        body = m.group(2)
        # basic blocks
        blocks = re.split(r'^(\.LBB\S+):.*$', body, flags=re.M)
        print(m.group(1)[:90])
        names = ["entry"] + blocks[1::2]
        texts = [blocks[0]] + blocks[2::2]
        for nm, tx in zip(names, texts):
            ops = collections.Counter(l.split()[0] for l in tx.split("\n") if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";")))
            n = sum(ops.values())
            if n < 40:
                continue
            cls = collections.Counter()
            for k, v in ops.items():
                c = ("mfma" if "mfma" in k else "exp" if k.startswith(("v_exp", "v_rcp", "v_log")) else "pk" if k.startswith("v_pk") else
                     "valu" if k.startswith("v_") else "lds" if k.startswith("ds_") else "vmem" if k.startswith(("buffer", "global")) else
                     "salu" if k.startswith("s_") else "other")
                cls[c] += v
            print(f"  {nm:14s} {n:5d}  " + "  ".join(f"{k}={v}" for k, v in sorted(cls.items())))
            if len(sys.argv) > 3 and sys.argv[3] in nm:
                for k, v in ops.most_common(30):
                    print("        ", k, v)


if __name__ == "__main__":
    main()
