#!/usr/bin/env python3
"""Find serialised global round trips in compiled kernels: counts, per kernel of a hipcc -save-temps .s file, the global /
buffer loads that are waited for with `s_waitcnt vmcnt(0)` before the NEXT load is issued (a chain of such pairs is a chain
of memory latencies nothing hides -- e.g. a loop `load; split(asm volatile); load; ...` that the compiler did not batch).
usage: isa_serial_loads.py file.s [min_chain]"""
import re
import sys


def main():
    s = open(sys.argv[1]).read()
    thr = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    for m in re.finditer(r"^(_Z\w+):\s*; @\w+\n(.*?)^\s*\.end_amdhsa_kernel", s, re.S | re.M):
        name, body = m.group(1), m.group(2)
        ev = re.findall(r"^\s*(global_load_\w+|buffer_load_\w+|s_waitcnt vmcnt\(\d+\)|s_waitcnt[^\n]*vmcnt\(\d+\))", body, re.M)
        chain = best = total = 0
        pending = 0
        for e in ev:
            if "load" in e:
                pending += 1
            else:
                zero = "vmcnt(0)" in e
                if zero and pending == 1:
                    chain += 1
                    total += 1
                    best = max(best, chain)
                elif zero and pending > 1:
                    chain = 0
                if zero:
                    pending = 0
        if best >= thr:
            print(f"{best:4d} longest chain, {total:4d} single-load waits   {name[:110]}")


main()
