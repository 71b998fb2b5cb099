"""Per-kernel ISA statistics from a hipcc -save-temps .s file (registers, scratch, instruction mix)."""
import re
import sys


def main():
    s = open(sys.argv[1]).read()
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    for m in re.finditer(r"^(_Z\w+):\s*; @\w+\n(.*?)^\s*\.end_amdhsa_kernel", s, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if pat and pat not in name:
            continue
        def f(k):
            r = re.search(k + r"[ ,]+(\d+)", body)
            return r.group(1) if r else "?"
        print(name[:90])
        print("   vgpr", f(r"\.num_vgpr"), "agpr", f(r"\.num_agpr"), "scratch", f(r"\.amdhsa_private_segment_fixed_size"),
              "lds", f(r"\.amdhsa_group_segment_fixed_size"),
              "| mfma", body.count("v_mfma"), "glds", body.count("global_load_lds"), "ds_read_b128", body.count("ds_read_b128"),
              "ds_write", body.count("ds_write"), "vmcnt(0)", len(re.findall(r"vmcnt\(0\)", body)), "s_barrier", body.count("s_barrier"),
              "scratch_ops", body.count("scratch_"))


main()
