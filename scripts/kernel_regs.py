"""Developer tool (no GPU): registers, scratch and spills of the built kernels whose name contains one of the given substrings.
usage: python scripts/kernel_regs.py place_hash64 place_packed16s [--lib path.so]"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rappas_amd.tools import check_isa as ci


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rappas_amd", "librappas_place.so")
    for a in sys.argv[1:]:
        if a.startswith("--lib="):
            lib = a[6:]
    with tempfile.TemporaryDirectory() as d:
        for co in ci.device_code_objects(lib, d):
            notes = subprocess.run([ci._tool("llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
            for e in re.split(r"\n\s+- \.agpr_count:", "\n" + notes)[1:]:
                n = re.search(r"\.name:\s+(\S+)", e)
                if not n or not any(s in n.group(1) for s in args):
                    continue
                name = subprocess.run(["c++filt", n.group(1)], capture_output=True, text=True).stdout.strip()
                g = lambda k: re.search(r"\." + k + r":\s+(\d+)", e).group(1)
                print(f"{name[:110]:110s} vgpr {g('vgpr_count'):>3s} sgpr {g('sgpr_count'):>3s} scratch {g('private_segment_fixed_size'):>4s} B  spills {g('vgpr_spill_count')}")


if __name__ == "__main__":
    main()
