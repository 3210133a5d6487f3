#!/usr/bin/env python3
"""Registers / LDS / occupancy of every kernel of the translation unit csrc/lam_hip.hip (+ csrc/lam_*.h) as the compiler reports them
(-Rpass-analysis=kernel-resource-usage; cross-compiles, no GPU needed).
    usage: kernel_resources.py [substring ...] [--tuning]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "2024-eumaster4hpc-student-challenge_amd", "csrc", "lam_hip.hip")


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
           '-DLAM_SOURCE_ID="x"', "-Rpass-analysis=kernel-resource-usage", "-c", SRC, "-o", "/tmp/lam_res.o"]
    if "--tuning" in sys.argv:
        cmd.insert(1, "-DLAM_TUNING_VARIANTS")
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    names = re.findall(r"Function Name: (\S+)", txt)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
    for b, d in zip(blocks, dem):
        if args and not any(a in d for a in args):
            continue
        g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]   # noqa: E731
        d = re.sub(r"\(lam::.*$", "", d).replace("void lam::", "")
        print(f"{d[:100]:100s} VGPR {g('VGPRs'):>3} AGPR {g('AGPRs'):>3} SGPR {g('SGPRs'):>3} waves/SIMD {g('Occupancy .waves/SIMD.'):>2} "
              f"LDS {g('LDS Size .bytes/block.'):>6} scratch {g('ScratchSize .bytes/lane.')}")


if __name__ == "__main__":
    main()
