#!/usr/bin/env python3
"""Compiler resource usage of every step-kernel instantiation -> profiles/<label>_resource_usage.txt
(hipcc -Rpass-analysis=kernel-resource-usage; runs without a GPU).  python tools/resource_usage.py r02"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "cudaparticlesfoam_amd", "csrc")


def collect():
    rows = []
    for src in ("cpf_stream.hip", "cpf_kernels.hip"):
        extra = ["-mllvm", "--amdgpu-sched-strategy=max-ilp"] if src == "cpf_stream.hip" else []      # as in csrc/Makefile
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "-ffp-contract=off"] + extra +
                           ["-I" + os.path.join(ROOT, "include"), "-I" + CS, "-c", os.path.join(CS, src), "-o", "/dev/null",
                            "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
        txt = re.sub(r" \[-Rpass-analysis=kernel-resource-usage\]", "", r.stderr)
        for b in re.split(r"(?=remark: [^\n]*Function Name:)", txt):
            m = re.search(r"Function Name:\s*(\S+)", b)
            if not m:
                continue
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
            if "step_kernel" not in name:
                continue
            g = lambda k: (re.search(k + r":\s*(\S+)", b) or [None, "?"])[1]   # noqa: E731
            rows.append((name, g("VGPRs"), g("AGPRs"), g("TotalSGPRs"), g(r"ScratchSize \[bytes/lane\]"),
                         g(r"Occupancy \[waves/SIMD\]"), g("SGPRs Spill"), g("VGPRs Spill"), g(r"LDS Size \[bytes/block\]")))
    return rows


def main():
    label = sys.argv[1] if len(sys.argv) > 1 else "r03"
    rows = collect()
    out = ["# Compiler resource usage of the step kernels: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off",
           "# -Rpass-analysis=kernel-resource-usage (ROCm 7.2).  tools/resource_usage.py " + label,
           "# kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | waves/SIMD (registers) | SGPR spills | VGPR spills | LDS B/block"]
    out += [" | ".join(r) for r in rows]
    path = os.path.join(ROOT, "profiles", label + "_resource_usage.txt")
    open(path, "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
