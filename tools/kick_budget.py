#!/usr/bin/env python3
"""CPU (hipcc cross-compiles): the instruction budget of the Brownian kick.  Compiles `normal3` (Philox4x32-R + the fp32 Box-Muller,
csrc/cpf_walk.h) alone in a one-line kernel for gfx950 and prints its instruction mix, for R = 7 (the product) and R = 10, next to the
static instruction counts of the streaming kernel's instantiations (D = 0 flat, D = 0, with the kick).
  python tools/kick_budget.py > profiles/r05_kick_instruction_mix.txt"""
import collections, os, re, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "cudaparticlesfoam_amd", "csrc")
QUARTER = ("v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_log_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_f32", "v_exp_f32",
           "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_cvt_f64_f32", "v_cvt_f32_f64")
SRC = r'''
#include <hip/hip_runtime.h>
#include "cpf_device.h"
#include "cpf_walk.h"
__global__ void k(const uint64_t* g, double* out, uint32_t step, uint32_t seed) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    const cpf::D3 r = cpf::normal3(g[i], step, seed);
    out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
}
'''


def asm(path, flags, tmp=tempfile.gettempdir()):
    out = os.path.join(tmp, os.path.basename(path) + ".budget.s")          # never next to the sources
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                    "-I" + CS, "-S", "--cuda-device-only", path, "-o", out] + flags, check=True, stderr=subprocess.DEVNULL)
    return open(out).read()


def mix(text):
    ins = [l.split()[0] for l in text.split("\n") if re.match(r"\s+(v_|s_|ds_|global_|buffer_|flat_)", l)]
    return collections.Counter(ins)


def main():
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "k.hip"); open(p, "w").write(SRC)
        for R in (7, 10):
            c = mix(asm(p, ["-DCPF_PHILOX_ROUNDS=%d" % R]))
            valu = {k: v for k, v in c.items() if k.startswith("v_")}
            q = sum(v for k, v in valu.items() if k.split("_e")[0] in QUARTER or k in QUARTER)
            print("normal3 with Philox4x32-%d: %d instructions in the kernel (incl. 1 load, 3 stores, address arithmetic), %d vector, of which %d quarter-rate"
                  % (R, sum(c.values()), sum(valu.values()), q))
            print("  " + ", ".join("%s x %d" % (k, v) for k, v in sorted(valu.items(), key=lambda kv: -kv[1])))
        text = asm(os.path.join(CS, "cpf_stream.hip"), ["-mllvm", "--amdgpu-sched-strategy=max-ilp"])
        cur, bodies = None, {}
        for l in text.split("\n"):
            m = re.match(r"^(_ZN3cpf18step_kernel_stream\S*):", l)
            if m:
                cur = m.group(1); bodies[cur] = []
                continue
            if l.startswith(".Lfunc_end"):
                cur = None
            if cur:
                bodies[cur].append(l)
        print("static instruction counts of step_kernel_stream<BROWNIAN, REFLECT, STORE_VEL, STATS, LOOKUP> (whole kernel, all paths):")
        for name, b in bodies.items():
            t = re.search(r"ILb(\d)ELb(\d)ELb(\d)ELb(\d)ELi(\d+)E", name).groups()
            if t[1:4] == ("1", "0", "0") and t[4] in ("0", "8", "6"):
                c = mix("\n".join(b))
                print("  <%s, 1, 0, 0, %s>: %d instructions, %d vector, %d scalar, %d v_mad_u64_u32, %d v_writelane (scalar spills)"
                      % (t[0], t[4], sum(c.values()), sum(v for k, v in c.items() if k.startswith("v_")),
                         sum(v for k, v in c.items() if k.startswith("s_")), c["v_mad_u64_u32"], c["v_writelane_b32"]))


if __name__ == "__main__":
    main()
