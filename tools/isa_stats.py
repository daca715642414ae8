#!/usr/bin/env python
"""ISA statistics of the product kernels (measurement tool): compiles g2048_kernels.hip to gfx950 assembly with the
product's flags and prints, per kernel whose demangled name contains the filter, VGPR / SGPR / LDS / scratch and the
static instruction mix (VALU count, quarter-rate multiplies, v_perm, v_bitop3, DPP adds, LDS ops, branches).

    python tools/isa_stats.py ['step_kernel<1, true, true, false>'] [--keep out.s] [--src file.hip]
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = [a for a in sys.argv[1:]]
    keep = src = None
    if "--keep" in args:
        k = args.index("--keep"); keep = args[k + 1]; del args[k:k + 2]
    if "--src" in args:
        k = args.index("--src"); src = args[k + 1]; del args[k:k + 2]
    flt = args[0] if args else "step_kernel<1, true, true, false>"
    src = src or os.path.join(ROOT, "gym-2048_amd", "csrc", "g2048_kernels.hip")
    out = keep or os.path.join(tempfile.mkdtemp(), "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm",
                           "-amdgpu-kernarg-preload-count=16", "-S", "--cuda-device-only", "-o", out, src],
                          stderr=subprocess.DEVNULL)
    text = open(out).read()
    names = re.findall(r"^\t\.globl\t(\S+)", text, re.M)
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
    for mangled, pretty in zip(names, dem):
        if flt not in pretty:
            continue
        start = text.index("\n" + mangled + ":")
        end = text.index(".end_amdhsa_kernel", start)
        body = text[start:text.index("s_endpgm", start)]
        ops = []
        for line in body.splitlines():
            t = line.strip()
            if not line.startswith("\t") or not t or t[0] in ".;":
                continue
            ops.append((t.split()[0], t))
        c = collections.Counter(o for o, _ in ops)
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        dpp = sum(1 for o, t in ops if "row_shr" in t or "row_bcast" in t or "quad_perm" in t or o.endswith("_dpp"))
        meta = text[end - 3000:end + 2500]
        get = lambda key: (re.search(key + r":\s*(\d+)", text[end:end + 3000]) or [None, "?"])[1]
        print(pretty)
        print(f"  NumVgprs {get('; NumVgprs')}  NumSgprs {get('; NumSgprs')}  LDS {get('; LDSByteSize')} B  scratch {get('; ScratchSize')}  occupancy {get('; Occupancy')}")
        print(f"  static instructions {len(ops)}: VALU {valu}, SALU {sum(v for k, v in c.items() if k.startswith('s_') and not k.startswith(('s_waitcnt', 's_nop', 's_load', 's_cbranch', 's_branch')))}, "
              f"s_nop {c['s_nop']}, s_waitcnt {c['s_waitcnt']}, branches {sum(v for k, v in c.items() if 'branch' in k)}, "
              f"LDS {sum(v for k, v in c.items() if k.startswith('ds_'))}, global {sum(v for k, v in c.items() if k.startswith('global_'))}, s_load {sum(v for k, v in c.items() if k.startswith('s_load'))}")
        print(f"  v_mad_u64_u32 {c['v_mad_u64_u32']}, v_mul_* {sum(v for k, v in c.items() if k.startswith('v_mul'))}, v_perm_b32 {c['v_perm_b32']}, "
              f"v_bitop3_b32 {c['v_bitop3_b32']}, DPP {dpp}, v_bcnt {c['v_bcnt_u32_b32']}, v_cndmask {c['v_cndmask_b32_e32'] + c['v_cndmask_b32_e64']}, "
              f"v_mov {c['v_mov_b32_e32'] + c['v_mov_b32_e64']}, v_readlane/firstlane {c['v_readlane_b32'] + c['v_readfirstlane_b32']}")


if __name__ == "__main__":
    main()
