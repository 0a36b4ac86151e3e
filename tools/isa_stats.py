#!/usr/bin/env python3
"""Per-kernel instruction count / registers / scratch from the device assembly of qmcp_kernels.hip.
usage: tools/isa_stats.py [name substring ...]   (writes /tmp/isa/kernels.s; -k keeps an existing one)"""
import os, re, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/isa/kernels.s"
args = [a for a in sys.argv[1:] if a != "-k"]
if "-k" not in sys.argv or not os.path.exists(out):
    os.makedirs("/tmp/isa", exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{R}/include",
                    f"-I{R}/genome-downsampler_amd/csrc", "--cuda-device-only", "-S",
                    f"{R}/genome-downsampler_amd/csrc/qmcp_kernels.hip", "-o", out], check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
for m in re.finditer(r'^(_ZN4qmcp\w+):.*?\n(.*?)\n\s*\.amdhsa_kernel \1(.*?)\.end_amdhsa_kernel', s, re.S | re.M):
    sym, body, md = m.group(1), m.group(2), m.group(3)
    if args and not any(a in sym for a in args):
        continue
    code = body.split(".section")[0]
    lines = [l.strip() for l in code.split("\n") if l.strip() and not l.strip().startswith((";", ".", "_Z")) and not l.strip().endswith(":")]
    def n(pat): return sum(1 for l in lines if re.match(pat, l))
    g = lambda k: re.search(k + r" (\d+)", md).group(1)
    print(f"{sym[:60]:60s} instr {len(lines):5d}  valu {n(r'v_'):5d} salu {n(r's_(?!waitcnt|barrier|load|nop)'):5d} smem {n(r's_load|s_buffer'):3d} "
          f"vmem {n(r'global_|flat_|buffer_'):3d} lds {n(r'ds_'):3d} barrier {n(r's_barrier'):2d} | vgpr {g('amdhsa_next_free_vgpr')} sgpr {g('amdhsa_next_free_sgpr')} "
          f"scratch {g('amdhsa_private_segment_fixed_size')} lds {g('amdhsa_group_segment_fixed_size')}")
