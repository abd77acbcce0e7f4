"""Registers, spills, scratch and LDS of every walk-kernel variant in a device-only assembly listing:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I dctdomain_amd/csrc [-DDCTFP_EXPERIMENTS] --cuda-device-only -S -o /tmp/k_walk.s dctdomain_amd/csrc/k_walk.hip
    python tools/kernel_regs.py /tmp/k_walk.s [name-substring]"""
import re, sys
text = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else 'walk_ab_kernel'
meta = text[text.rindex('amdhsa.kernels:'):]
for blk in meta.split('  - .agpr_count:')[1:]:
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    if want not in name:
        continue
    g = lambda k: re.search(k + r':\s+(\d+)', blk).group(1)
    print(f"{name:75s} agpr {blk.split()[0]:>3s} vgpr {g('.vgpr_count'):>3s} sgpr {g('.sgpr_count'):>3s} vspill {g('.vgpr_spill_count'):>3s} "
          f"sspill {g('.sgpr_spill_count'):>3s} scratch {g('.private_segment_fixed_size'):>4s} lds {g('.group_segment_fixed_size'):>6s}")
