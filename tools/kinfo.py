"""Print VGPR/AGPR/scratch/spill metadata of every kernel in a HIP source (cross-compiles for gfx950)."""
import re, subprocess, sys
src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                      "--cuda-device-only", "-S", src, "-o", "-"], capture_output=True, text=True).stdout
for blk in asm.split("- .agpr_count:")[1:]:
    g = lambda k: re.search(rf"\.{k}:\s+(\S+)", blk)
    name = g("name").group(1)
    if pat and pat not in name: continue
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.split("(")[0]
    print(f"{dem:70s} agpr={blk.split()[0]:>4s} vgpr={g('vgpr_count').group(1):>4s} spill={g('vgpr_spill_count').group(1):>3s} "
          f"scratch={g('private_segment_fixed_size').group(1):>5s} sgpr={g('sgpr_count').group(1):>4s} lds={g('group_segment_fixed_size').group(1)}")
