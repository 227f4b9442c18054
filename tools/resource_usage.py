"""One line per kernel out of csrc/resource_usage.txt (`make -C cpu-raytracer_amd/csrc asm`): registers, scratch, occupancy, spills, LDS."""
import os, re, sys
p = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cpu-raytracer_amd", "csrc", "resource_usage.txt")
keys = [("SGPR", "TotalSGPRs"), ("VGPR", "VGPRs"), ("scratch", "ScratchSize [bytes/lane]"), ("waves/SIMD", "Occupancy [waves/SIMD]"),
        ("sgpr-spill", "SGPRs Spill"), ("vgpr-spill", "VGPRs Spill"), ("LDS", "LDS Size [bytes/block]")]
for b in re.split(r"remark: Function Name: ", open(p).read())[1:]:
    name = b.split()[0]
    if "k_" not in name:
        continue
    vals = []
    for label, k in keys:
        m = re.search(re.escape(k) + r": (\d+)", b)
        vals.append(f"{label} {m.group(1) if m else '?':>5}")
    print(f"{name[:64]:64s} " + "  ".join(vals))
